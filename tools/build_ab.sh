#!/bin/bash
# build ablation / variant libraries for tools/ab.sh: tools/build_ab.sh name "EXTRA_HIPFLAGS" [name "flags" ...]
cd /root/repo/scrappie_amd/csrc || exit 1
mkdir -p /root/repo/build/ab
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  rm -f scrappie_hip.o
  make -s ../libscrappie_hip.so EXTRA_HIPFLAGS="$flags" || exit 1
  cp ../libscrappie_hip.so /root/repo/build/ab/lib_$name.so
done
rm -f scrappie_hip.o && make -s ../libscrappie_hip.so
