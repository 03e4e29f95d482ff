#!/bin/bash
# registers / scratch / instruction shape of one kernel: tools/kres.sh <mangled-name-prefix> [min_mfma]
cd /root/repo/scrappie_amd/csrc || exit 1
mkdir -p /tmp/t
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $KFLAGS -I../../include -I. --cuda-device-only -S -o /tmp/t/k.s scrappie_hip.hip \
  -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A8 "Function Name: $1" | grep -E "Function Name|VGPRs:|ScratchSize|Occupancy" | head -8
awk -v p="$1" 'index($0,p)==1 && $1 ~ /:$/ {f=1} f{print} f && /s_endpgm/{exit}' /tmp/t/k.s > /tmp/t/kk.s
echo "packed VALU: $(grep -c 'v_pk_' /tmp/t/kk.s)"
python3 /root/repo/tools/isa_shape.py /tmp/t/k.s "$1" "${2:-50}"
