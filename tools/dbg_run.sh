cd $GRAFT_REPO_ROOT
for f in build/ab/lib_*.so; do cp $f scrappie_amd/libscrappie_hip.so; echo $f; timeout 100 python tools/dbg_trunk.py 2>&1 | tail -4; done
