cd $GRAFT_REPO_ROOT
for v in wb0 wb1 wb2; do
  echo "== $v (wb0: launch_bounds(64,2) max-ilp; wb1: launch_bounds(64,1); wb2: (64,2) default scheduler)"
  ALORE_NMPC_LIB=$PWD/ab/libalore_nmpc_$v.so python3 tools/wb_profile.py 2>&1 | tail -4
done
