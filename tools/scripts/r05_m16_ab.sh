# A/B/A/B of the deep 3x3 tiles (one process per arm, same box): 32x32x16 double-buffered tile (SATCV_M16=0), the symmetric 16x16x32 kernel
# (SATCV_M16_WS=0) and the 16x16x32 kernel with wave roles (default)
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
cd $R
for aff in "--affine" ""; do
for i in 1 2; do
  echo "== 32x32x16 tile (SATCV_M16=0) $aff"; SATCV_M16=0 python3 tools/conv_probe.py --shapes deep $aff --reps 30
  echo "== 16x16x32 symmetric (SATCV_M16_WS=0) $aff"; SATCV_M16_WS=0 python3 tools/conv_probe.py --shapes deep $aff --reps 30
  echo "== 16x16x32 wave roles (default) $aff"; python3 tools/conv_probe.py --shapes deep $aff --reps 30
done
done 2>&1 | grep -v "amdgpu.ids\|^lib"
