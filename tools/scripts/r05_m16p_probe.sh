# per-shape A/B of the persistent 16x16x32 kernel (option m16p) against the one-tile kernels, forward with statistics, with and without the fused input BatchNorm
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
cd $R
SH="64,128,128,64,128 64,128,128,128,128 64,64,64,64,128 64,64,64,128,128 64,64,64,128,256 64,64,64,256,128 64,32,32,256,256 64,32,32,256,512 64,32,32,512,256 64,16,16,256,512"
for rep in 1 2; do
  for o in "m16p=0" "m16p=1" "m16p=2"; do
    echo "== $o"; python3 tools/conv_probe.py --opt igemm_m16=2 --opt $o --shapes $SH 2>&1 | grep " n64"
    echo "== $o --affine"; python3 tools/conv_probe.py --opt igemm_m16=2 --opt $o --affine --shapes $SH 2>&1 | grep " n64"
  done
done
