R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
O=$R/gpurun_out
cd $R
timeout 900 python3 -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "wgrad" 2>&1 | tail -3
for v in 0 1; do echo "== SATCV_WGRAD_M16=$v"; SATCV_WGRAD_M16=$v python3 tools/wgrad_probe.py 2>&1 | grep " n64" | tail -10; done > $O/r06_wgm16_probe2.txt 2>&1
cat $O/r06_wgm16_probe2.txt
