R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
O=$R/gpurun_out
cd $R
timeout 2400 python3 -m pytest tests -x -q -m gpu --durations=8 > $O/r06_tests.log 2>&1
tail -15 $O/r06_tests.log
