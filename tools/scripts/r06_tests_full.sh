R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
O=$R/gpurun_out
cd $R
timeout 300 python3 __graft_entry__.py smoke > $O/r06_smoke.log 2>&1; tail -1 $O/r06_smoke.log
timeout 2400 python3 -m pytest tests -q -m gpu --durations=6 > $O/r06_tests_full.log 2>&1
tail -12 $O/r06_tests_full.log
