# round-6 baseline on one box: new tests, bench line (events off in the timed regions), isolated step probe
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
O=$R/gpurun_out
cd $R
timeout 900 python3 -m pytest tests/test_lstm_gpu.py tests/test_ops_gpu.py -x -q -m gpu -k "graph or reduce_slabs" > $O/r06_base_tests.log 2>&1
tail -3 $O/r06_base_tests.log
timeout 400 python3 bench.py --steps 20 --warmup 5 --no-infer --no-cpu-baseline > $O/r06_base_bench.json 2> $O/r06_base_bench.err
timeout 400 python3 bench.py --steps 20 --warmup 5 --no-infer --no-cpu-baseline --prof-in-timed > $O/r06_base_bench_prof.json 2>> $O/r06_base_bench.err
tail -c 600 $O/r06_base_bench.json
timeout 600 python3 tools/step_probe.py > $O/r06_base_step_probe.txt 2>&1
tail -8 $O/r06_base_step_probe.txt
