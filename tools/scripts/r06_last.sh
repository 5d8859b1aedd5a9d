R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
O=$R/gpurun_out
cd $R
timeout 300 python3 __graft_entry__.py smoke > $O/r06_smoke.log 2>&1; tail -1 $O/r06_smoke.log
timeout 2400 python3 -m pytest tests -q -m gpu > $O/r06_tests_full.log 2>&1
tail -3 $O/r06_tests_full.log
timeout 500 python3 bench.py --steps 20 --warmup 5 > $O/last_bench.json 2> $O/last_bench.err
python3 -c "
import json; d=json.loads(open('$O/last_bench.json').read().strip().splitlines()[-1]); r=d['roofline']; print(d['ms_per_step'], d['value'], r['frac'], r['traffic'], r['traffic_stale'], d['extra']['box_probe'])"
