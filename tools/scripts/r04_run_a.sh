O=gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 > $O/r04_gpu_tests_a.txt 2>&1; tail -5 $O/r04_gpu_tests_a.txt
timeout 200 python tools/wgrad_probe.py > $O/r04_wgrad_probe3.txt 2>&1; grep TOTAL $O/r04_wgrad_probe3.txt
for w in 128 160; do echo "== WGS $w"; SATCV_WGRAD_WGS=$w timeout 300 python bench.py --no-cpu-baseline --no-infer 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['extra'].get('region_ms_per_step'))"; done > $O/r04_ab_wgs2.txt 2>&1; cat $O/r04_ab_wgs2.txt
