O=gpurun_out
bash tools/scripts/refresh_profiles.sh > $O/r04_refresh.log 2>&1; tail -3 $O/r04_refresh.log | cut -c1-300
timeout 300 python tools/step_probe.py > $O/r04_step_probe.txt 2>&1; tail -12 $O/r04_step_probe.txt
timeout 300 python tools/conv_probe.py > $O/r04_conv_probe.txt 2>&1; tail -5 $O/r04_conv_probe.txt
timeout 300 python bench.py --channels 13 --no-cpu-baseline > $O/r04_bench_13band.json 2>/dev/null; cut -c1-300 $O/r04_bench_13band.json
