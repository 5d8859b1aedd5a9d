# round-4 evidence set -> gpurun_out/ (copied into profiles/r04_* by hand afterwards)
O=gpurun_out
bash tools/scripts/refresh_profiles.sh > $O/r04_refresh.log 2>&1; tail -2 $O/r04_refresh.log | cut -c1-200
timeout 300 python tools/step_probe.py > $O/r04_step_probe.txt 2>&1; grep TOTAL $O/r04_step_probe.txt
timeout 300 python tools/conv_probe.py > $O/r04_conv_probe.txt 2>&1; tail -3 $O/r04_conv_probe.txt
bash tools/scripts/deeplab_trace.sh > $O/r04_dl_trace.txt 2>&1; grep "^==" $O/r04_dl_trace.txt
bash tools/scripts/infer_trace.sh > $O/r04_infer_trace.txt 2>&1; tail -3 $O/r04_infer_trace.txt | cut -c1-150
timeout 300 python bench.py --channels 13 --no-cpu-baseline > $O/r04_bench_13band.json 2>/dev/null; cut -c1-200 $O/r04_bench_13band.json
timeout 200 python tools/lstm_time.py > $O/r04_lstm_time.txt 2>&1; tail -2 $O/r04_lstm_time.txt
timeout 200 python tools/deeplab_time.py > $O/r04_deeplab_time.txt 2>&1; grep "^b" $O/r04_deeplab_time.txt
