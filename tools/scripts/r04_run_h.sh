O=gpurun_out
timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_configs_gpu.py tests/test_fp8_gpu.py -m gpu -q > $O/r04_t_h.txt 2>&1; tail -5 $O/r04_t_h.txt
bash tools/scripts/deeplab_trace.sh > $O/r04_dl_trace2.txt 2>&1; head -40 $O/r04_dl_trace2.txt | cut -c1-190
