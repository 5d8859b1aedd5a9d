O=gpurun_out
timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_ops_gpu.py -m gpu -q -k "deeplab or split_k or invariance or residual or aspp or strid or dilat or siamese" > $O/r04_t_k.txt 2>&1; tail -4 $O/r04_t_k.txt
for i in 1 2; do timeout 200 python tools/deeplab_time.py 2>&1 | grep "^b"; done
SATCV_SPLITK=1 timeout 200 python tools/step_probe.py 2>&1 | grep "8x8" | head -4
