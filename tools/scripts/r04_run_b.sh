O=gpurun_out
timeout 2000 python -m pytest tests -m gpu -q --timeout 900 > $O/r04_gpu_tests_b.txt 2>&1; tail -8 $O/r04_gpu_tests_b.txt
timeout 300 python bench.py --no-cpu-baseline > $O/r04_bench8.json 2>/dev/null; cut -c1-400 $O/r04_bench8.json
timeout 300 python tools/infer_probe.py > $O/r04_dl_after2.txt 2>&1; tail -12 $O/r04_dl_after2.txt
