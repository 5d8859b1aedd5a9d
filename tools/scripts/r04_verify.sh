O=gpurun_out
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 > $O/r04_gpu_tests_final.txt 2>&1; tail -4 $O/r04_gpu_tests_final.txt
timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py > $O/r04_bench_final2.json 2> $O/r04_bench_final2.err; cut -c1-300 $O/r04_bench_final2.json
