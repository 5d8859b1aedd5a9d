# round-6 measurement set at the tree being run: default bench line, kernel statistics, PMC traffic / MFMA passes, step probe, timeline
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/fin_stats $O/fin_f $O/fin_w $O/fin_m
timeout 500 python3 $R/bench.py --steps 20 --warmup 5 > $O/fin_bench.json 2> $O/fin_bench.err
timeout 300 python3 $R/bench.py --steps 20 --warmup 5 --channels 13 --no-infer --no-cpu-baseline > $O/fin_bench13.json 2>> $O/fin_bench.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fin_stats -o st -- python3 $R/bench.py --steps 3 --warmup 2 --repeats 1 --no-infer --no-cpu-baseline > $O/fin_stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fin_f -o f -- python3 $R/bench.py --steps 3 --warmup 2 --repeats 1 --no-infer --no-cpu-baseline > $O/fin_f.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/fin_w -o w -- python3 $R/bench.py --steps 3 --warmup 2 --repeats 1 --no-infer --no-cpu-baseline > $O/fin_w.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES --output-format csv -d $O/fin_m -o m -- python3 $R/bench.py --steps 3 --warmup 2 --repeats 1 --no-infer --no-cpu-baseline > $O/fin_m.log 2>&1
cd $R
python3 tools/pmc_summary.py $O/fin_f $O/fin_w $O/fin_pmc_traffic.json "$(cat $R/satellite_computervision_amd/_build_commit.txt 2>/dev/null || echo working-tree)" > $O/fin_pmc_traffic.log 2>&1
python3 tools/pmc_mfma_summary.py $O/fin_m $O/fin_pmc_mfma.json > $O/fin_pmc_mfma.log 2>&1
python3 tools/timeline.py $O/fin_stats -v > $O/fin_timeline.txt 2>&1
cp $(find $O/fin_stats -name '*kernel_stats.csv' | head -1) $O/fin_kernel_stats.csv 2>/dev/null
rm -f $O/fin_f/*kernel_trace.csv $O/fin_w/*kernel_trace.csv $O/fin_m/*kernel_trace.csv
rm -rf $O/fin_stats
timeout 600 python3 tools/step_probe.py > $O/fin_step_probe.txt 2>&1
tail -c 1500 $O/fin_bench.json; echo; head -8 $O/fin_timeline.txt; tail -7 $O/fin_step_probe.txt; cat $O/fin_pmc_traffic.log | head -20
