# kernel trace of three training steps + the timeline / per-kernel summary of the last one -> gpurun_out/trace_step/
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
O=$R/gpurun_out/trace_step
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw -o st -- python3 $R/bench.py --steps 3 --warmup 2 --repeats 1 --no-infer --no-cpu-baseline > $O/bench.log 2>&1
cd $R
python3 tools/timeline.py $O/raw -v > $O/timeline.txt 2>&1
cp $(find $O/raw -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv 2>/dev/null
rm -rf $O/raw
head -45 $O/timeline.txt
