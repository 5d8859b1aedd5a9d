# times + rocprofv3 kernel statistics of the other model families (tools/family_time.py) -> gpurun_out/family_times.txt, family_kernel_stats.csv
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/fam_raw
timeout 600 python3 $R/tools/family_time.py > $O/family_times.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fam_raw -o f -- python3 $R/tools/family_time.py > $O/family_trace.log 2>&1
cp $(find $O/fam_raw -name "*kernel_stats.csv" | head -1) $O/family_kernel_stats.csv 2>/dev/null
rm -rf $O/fam_raw
grep -v amdgpu.ids $O/family_times.txt; head -22 $O/family_kernel_stats.csv | cut -c1-150
