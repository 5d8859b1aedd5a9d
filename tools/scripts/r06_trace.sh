R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
export SATCV_EARLY_OPT=0
bash $R/tools/scripts/trace_step.sh > /dev/null 2>&1
cp $R/gpurun_out/trace_step/timeline.txt $R/gpurun_out/r06_timeline_eo0.txt
head -30 $R/gpurun_out/r06_timeline_eo0.txt
