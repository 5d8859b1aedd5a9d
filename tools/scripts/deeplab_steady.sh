# steady-state kernel breakdown of DeepLab-v3 inference at batch $1 (default 1): 44 passes under rocprofv3, per-pass averages
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
B=${1:-1}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/dl1
B=$B PASSES=44 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/dl1 -o l -- python3 $R/tools/deeplab_trace.py > /dev/null 2>&1
cd $R
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/dl1/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
P = 44
tot = sum(int(r["Calls"]) for r in rows); tt = sum(float(r["TotalDurationNs"]) for r in rows)
print("calls per pass", round(tot / P, 1), "kernel us per pass", round(tt / 1e3 / P, 1))
for r in rows[:24]:
    print(str(round(int(r["Calls"]) / P, 1)).rjust(6), r["AverageNs"][:7].rjust(8), str(round(float(r["TotalDurationNs"]) / 1e3 / P, 1)).rjust(7), r["Name"][:110])
PY
rm -rf gpurun_out/dl1
