# kernel statistics of the folded inference plan, bf16 and fp8 storage
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for m in bf16 fp8; do
  rm -rf $O/inf_$m
  if [ $m = fp8 ]; then F=--fp8; else F=; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/inf_$m -o inf -- python3 $R/tools/infer_profile.py $F --reps 5 > $O/inf_$m.log 2>&1
  tail -1 $O/inf_$m.log
  python3 - $O/inf_$m <<'PY'
import csv, glob, sys, re
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:14]:
    name = r['Name'].split('(')[0][:110]
    print(f"  {float(r['TotalDurationNs']) / 1e3 / 7:9.1f} us/pass {int(r['Calls']):5d} calls  {name}")
PY
  rm -f $O/inf_$m/*kernel_trace.csv
done
