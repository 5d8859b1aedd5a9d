# kernel statistics of DeepLab-v3 / ResNet-50 inference on one 512 x 512 x 4 tile (BASELINE configs[2]) and on 16 -> gpurun_out/dl/
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
O=$R/gpurun_out/dl
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for b in 1 16; do
  B=$b timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw$b -o dl -- python3 $R/tools/deeplab_trace.py > $O/run$b.log 2>&1
  cp $(find $O/raw$b -name "*kernel_stats.csv" | head -1) $O/deeplab_b${b}_kernel_stats.csv 2>/dev/null
  python3 - $O/deeplab_b${b}_kernel_stats.csv $b <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
passes = 4
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"== batch {sys.argv[2]}: {tot / 1e3 / passes:9.1f} us of kernel time per pass, {sum(int(r['Calls']) for r in rows) // passes} launches per pass")
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:16]:
    print(f"  {float(r['TotalDurationNs']) / 1e3 / passes:9.1f} us/pass {int(r['Calls']) // passes:4d} launches  avg {float(r['TotalDurationNs']) / 1e3 / int(r['Calls']):7.1f} us  {r['Name'].split('(')[0][:120]}")
PY
  python3 $R/tools/timeline.py $O/raw$b > $O/timeline_b$b.txt 2>&1
  rm -rf $O/raw$b
done
