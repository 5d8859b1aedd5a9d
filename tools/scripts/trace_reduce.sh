R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
cd $R; mkdir -p gpurun_out/red; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/red -o red -- python3 $R/tools/step_probe.py --only wgrad > $R/gpurun_out/red/probe.txt 2>&1
f=$(find /tmp/red -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY' > $R/gpurun_out/red/reduce_launches.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# pair each reduce kernel with the wgrad kernel just before it
prev = None; out = collections.OrderedDict()
for r in rows:
    nm = r['Kernel_Name']; d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    if 'wgrad_reduce' in nm and prev is not None:
        key = (prev[0][:70], nm.split('(')[0], prev[2], r['Grid_Size'] if 'Grid_Size' in r else r.get('Grid_Size_X',''))
        out.setdefault(key, []).append((prev[1], d))
    if 'wgrad' in nm and 'reduce' not in nm: prev = (nm, d, r.get('Grid_Size', r.get('Grid_Size_X','')))
for k, v in out.items():
    v = v[len(v)//2:]
    print(f"{k[0]:72s} grid {k[2]:>8s} main {sum(a for a,_ in v)/len(v):8.1f} us | {k[1]:24s} grid {k[3]:>8s} {sum(b for _,b in v)/len(v):7.1f} us  (n={len(v)})")
PY
