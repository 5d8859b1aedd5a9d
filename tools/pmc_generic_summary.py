"""Per-kernel (short name + grid) table of arbitrary rocprofv3 --pmc counters of one bench.py pass, sorted by total duration."""
import csv, glob, re, sys, collections
d = sys.argv[1]
f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
dur = {r['Dispatch_Id']: int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in csv.DictReader(open(glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]))}


def short(n):
    n = n.replace('_Z17igemm_fast_kernelI', 'F<').replace('EEv9IgemmArgs', '>').replace('_Z12wgrad_kernelI', 'W<').replace('EEv9WgradArgs', '>')
    n = n.replace('DF16b', 'bf16,').replace('Li', '').replace('ELb', ',b').replace('E', ',')
    return re.sub(r'\(.*', '', n)[:44]


agg = collections.defaultdict(lambda: collections.defaultdict(float))
seen = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    key = (short(r['Kernel_Name']), int(r['Grid_Size']) // max(int(r['Workgroup_Size']), 1))
    agg[key][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Dispatch_Id'] not in seen[key]:
        seen[key].add(r['Dispatch_Id'])
        agg[key]['_ns'] += dur.get(r['Dispatch_Id'], 0)
names = sorted({c for v in agg.values() for c in v if c != '_ns'})
print(f"{'kernel':46s} {'wgs':>6s} {'n':>3s} {'us/launch':>9s}  " + '  '.join(f'{n[:18]:>18s}' for n in names))
for key, v in sorted(agg.items(), key=lambda kv: -kv[1]['_ns'])[:int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    n = len(seen[key])
    print(f'{key[0]:46s} {key[1]:6d} {n:3d} {v["_ns"] / n / 1e3:9.1f}  ' + '  '.join(f'{v.get(c, 0) / n:18.4g}' for c in names))
