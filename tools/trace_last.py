"""Per-launch list of the LAST repetition in a rocprofv3 kernel trace of tools/infer_trace.py (delimited by the ingest kernel)."""
import csv, glob, re, sys
rows = sorted(csv.DictReader(open(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0])), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'ingest' in r['Kernel_Name']]
last = rows[idx[-1]:]
t0 = int(last[0]['Start_Timestamp'])
tot = 0
for r in last:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot += d
    n = re.sub(r'\(.*', '', r['Kernel_Name']).replace('_Z17igemm_fast_kernelI', 'F<').replace('EEv9IgemmArgs', '>').replace('DF16b', 'bf16,').replace('Li', '').replace('ELb', ',b').replace('E', ',')[:52]
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} {d:7.1f} us  grid {int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1):6d}  {n}")
print(f'sum of kernel time {tot:.1f} us, wall {(int(last[-1]["End_Timestamp"]) - t0) / 1e3:.1f} us')
