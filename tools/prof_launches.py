"""Per-launch table of the conv kernels in the last training step of a rocprofv3 kernel trace."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('adam_kernel')]
step = rows[idx[-2] + 1: idx[-1] + 1]
t0 = int(step[0]['Start_Timestamp'])
pat = sys.argv[2] if len(sys.argv) > 2 else 'igemm|wgrad_kernel'
import re
for r in step:
    n = r['Kernel_Name']
    if re.search(pat, n):
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        short = (n.replace('_Z17igemm_fast_kernelIDF16b', 'F<').replace('EEv9IgemmArgs', '>').replace('_Z12wgrad_kernelIDF16b', 'W<')
                 .replace('EEv9WgradArgs', '>').replace('_Z12igemm_kernelIDF16b', 'G<').replace('Li', '').replace('E', ','))[:44]
        print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} {d:8.1f} us grid {int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']):>7d}x{r['Grid_Size_Y']:>5s} {short}")
