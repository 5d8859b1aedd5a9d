"""GPU timeline of the last training step in a rocprofv3 kernel trace: union busy time, idle gaps, per-queue busy time,
time per kernel class, and (with -v) every launch."""
import csv, glob, re, sys
from collections import defaultdict
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
# the steps between consecutive Adam launches; the SHORTEST of the last three is shown (a profiled run now and then holds one step up for
# milliseconds on the host side -- seen as a single multi-ms gap -- which says nothing about the kernels)
cands = [rows[idx[k] + 1: idx[k + 1] + 1] for k in range(max(len(idx) - 4, 0), len(idx) - 1)]
step = min(cands, key=lambda st: max(int(r['End_Timestamp']) for r in st) - int(st[0]['Start_Timestamp']))
t0 = int(step[0]['Start_Timestamp']); t1 = max(int(r['End_Timestamp']) for r in step)
print(f'step wall {(t1 - t0) / 1e3:.1f} us, {len(step)} launches')
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in step)
busy, cur_s, cur_e, gaps = 0, iv[0][0], iv[0][1], []
for s, e in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append(s - cur_e); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f'union busy {busy / 1e3:.1f} us, idle {sum(gaps) / 1e3:.1f} us in {len(gaps)} gaps (max {max(gaps) / 1e3:.1f} us)')
q = defaultdict(int)
for r in step:
    q[r.get('Queue_Id', '?')] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
print('per queue busy us:', {k: round(v / 1e3, 1) for k, v in q.items()})


def cls(n):
    for k in ('convt_bwd_fused', 'bwd_fused', 'igemm_m16sym', 'igemm_m16', 'igemm_tr_kernel', 'reduce_slabs_batched', 'convt_thin_dgrad', 'convt_thin', 'igemm_ws', 'igemm_fast', 'igemm_kernel', 'wgrad_dma', 'wgrad_db', 'bn_bwd_finalize2', 'wgrad_kernel', 'wgrad_reduce', 'bn_bwd_dense', 'bn_bwd_finalize', 'bn_bwd', 'bn_relu_pool', 'bn_finalize',
              'pack_kernel', 'head_', 'loss', 'adam', 'Fill', 'ingest', 'dropout', 'maxpool'):
        if k in n:
            return k
    return n[:40]


c = defaultdict(lambda: [0, 0])
for r in step:
    k = cls(r['Kernel_Name']); c[k][0] += int(r['End_Timestamp']) - int(r['Start_Timestamp']); c[k][1] += 1
for k, (t, n) in sorted(c.items(), key=lambda kv: -kv[1][0]):
    print(f'{k:28s} {t / 1e3:9.1f} us {n:4d} launches')
if '-v' in sys.argv:
    for r in step:
        n = re.sub(r'\(.*', '', r['Kernel_Name'])[:60]
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f} q{r.get('Queue_Id', '?')} "
              f"grid {int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']):>7d} {n}")
