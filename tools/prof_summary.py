"""Summarise a rocprofv3 --kernel-trace --stats CSV directory: per-kernel totals per training step."""
import csv, glob, sys, re
d = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1
f = glob.glob(d + '/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f'total kernel time {tot/1e6:.2f} ms = {tot/1e6/steps:.3f} ms/step')
groups = {}
for r in rows:
    n = r['Name']
    key = re.sub(r'<.*', '', n)
    for pat, k in (('igemm_fast', 'igemm_fast'), ('igemm_kernel', 'igemm_generic'), ('wgrad_kernel', 'wgrad'), ('wgrad_reduce', 'wgrad_reduce'),
                   ('bn_bwd_kernel', 'bn_bwd'), ('bn_relu_pool', 'bn_relu_pool'), ('head_bwd', 'head_bwd'), ('head_fwd', 'head_fwd'),
                   ('pack_kernel', 'pack'), ('adam', 'adam'), ('loss_kernel', 'loss'), ('finalize', 'bn_finalize'), ('ingest', 'ingest'),
                   ('elementwise', 'torch_fill/copy'), ('copyBuffer', 'torch_fill/copy'), ('rccl', 'rccl'), ('nccl', 'rccl')):
        if pat in n:
            key = k
            break
    g = groups.setdefault(key, [0.0, 0])
    g[0] += float(r['TotalDurationNs']); g[1] += int(r['Calls'])
for k, (ns, c) in sorted(groups.items(), key=lambda kv: -kv[1][0]):
    print(f'{ns/1e6/steps:9.3f} ms/step {100*ns/tot:6.2f}%  calls/step {c/steps:7.1f}  {k}')
