"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs) of bench.py into HBM bytes per launch.

gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 64 B per 128-B request of a wide
coalesced stream, i.e. HALF the bytes -> doubled here; WRITE_SIZE is exact for 16-byte-per-lane stores.
Counter unit: KiB."""
import csv, glob, json, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fdir, wdir, out = sys.argv[1], sys.argv[2], sys.argv[3]
commit = sys.argv[4] if len(sys.argv) > 4 else 'unknown'
CLASSES = (('convt_bwd_fused', 'convT_bwd_fused'), ('bwd_fused', 'bwd_fused_3x3'), ('igemm_m16', 'igemm_3x3'), ('igemm_tr_kernel', 'igemm_3x3'), ('reduce_slabs_batched', 'wgrad_reduce'), ('convt_thin', 'convT_streaming'), ('Li9ELb', 'igemm_3x3'), ('igemm_fast', 'igemm_1x1_convT'), ('igemm_kernel', 'igemm_generic'), ('wgrad_kernel', 'wgrad'), ('wgrad_db_kernel', 'wgrad'), ('wgrad_dma_kernel', 'wgrad'), ('wgrad_reduce', 'wgrad_reduce'), ('igemm_ws', 'igemm_3x3'),
           ('bn_bwd', 'bn_bwd'), ('bn_relu_pool', 'bn_relu_pool'), ('head_', 'head'), ('adam', 'adam'), ('pack_kernel', 'pack'))

def load(d):
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        key = 'other'
        for pat, k in CLASSES:
            if pat in n:
                key = k
                break
        agg[key][0] += float(r['Counter_Value']) * 1024.0
        agg[key][1] += 1
    return agg

fe, wr = load(fdir), load(wdir)
res = {}
for k in sorted(set(fe) | set(wr)):
    fb, fc = fe.get(k, [0, 0]); wb, wc = wr.get(k, [0, 0])
    calls = max(fc, wc, 1)
    res[k] = {'launches': calls, 'fetch_bytes_raw': fb, 'fetch_bytes_corrected_x2': 2 * fb, 'write_bytes': wb,
              'hbm_bytes_per_launch': (2 * fb + wb) / calls}
import bench as _bench
json.dump({'commit': commit, 'csrc_digest': _bench.csrc_digest(), 'note': 'FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 64 B per 128-B request); separate --pmc passes; '
                   'bench.py --steps 3 --warmup 2 --repeats 1 (5 training steps, batch 64; per-launch figures do not depend on the step count)', 'classes': res}, open(out, 'w'), indent=1)
for k, v in res.items():
    print(f"{k:24s} launches {v['launches']:5d}  HBM bytes/launch {v['hbm_bytes_per_launch']/1e6:9.1f} MB")
