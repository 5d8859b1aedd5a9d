"""Summarise a rocprofv3 --pmc pass of bench.py with SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE, SQ_LDS_BANK_CONFLICT, SQ_LDS_IDX_ACTIVE,
SQ_WAIT_ANY, SQ_WAVE_CYCLES into per-kernel-class MFMA utilisation and LDS bank-conflict rate.

MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES (pipe-busy cycles summed over every SIMD of the chip; 32 per v_mfma_f32_32x32x16_bf16,
MI355X_MICROARCH.md) / (GRBM_GUI_ACTIVE cycles of the dispatch x 1024 SIMDs)."""
import csv, glob, json, sys, collections
d, out = sys.argv[1], sys.argv[2]
CLASSES = (('convt_bwd_fused', 'convT_bwd_fused'), ('bwd_fused', 'bwd_fused_3x3'), ('igemm_m16', 'igemm_3x3'), ('igemm_tr_kernel', 'igemm_3x3'), ('reduce_slabs_batched', 'wgrad_reduce'), ('convt_thin', 'convT_streaming'), ('Li9ELb', 'igemm_3x3'), ('igemm_fast', 'igemm_1x1_convT'), ('igemm_kernel', 'igemm_generic'), ('wgrad_kernel', 'wgrad'), ('wgrad_db_kernel', 'wgrad'), ('wgrad_dma_kernel', 'wgrad'), ('wgrad_reduce', 'wgrad_reduce'), ('igemm_ws', 'igemm_3x3'),
           ('bn_bwd', 'bn_bwd'), ('bn_relu_pool', 'bn_relu_pool'), ('head_', 'head'))
f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
CLOCK_GHZ = 2.4                                  # peak engine clock; the dispatch duration comes from the kernel trace of the same run
dur = {r['Dispatch_Id']: int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in csv.DictReader(open(glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]))}
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    key = next((k for pat, k in CLASSES if pat in n), None)
    if key is None:
        continue
    agg[key][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Dispatch_Id'] not in disp[key]:
        agg[key]['_ns'] += dur.get(r['Dispatch_Id'], 0)
    disp[key].add(r['Dispatch_Id'])
res = {}
for k, c in agg.items():
    gui = max(c.get('GRBM_GUI_ACTIVE', 0.0), 1.0)
    res[k] = {'dispatches': len(disp[k]),
              'mfma_util': round(c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / max(c['_ns'] * CLOCK_GHZ * 1024, 1.0), 4),
              'mfma_util_vs_gui_active': round(c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (gui * 1024 / 8), 4),
              'lds_bank_conflict_rate': round(c.get('SQ_LDS_BANK_CONFLICT', 0.0) / max(c.get('SQ_LDS_IDX_ACTIVE', 0.0), 1.0), 4),
              'wave_wait_fraction': round(c.get('SQ_WAIT_ANY', 0.0) / max(c.get('SQ_WAVE_CYCLES', 0.0), 1.0), 4)}
json.dump({'note': 'rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES '
                   '-- python3 bench.py --steps 3 --warmup 2 --repeats 1 --no-cpu-baseline; mfma_util = MFMA pipe-busy cycles / (dispatch ns x 2.4 GHz x 1024 SIMDs); _vs_gui_active uses GRBM_GUI_ACTIVE/8 (the counter sums the 8 XCDs); '
                   'kernels of both streams overlap, so GUI_ACTIVE of a dispatch includes time shared with the other stream',
           'classes': res}, open(out, 'w'), indent=1)
for k, v in res.items():
    print(f"{k:18s} dispatches {v['dispatches']:4d}  MFMA util {v['mfma_util']:.3f}  LDS conflict rate {v['lds_bank_conflict_rate']:.3f}  wave-wait {v['wave_wait_fraction']:.3f}")
