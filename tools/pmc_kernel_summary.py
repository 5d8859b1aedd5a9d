"""Per-kernel summary of one or more `rocprofv3 --kernel-trace --pmc ...` passes of the SAME command (counters that do not fit one pass
are collected in separate passes: MI355X_MICROARCH.md, rocprofv3 PMC slots).

    python3 tools/pmc_kernel_summary.py out.json <pass dir> [<pass dir> ...] [--match substr]

For every kernel name: dispatches, mean duration (kernel trace of the first pass) and the mean of every counter per dispatch, plus the
derived figures used in DESIGN.md: MFMA pipe-busy fraction (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)), the shares of
SQ_WAVE_CYCLES spent parked (SQ_WAIT_ANY), stalled at issue (SQ_WAIT_INST_ANY; its LDS sub-bucket SQ_WAIT_INST_LDS) and issuing
(SQ_ACTIVE_INST_ANY), the LDS-array busy fraction (SQ_LDS_IDX_ACTIVE per CU-cycle) and the bank-conflict rate."""
import csv, glob, json, sys, collections
args = [a for a in sys.argv[1:] if not a.startswith('--match')]
match = None
if '--match' in sys.argv:
    match = sys.argv[sys.argv.index('--match') + 1]
    args = [a for a in args if a != match]
out, dirs = args[0], args[1:]
res = collections.OrderedDict()
for di, d in enumerate(dirs):
    cf = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
    kt = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)
    if not cf:
        continue
    dur = {}
    if kt:
        for r in csv.DictReader(open(kt[0])):
            dur[r['Dispatch_Id']] = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    seen = collections.defaultdict(set)
    for r in csv.DictReader(open(cf[0])):
        n = r['Kernel_Name']
        if match and match not in n:
            continue
        e = res.setdefault(n, {'counters': collections.defaultdict(float), 'disp': collections.defaultdict(int), 'ns': 0.0, 'ns_n': 0,
                               'vgpr': r.get('VGPR_Count') or r.get('Arch_VGPR_Count'), 'lds': r.get('LDS_Block_Size'), 'grid': r.get('Grid_Size'), 'wg': r.get('Workgroup_Size')})
        e['counters'][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Dispatch_Id'] not in seen[(n, r['Counter_Name'])]:
            seen[(n, r['Counter_Name'])].add(r['Dispatch_Id'])
            e['disp'][r['Counter_Name']] += 1
        if di == 0 and r['Dispatch_Id'] not in seen[(n, '_d')]:
            seen[(n, '_d')].add(r['Dispatch_Id'])
            e['ns'] += dur.get(r['Dispatch_Id'], 0); e['ns_n'] += 1
final = {}
for n, e in res.items():
    c = {k: v / max(e['disp'][k], 1) for k, v in e['counters'].items()}
    g = c.get('GRBM_GUI_ACTIVE', 0.0) / 8.0                      # (the counter sums the 8 XCDs)
    wc = c.get('SQ_WAVE_CYCLES', 0.0)
    d = {'dispatches': max(e['disp'].values()), 'mean_us_profiled': round(e['ns'] / max(e['ns_n'], 1) / 1e3, 2), 'grid': e['grid'], 'workgroup': e['wg'], 'vgpr': e['vgpr'], 'lds_bytes': e['lds'],
         'counters_per_dispatch': {k: round(v, 1) for k, v in sorted(c.items())}}
    der = {}
    if g > 0:
        der['clock_GHz_from_GUI_ACTIVE'] = round(g / max(e['ns'] / max(e['ns_n'], 1), 1.0), 3)
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in c:
            der['mfma_busy_frac'] = round(c['SQ_VALU_MFMA_BUSY_CYCLES'] / (g * 1024), 4)
        if 'SQ_BUSY_CYCLES' in c:
            der['sq_busy_over_gui'] = round(c['SQ_BUSY_CYCLES'] / max(c.get('GRBM_GUI_ACTIVE', 1.0), 1.0), 4)
        if 'SQ_LDS_IDX_ACTIVE' in c:
            der['lds_array_busy_frac_per_cu'] = round(c['SQ_LDS_IDX_ACTIVE'] / (g * 256), 4)
    if wc > 0:
        for k, nm in (('SQ_WAIT_ANY', 'wave_parked_frac'), ('SQ_WAIT_INST_ANY', 'issue_stall_frac'), ('SQ_WAIT_INST_LDS', 'issue_stall_lds_frac'),
                      ('SQ_ACTIVE_INST_ANY', 'issuing_frac'), ('SQ_ACTIVE_INST_LDS', 'issuing_lds_frac'), ('SQ_ACTIVE_INST_VALU', 'issuing_valu_frac'),
                      ('SQ_INST_CYCLES_VMEM', 'vmem_inst_cycles_frac'), ('SQ_ACTIVE_INST_VMEM', 'issuing_vmem_frac')):
            if k in c:
                der[nm] = round(c[k] / wc, 4)
    if 'SQ_LDS_BANK_CONFLICT' in c and c.get('SQ_LDS_IDX_ACTIVE', 0) > 0:
        der['lds_bank_conflict_rate'] = round(c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE'], 4)
    if 'SQ_INSTS_VALU_MFMA' in c and 'SQ_VALU_MFMA_BUSY_CYCLES' in c and c['SQ_INSTS_VALU_MFMA'] > 0:
        der['mfma_busy_cycles_per_mfma_inst'] = round(c['SQ_VALU_MFMA_BUSY_CYCLES'] / c['SQ_INSTS_VALU_MFMA'], 2)
    d['derived'] = der
    final[n] = d
json.dump({'note': 'mean per dispatch over the passes listed; SQ_* wave counters are in quad-cycles summed over waves (MI355X_MICROARCH.md), '
                   'SQ_VALU_MFMA_BUSY_CYCLES in cycles summed over SIMDs; GRBM_GUI_ACTIVE summed over the 8 XCDs', 'passes': dirs, 'kernels': final}, open(out, 'w'), indent=1)
for n, d in final.items():
    print(n[:110]); print('   ', d['dispatches'], 'dispatches', d['mean_us_profiled'], 'us', d['derived'])
