"""Generate golden vectors by RUNNING the reference's own Python in the build container.

Run here only (needs /root/reference, which does not exist on the GPU box):
    python tests/golden/make_reference_fixtures.py
Writes tests/golden/tiling_reference.npz and tests/golden/array_tools_reference.npz
(inputs + expected outputs only -- no reference source travels).

* utils/prediction_tools.py cannot be imported (tensorflow/rasterio missing), so
  the bodies of generate_chip_indices / extract_chips / predict_chips (:87-156)
  are AST-extracted and executed against NumPy with a fake model
  predict(x) = 2 * x[..., :1].
* utils/array_tools.py imports cleanly (NumPy only) and is called directly.
"""
import ast
import io
import contextlib
import os
import sys
import numpy as np

REF = '/root/reference'
OUT = os.path.dirname(os.path.abspath(__file__))


def extract_functions(path, names):
    src = open(path).read()
    tree = ast.parse(src)
    ns = {'np': np}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            code = compile(ast.Module(body=[node], type_ignores=[]), path, 'exec')
            exec(code, ns)
    return ns


class FakeModel:
    def predict(self, x, verbose=0):
        return 2.0 * x[..., :1]


def timeseries_fixtures():
    """utils/processing.py:185-223 (normalize_timeseries, rearrange_timeseries, sin_cos) and utils/array_tools.py:12-24 (make_harmonics):
    the reference's own bodies executed with seeded `random` -> tests/golden/timeseries_reference.npz"""
    import math
    import random
    pr = extract_functions(f'{REF}/utils/processing.py', {'normalize_timeseries', 'rearrange_timeseries', 'sin_cos'})
    pr['randint'] = random.randint
    pr['math'] = math
    sys.path.insert(0, f'{REF}/utils')
    import array_tools
    rng = np.random.default_rng(21)
    arr = rng.random((3, 6, 5, 4, 7)) * 9000
    arr[0, 2, 1, 1, 3] = np.nan
    out = {'arr': arr, 'normalized': pr['normalize_timeseries'](arr, axis=1), 'normalized_s1': pr['normalize_timeseries'](arr, maxval=-50.0, axis=1)}
    for seed in (0, 1, 2, 3):
        random.seed(seed)
        with contextlib.redirect_stdout(io.StringIO()):
            f, l, st = pr['rearrange_timeseries'](out['normalized'], 4)
        out[f'feats_{seed}'], out[f'labels_{seed}'], out[f'start_{seed}'] = f, l, np.int64(st)
    out['harmonics'] = array_tools.make_harmonics(np.array([3, 10, 17]), 6, (4, 5))
    out['sin_cos'] = np.array([pr['sin_cos'](t, 6) for t in range(8)])
    np.savez_compressed(os.path.join(OUT, 'timeseries_reference.npz'), **out)


def main():
    timeseries_fixtures()
    pt = extract_functions(f'{REF}/utils/prediction_tools.py',
                           {'generate_chip_indices', 'extract_chips', 'predict_chips'})
    rt = extract_functions(f'{REF}/utils/raster_tools.py', {'generate_chip_indices'})
    out = {}
    shapes = [(1024, 1024, 4), (384, 384, 4), (640, 640, 4), (1000, 1300, 4), (900, 700, 3), (2048, 1500, 2)]
    cfgs = [(128, 256), (64, 128), (0, 256), (32, 96)]
    cases = []
    for s in shapes:
        for buff, kernel in cfgs:
            idx = pt['generate_chip_indices'](np.zeros(s, np.float32), buff, kernel)
            key = f'idx_{s[0]}_{s[1]}_{s[2]}_{buff}_{kernel}'
            out[key] = np.asarray(idx, np.int64).reshape(-1, 2)
            cases.append(key)
    out['cases'] = np.asarray(cases)
    for (h, w, b, k) in [(1024, 1024, 128, 256), (1000, 1300, 128, 256), (700, 900, 64, 128)]:
        out[f'rt_idx_{h}_{w}_{b}_{k}'] = np.asarray(rt['generate_chip_indices'](h, w, b, k), np.int64).reshape(-1, 2)

    rng = np.random.default_rng(7)
    arr = rng.integers(0, 255, (160, 224, 3)).astype(np.float32)
    out['pc_arr'] = arr
    idx = pt['generate_chip_indices'](arr, 16, 32)
    out['pc_idx'] = np.asarray(idx, np.int64)
    with contextlib.redirect_stdout(io.StringIO()):
        tmpl = pt['predict_chips'](arr, idx, np.zeros(arr.shape[:2]), FakeModel(), 32, 16)
    out['pc_template'] = tmpl
    sq = rng.integers(0, 255, (128, 128, 2)).astype(np.float32)
    chips = pt['extract_chips'](sq, 16, 32)
    out['ec_arr'] = sq
    out['ec_chips'] = np.stack(chips)

    # make_array_predictions (utils/prediction_tools.py:293-373) only needs json + numpy + model.predict: run the real body
    import json, tempfile
    mp = extract_functions(f'{REF}/utils/prediction_tools.py', {'make_array_predictions'})
    mp['json'] = json

    class StackModel:
        def __init__(self, preds):
            self.preds = preds

        def predict(self, dataset, steps=None, verbose=0):
            return self.preds
    rngm = np.random.default_rng(42)
    for tag, (kernel, buff, cols, rows_) in {'a': ((8, 8), (4, 4), 3, 2), 'b': ((6, 10), (4, 2), 2, 3)}.items():
        n = cols * rows_
        preds = rngm.random((n, kernel[0] + buff[0], kernel[1] + buff[1], 2)).astype(np.float32)
        with tempfile.NamedTemporaryFile('w', suffix='.json', delete=False) as f:
            json.dump({'totalPatches': n, 'patchesPerRow': cols}, f)
        with contextlib.redirect_stdout(io.StringIO()):
            mosaic = mp['make_array_predictions'](None, StackModel(preds), f.name, list(kernel), list(buff))
        out[f'mosaic_{tag}_preds'] = preds
        out[f'mosaic_{tag}_cfg'] = np.array([kernel[0], kernel[1], buff[0], buff[1], cols, n])
        out[f'mosaic_{tag}_out'] = mosaic
        os.unlink(f.name)
    # normalize_confusion_matrix (utils/model_tools.py:1111-1126) is NumPy only: run the real body
    ncm = extract_functions(f'{REF}/utils/model_tools.py', {'normalize_confusion_matrix'})
    cm = np.random.default_rng(11).integers(1, 5000, (5, 5)).astype(np.int64)
    out['ncm_in'] = cm
    out['ncm_out'] = ncm['normalize_confusion_matrix'](cm)
    np.savez_compressed(f'{OUT}/tiling_reference.npz', **out)
    sys.path.insert(0, f'{REF}/utils')
    import array_tools as at
    a = {}
    x = rng.random((2, 8, 8, 3)).astype(np.float32)
    a['morph_in'] = x
    for v in (False, True):
        for h in (False, True):
            for r in (0, 1, 2, 3):
                a[f'morph_{int(v)}{int(h)}{r}'] = at.aug_array_morph(x, v, h, r)
    a['merge_out'] = at.merge_classes(np.array([1, 2, 3, 2]), [(2, 9), (3, 7)], np.array([1, 2, 3, 2]))
    img = (rng.random((16, 16, 4)) * 3000).astype(np.float32)
    a['rescale_in'] = img
    a['rescale_moments'] = at.rescale_array(img, moments=[(0, 3000)] * 4)
    a['rescale_axes01'] = at.rescale_array(img, axes=(0, 1))
    a['normalize_axes01'] = at.normalize_array(img, axes=(0, 1))
    # aug_array_color draws two uniform(0.95, 1.05) multipliers from `random`; seed it and record the draws
    import random
    img4 = rng.random((3, 12, 10, 4)).astype(np.float32)
    img4[1, 2, 3, 1] = np.nan                                  # nanmean path
    a['color_in'] = img4
    for seed in (0, 1):
        random.seed(seed)
        a[f'color_out_{seed}'] = at.aug_array_color(img4)
        random.seed(seed)
        a[f'color_mul_{seed}'] = np.array([random.uniform(0.95, 1.05), random.uniform(0.95, 1.05)])
    lab = np.arange(256).reshape(1, 1, 16, 16)
    a['merge_lc_default'] = at.merge_classes(lab, [(12, 3), (11, 3), (10, 3), (9, 8), (255, 0)], lab)
    a['merge_dup_rule'] = at.merge_classes(lab, [(5, 1), (1, 7), (5, 2)], lab)          # rules test the ORIGINAL values; later wins
    np.savez_compressed(f'{OUT}/array_tools_reference.npz', **a)
    # file-list helpers of utils/processing.py:26-114 are pure Python (the module itself needs tensorflow): run the real bodies
    import copy, json
    from pathlib import Path
    fh = extract_functions(f'{REF}/utils/processing.py', {'get_file_id', 'match_files', 'split_files'})
    fh.update(Path=Path, copy=copy)
    rngf = np.random.default_rng(21)
    urls = []
    for var in ('naip', 's2', 'label', 'lidar'):
        for t in range(12):
            if rngf.random() < 0.8:
                urls.append(f'/data/train/{var}/md_2019_{var}_{t // 4:03d}_{t % 4:03d}.npy')
    flat = [f'/blob/train/aoi_2019_{var}_{t:03d}_{(t * 7) % 5:03d}.npy' for var in ('naip', 's2', 'label') for t in range(9) if (t + len(var)) % 4]
    rngf.shuffle(urls)
    cases = {
        'ids': [[u, list(fh['get_file_id'](u))] for u in urls[:6]] + [['a-b-c-d-e-f.npy', list(fh['get_file_id']('a-b-c-d-e-f.npy', '-', slice(1, 4)))]],
        'match': fh['match_files'](urls, {'naip': {'files': []}, 's2': {'files': []}, 'label': {'files': []}, 'lidar': {'files': None}}),
        'match_subset': fh['match_files'](urls, {'naip': {'files': [], 'bands': 4}, 'label': {'files': []}}, subset={('000', '001'), ('001', '002'), ('002', '003')}),
        'match_flat': fh['match_files'](flat, {'naip': {'files': []}, 's2': {'files': []}, 'label': {'files': []}}, parts=slice(3, 5), flatdirectory=True),
        'split': fh['split_files'](urls, labels=['label', 'naip', 's2']),
        'urls': urls, 'flat': flat,
    }
    with open(f'{OUT}/file_helpers_reference.json', 'w') as f:
        json.dump(cases, f, indent=1, sort_keys=True)
    print('wrote fixtures:', len(out), len(a))


if __name__ == '__main__':
    main()
