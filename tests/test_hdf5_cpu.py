"""hdf5_io (pure-Python HDF5 reader for Keras weight files) against files written by the REAL HDF5 library
(tests/golden/make_h5_fixtures.py, run with the side interpreter that has h5py)."""
import os
import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def test_keras_weight_files_read_back_exactly():
    from satellite_computervision_amd import hdf5_io as H
    z = np.load(os.path.join(GOLD, 'keras_unet_weights.npz'))
    keys = sorted(z.files)
    for fname in ('keras_unet_weights.h5', 'keras_unet_model.h5'):
        path = os.path.join(GOLD, fname)
        assert H.is_hdf5(path) and not H.is_hdf5(os.path.join(GOLD, 'keras_unet_weights.npz'))
        layers = H.read_keras_weights(path)
        assert [l for l, _ in layers][:3] == ['input_1', 'encoder_0', 'encoder_1'] and len(layers) == 26
        flat = [(ln, wn, a) for ln, ws in layers for wn, a in ws]
        assert len(flat) == len(keys) == 56
        for k, (ln, wn, a) in zip(keys, flat):
            _, kl, kw = k.split('|')
            assert (kl, kw) == (ln, wn) and a.dtype == np.float32 and np.array_equal(a, z[k])
    with H.File(os.path.join(GOLD, 'keras_unet_model.h5')) as f:
        assert f.keys() == ['extras', 'model_weights', 'optimizer_weights']
        assert f.attrs['backend'] == 'tensorflow' and f.attrs['keras_version'] == '2.6.0'              # variable-length strings (global heap)
        assert '"class_name": "Functional"' in f.attrs['model_config']
        assert f['optimizer_weights/Adam/iter:0'].read() == 1234 and f['optimizer_weights/Adam/iter:0'].dtype == np.int64   # scalar dataset
        assert np.array_equal(f['extras/chunked'].read(), np.arange(6000, dtype=np.float32).reshape(60, 100))              # chunked + gzip
        assert np.allclose(f['extras/f64'].read(), np.linspace(0, 1, 7)) and f['extras'].attrs['note'] == 'variable-length utf-8 é'
        assert 'model_weights/probs/probs/kernel:0' in f and 'model_weights/nope' not in f
        with pytest.raises(KeyError):
            f['model_weights/probs/missing']


def test_structures_of_larger_files():
    """300 links in one group (multi-level B-tree), header continuation blocks, chunked attributes, other numeric types, a file
    written with libver='latest'."""
    from satellite_computervision_amd import hdf5_io as H
    z = np.load(os.path.join(GOLD, 'h5_stress_expected.npz'))
    with H.File(os.path.join(GOLD, 'h5_stress.h5')) as f:
        assert f.keys() == sorted([f'layer_{i:03d}' for i in range(300)] + ['many_attrs', 'types'])
        for k in z.files:
            path = k.replace('|', '/')
            if path.startswith('latest/'):
                continue
            a = f[path].read()
            assert a.shape == z[k].shape and np.array_equal(a, z[k]), path
        at = f['many_attrs'].attrs
        assert all(np.array_equal(at[f'attr_{i:02d}'], np.arange(i + 1) * 3) for i in range(40))
        assert at['fixed'] == b'fixed-length' and at['f64'] == 2.5
    layers = H.read_keras_weights(os.path.join(GOLD, 'h5_stress.h5'))             # layer_names0 + layer_names1
    assert [l for l, _ in layers] == [f'layer_{i:03d}' for i in range(300)]
    assert [l for l, ws in layers if ws] == [f'layer_{i:03d}' for i in range(0, 300, 50)]
    latest = H.read_keras_weights(os.path.join(GOLD, 'h5_latest.h5'))
    assert [(l, [w for w, _ in ws]) for l, ws in latest] == [('dense', ['dense/kernel:0', 'dense/bias:0']), ('act', [])]
    assert np.array_equal(latest[0][1][0][1], z['latest|dense|dense|kernel:0']) and np.array_equal(latest[0][1][1][1], z['latest|dense|dense|bias:0'])
    with pytest.raises(ValueError):
        H.File(os.path.join(GOLD, 'h5_stress_expected.npz'))


def test_unet_arguments_recovered_from_a_keras_file():
    from satellite_computervision_amd import hdf5_io as H
    from satellite_computervision_amd import model_tools as mt
    layers = H.read_keras_weights(os.path.join(GOLD, 'keras_unet_model.h5'))
    assert mt.unet_config_from_keras_weights(layers) == dict(nclasses=2, nchannels=4, filters=[16, 32], factors=[2, 2])
    # the graph built from them has the same variables, in the same order and shapes, as the file
    mt.reset_uids()
    m = mt.get_unet_model(2, 4, filters=[16, 32], factors=[2, 2])
    flat = [(wn, a.shape) for _, ws in layers for wn, a in ws]
    assert [tuple(ps.shape) for ps in m.param_specs] == [tuple(sh) for _, sh in flat]
    for ps, (wn, _) in zip(m.param_specs, flat):                  # innermost '<layer>/<variable>' scopes are Keras' automatic names
        want = '/'.join(wn.split(':')[0].split('/')[-2:]).replace('moving_variance', 'moving_var')
        assert ps.name == want, (ps.name, wn)


H5PY_PYTHON = '/opt/conda/bin/python3.9'         # side interpreter of the image that has the real h5py / HDF5 library


def _h5py_available():
    import subprocess
    try:
        return subprocess.run([H5PY_PYTHON, '-c', 'import h5py'], capture_output=True, timeout=60).returncode == 0
    except (OSError, subprocess.SubprocessError):
        return False


def test_written_files_round_trip_and_are_read_by_the_real_library(tmp_path):
    """write_keras_weights: Keras layout written in pure Python, re-read by hdf5_io and -- when the image's h5py interpreter is
    present -- by the real HDF5 library (every dataset and attribute); 1500 layers exercise several symbol-table nodes per group,
    a larger B-tree fan-out and Keras' attribute chunking."""
    import json, subprocess
    from satellite_computervision_amd import hdf5_io as H
    rng = np.random.default_rng(0)
    layers = H.read_keras_weights(os.path.join(GOLD, 'keras_unet_model.h5'))
    small = str(tmp_path / 'small.h5')
    H.write_keras_weights(small, layers, root_attrs={'model_config': json.dumps({'k': [1, 2]}), 'vec': np.arange(5, dtype=np.float32)},
                          model_weights_group=True, extra_groups={'optimizer_weights': [('Adam/iter:0', np.array(12, dtype=np.int64)),
                                                                                        ('Adam/m:0', np.ones((2, 3), np.float32))]})
    back = H.read_keras_weights(small)
    assert [(l, [n for n, _ in w]) for l, w in back] == [(l, [n for n, _ in w]) for l, w in layers]
    assert all(np.array_equal(a, b) for (_, w1), (_, w2) in zip(layers, back) for (_, a), (_, b) in zip(w1, w2))
    with H.File(small) as f:
        assert json.loads(f.attrs['model_config']) == {'k': [1, 2]} and np.array_equal(f.attrs['vec'], np.arange(5))
        assert f['optimizer_weights/Adam/iter:0'].read() == 12 and f['optimizer_weights/Adam/iter:0'].shape == ()
    name = 'layer_with_a_rather_long_name_to_fill_the_attribute_{:05d}'
    many = [(name.format(i), [(name.format(i) + '/kernel:0', rng.standard_normal((2, 3)).astype(np.float32))] if i % 7 == 0 else []) for i in range(1500)]
    big = str(tmp_path / 'many.h5')
    H.write_keras_weights(big, many, extra_groups={'optimizer_weights': [('f64', np.linspace(0, 1, 5))]})
    back = H.read_keras_weights(big)
    assert [l for l, _ in back] == [l for l, _ in many] and all(np.array_equal(a, b) for (_, w1), (_, w2) in zip(many, back) for (_, a), (_, b) in zip(w1, w2))
    if not _h5py_available():
        pytest.skip('no interpreter with h5py in this image: checked against hdf5_io only')
    code = r'''
import h5py, numpy as np, json, sys
small, big, out = sys.argv[1:4]
res = {}
with h5py.File(small, 'r') as f:
    g = f['model_weights']
    names = [n.decode() for n in g.attrs['layer_names']]
    res['names'] = names
    res['backend'] = g.attrs['backend'].decode() if isinstance(g.attrs['backend'], bytes) else str(g.attrs['backend'])
    res['cfg'] = json.loads(f.attrs['model_config'])
    res['iter'] = int(f['optimizer_weights/Adam/iter:0'][()])
    arrs = [g[n][w.decode()][()] for n in names for w in g[n].attrs['weight_names']]
    np.savez(out, *arrs)
with h5py.File(big, 'r') as f:
    names, i = [], 0
    while f'layer_names{i}' in f.attrs:
        names += [n.decode() for n in f.attrs[f'layer_names{i}']]; i += 1
    res['chunks'], res['n_big'] = i, len(names)
    res['sum_big'] = float(sum(f[n][w.decode()][()].sum() for n in names for w in f[n].attrs['weight_names']))
    res['f64'] = f['optimizer_weights/f64'][()].tolist()
print(json.dumps(res))
'''
    out = str(tmp_path / 'h5py_read.npz')
    r = subprocess.run([H5PY_PYTHON, '-c', code, small, big, out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads(r.stdout.strip().splitlines()[-1])
    assert res['names'] == [l for l, _ in layers] and res['backend'] == 'tensorflow' and res['cfg'] == {'k': [1, 2]} and res['iter'] == 12
    z = np.load(out)
    flat = [a for _, w in layers for _, a in w]
    assert len(z.files) == len(flat) and all(np.array_equal(z[f'arr_{i}'], a) for i, a in enumerate(flat))
    assert res['chunks'] >= 2 and res['n_big'] == 1500 and res['f64'] == np.linspace(0, 1, 5).tolist()
    assert abs(res['sum_big'] - float(sum(a.sum() for _, w in many for _, a in w))) < 1e-3
