"""Write Keras-format HDF5 weight files with the REAL HDF5 library, as fixtures for satellite_computervision_amd/hdf5_io.py.

Run in the build container with the side interpreter that has h5py (the main one does not):
    /opt/conda/bin/python3.9 tests/golden/make_h5_fixtures.py
Writes tests/golden/keras_unet_weights.h5 (layout of tf.keras `Model.save_weights('x.h5')`: root attributes layer_names / backend /
keras_version, one group per layer with a weight_names attribute, datasets at <layer>/<weight name>), keras_unet_model.h5 (layout
of `Model.save('x.h5')`: the same tree under /model_weights plus model_config / training_config attributes and an
/optimizer_weights group) and keras_unet_weights.npz (the same arrays, flat, in file order) to compare against.

The layer / variable names follow what tf.keras generates for the reference's get_unet_model(2, 4, filters=[16, 32],
factors=[2, 2]) (utils/model_tools.py:394-415): custom layers (encoder_{i} > conv_block > conv_batch_act) own nested variable
scopes, the layers of decoder_block (Conv2DTranspose, Concatenate, BatchNormalization, Activation, Conv2D) are top-level functional layers,
weightless layers are listed with an empty weight_names attribute.  Values are seeded random numbers: the files pin the CONTAINER
format, not a trained model.
"""
import json
import os
import numpy as np
import h5py

OUT = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(2024)


def cba(prefix, cin, cout, conv, bn, k=3):
    return [(f'{prefix}/{conv}/kernel:0', (k, k, cin, cout)), (f'{prefix}/{conv}/bias:0', (cout,)),
            (f'{prefix}/{bn}/gamma:0', (cout,)), (f'{prefix}/{bn}/beta:0', (cout,)),
            (f'{prefix}/{bn}/moving_mean:0', (cout,)), (f'{prefix}/{bn}/moving_variance:0', (cout,))]


def layers():
    L = [('input_1', [])]
    # explicit names of the reference: encoder_block(..., name=f'encoder_{i}') (:348), conv_block / conv_batch_act defaults (:213, :176)
    L.append(('encoder_0', cba('encoder_0/conv_block/conv_batch_act', 4, 16, 'conv2d', 'batch_normalization')))
    L.append(('encoder_1', cba('encoder_1/conv_block/conv_batch_act', 16, 32, 'conv2d_2', 'batch_normalization_2')))
    L.append(('conv_block', cba('conv_block/conv_batch_act', 32, 64, 'conv2d_4', 'batch_normalization_4')))
    n_bn, n_conv, n_act = 6, 6, 0
    bnw = lambda bn, c: (bn, [(f'{bn}/{v}:0', (c,)) for v in ('gamma', 'beta', 'moving_mean', 'moving_variance')])
    act = lambda i: ('activation' + ('' if i == 0 else f'_{i}'), [])
    for j, (cin, f) in enumerate(((64, 32), (32, 16))):       # decoder_block (:306-317): plain functional layers, no custom blocks
        t = 'conv2d_transpose' + ('' if j == 0 else f'_{j}')
        L.append((t, [(f'{t}/kernel:0', (2, 2, f, cin)), (f'{t}/bias:0', (f,))]))
        L.append(('concatenate' + ('' if j == 0 else f'_{j}'), []))
        L.append(bnw(f'batch_normalization_{n_bn}', 2 * f)); n_bn += 1
        L.append(act(n_act)); n_act += 1
        for r in range(2):
            c = f'conv2d_{n_conv}'; n_conv += 1
            L.append((c, [(f'{c}/kernel:0', (3, 3, 2 * f if r == 0 else f, f)), (f'{c}/bias:0', (f,))]))
            L.append(bnw(f'batch_normalization_{n_bn}', f)); n_bn += 1
            L.append(act(n_act)); n_act += 1
    L.append(('probs', [('probs/kernel:0', (1, 1, 16, 2)), ('probs/bias:0', (2,))]))
    L.append(('classes', []))
    return L


def write_tree(g, L, values):
    g.attrs['layer_names'] = np.array([n.encode('utf8') for n, _ in L])
    g.attrs['backend'] = b'tensorflow'
    g.attrs['keras_version'] = b'2.6.0'
    for name, ws in L:
        lg = g.create_group(name)
        lg.attrs['weight_names'] = np.array([w.encode('utf8') for w, _ in ws]) if ws else np.array([], dtype='S1')
        for w, shape in ws:
            d = lg.create_dataset(w, shape, dtype='float32')
            if shape:
                d[:] = values[w]
            else:
                d[()] = values[w]


def main():
    L = layers()
    values, flat = {}, {}
    for i, (name, ws) in enumerate(L):
        for w, shape in ws:
            v = rng.standard_normal(shape).astype(np.float32)
            if 'moving_variance' in w:
                v = np.abs(v) + 0.5
            values[w] = v
            flat[f'{len(flat):03d}|{name}|{w}'] = v
    with h5py.File(f'{OUT}/keras_unet_weights.h5', 'w') as f:
        write_tree(f, L, values)
    with h5py.File(f'{OUT}/keras_unet_model.h5', 'w') as f:
        f.attrs['keras_version'] = b'2.6.0'
        f.attrs['backend'] = b'tensorflow'
        f.attrs['model_config'] = json.dumps({'class_name': 'Functional', 'config': {'name': 'model', 'layers': [{'name': n} for n, _ in L]}}).encode('utf8')
        f.attrs['training_config'] = json.dumps({'loss': None, 'optimizer_config': {'class_name': 'Adam', 'config': {'learning_rate': 0.0009}}}).encode('utf8')
        write_tree(f.create_group('model_weights'), L, values)
        og = f.create_group('optimizer_weights')
        og.attrs['weight_names'] = np.array([b'Adam/iter:0', b'Adam/probs/kernel/m:0'])
        og.create_dataset('Adam/iter:0', (), dtype='int64')[()] = 1234
        og.create_dataset('Adam/probs/kernel/m:0', (1, 1, 16, 2), dtype='float32')[:] = 0.25
        # things a reader must tolerate: a chunked + compressed dataset, a float64 one, a variable-length string attribute
        f.create_dataset('extras/chunked', data=np.arange(6000, dtype=np.float32).reshape(60, 100), chunks=(16, 32), compression='gzip')
        f.create_dataset('extras/f64', data=np.linspace(0, 1, 7))
        f['extras'].attrs['note'] = 'variable-length utf-8 é'
    np.savez_compressed(f'{OUT}/keras_unet_weights.npz', **flat)
    print('layers', len(L), 'arrays', len(flat), 'params', sum(v.size for v in values.values()))
    stress()


def stress():
    """Structures a reader meets in larger files: hundreds of links in one group (multi-level group B-tree), object-header
    continuation blocks (many attributes), Keras' chunked attributes (layer_names0, layer_names1, ...), big-endian / 16-bit / integer
    types, fixed-length scalar strings (h5py 2.x wrote bytes that way), an empty dataset, and a file written with libver='latest'
    (version-2 object headers, compact link messages)."""
    r = np.random.default_rng(7)
    expect = {}
    with h5py.File(f'{OUT}/h5_stress.h5', 'w') as f:
        names = [f'layer_{i:03d}' for i in range(300)]
        half = len(names) // 2
        f.attrs['layer_names0'] = np.array([n.encode() for n in names[:half]])
        f.attrs['layer_names1'] = np.array([n.encode() for n in names[half:]])
        for i, n in enumerate(names):
            g = f.create_group(n)
            if i % 50 == 0:
                w = r.standard_normal((3, 5)).astype(np.float32)
                g.attrs['weight_names'] = np.array([f'{n}/w:0'.encode()])
                g.create_dataset(f'{n}/w:0', data=w)
                expect[f'{n}/{n}/w:0'] = w
            else:
                g.attrs['weight_names'] = np.array([], dtype='S1')
        many = f.create_group('many_attrs')
        for i in range(40):
            many.attrs[f'attr_{i:02d}'] = np.arange(i + 1, dtype=np.int32) * 3
        many.attrs['fixed'] = np.bytes_(b'fixed-length')
        many.attrs['f64'] = np.float64(2.5)
        t = f.create_group('types')
        for key, arr in (('be_i16', np.arange(-5, 5, dtype='>i2')), ('f16', np.linspace(-1, 1, 9).astype(np.float16)), ('u8', np.arange(7, dtype=np.uint8)),
                         ('i64', np.array([[1, -2], [3, 2 ** 40]], dtype=np.int64)), ('be_f64', np.linspace(0, 1, 5).astype('>f8')),
                         ('empty', np.zeros((0, 4), np.float32)), ('strings', np.array([b'ab', b'cdef', b''], dtype='S4'))):
            t.create_dataset(key, data=arr)
            expect['types/' + key] = arr.astype(arr.dtype.newbyteorder('=')) if arr.dtype.kind in 'iuf' else arr
        t.create_dataset('never_written', (4, 3), dtype='float32')
        expect['types/never_written'] = np.zeros((4, 3), np.float32)
    with h5py.File(f'{OUT}/h5_latest.h5', 'w', libver='latest') as f:
        f.attrs['layer_names'] = np.array([b'dense', b'act'])
        g = f.create_group('dense')
        g.attrs['weight_names'] = np.array([b'dense/kernel:0', b'dense/bias:0'])
        k, b = r.standard_normal((6, 3)).astype(np.float32), r.standard_normal(3).astype(np.float32)
        g.create_dataset('dense/kernel:0', data=k); g.create_dataset('dense/bias:0', data=b)
        f.create_group('act').attrs['weight_names'] = np.array([], dtype='S1')
        expect['latest/dense/dense/kernel:0'], expect['latest/dense/dense/bias:0'] = k, b
    np.savez_compressed(f'{OUT}/h5_stress_expected.npz', **{k.replace('/', '|'): v for k, v in expect.items()})
    print('stress entries', len(expect))


if __name__ == '__main__':
    main()
