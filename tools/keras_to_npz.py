"""Weight interchange with a real tf.keras model.  RUN THIS ON A HOST THAT HAS TENSORFLOW -- it is not
importable on the MI355X image (no TensorFlow / h5py there), so this script is documentation-grade: it was written against
the Keras API and is not exercised by the test-suite.

    python tools/keras_to_npz.py  model.h5|weights.hdf5  out.npz  [--build unet --nclasses 2 --nchannels 4]
    python tools/keras_to_npz.py  --reverse in.npz  model.h5  out.h5

Forward: every variable of every layer is stored under '<layer.name>/<kernel|bias|gamma|beta|moving_mean|moving_variance>'
-- exactly the names satellite_computervision_amd.Model.load_weights(path, by_name=True, skip_mismatch=True) resolves
(the build's layer names follow Keras' auto-naming: conv2d, conv2d_1, batch_normalization, conv2d_transpose, probs, ...;
`moving_variance` is accepted as an alias of `moving_var`).  Kernel layouts need no conversion: the build keeps Keras'
HWIO conv kernels and (kh, kw, Cout, Cin) transposed-conv kernels.
Reverse: writes the arrays of an .npz saved by the build (`Model.save`) back into a Keras model (by layer name) and saves
it as HDF5, so that utils/model_tools.retrain_model / get_blob_model of the reference can pick it up.
"""
import argparse
import sys

import numpy as np


def keras_to_npz(src, dst, build=None, nclasses=2, nchannels=4):
    import tensorflow as tf
    try:
        model = tf.keras.models.load_model(src, compile=False)
    except Exception:
        if build != 'unet':
            raise
        sys.path.insert(0, '.')
        from utils import model_tools                       # the reference's own builder
        model = model_tools.get_unet_model(nclasses, nchannels)
        model.load_weights(src)
    out = {}
    for layer in model.layers:
        for var in layer.weights:
            short = var.name.split('/')[-1].split(':')[0]
            out[f'{layer.name}/{short}'] = var.numpy()
    np.savez(dst, **out)
    print(f'wrote {len(out)} arrays to {dst}')


def npz_to_keras(npz, model_path, dst):
    import tensorflow as tf
    model = tf.keras.models.load_model(model_path, compile=False)
    z = np.load(npz)
    alias = {'moving_var': 'moving_variance'}
    for layer in model.layers:
        vals = []
        for var in layer.weights:
            short = var.name.split('/')[-1].split(':')[0]
            keys = [f'{layer.name}/{short}'] + [f'{layer.name}/{k}' for k, v in alias.items() if v == short]
            hit = next((k for k in keys if k in z.files), None)
            vals.append(z[hit] if hit is not None and z[hit].shape == tuple(var.shape) else var.numpy())
        if vals:
            layer.set_weights(vals)
    model.save(dst)
    print(f'wrote {dst}')


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('paths', nargs='+')
    ap.add_argument('--reverse', action='store_true')
    ap.add_argument('--build', default=None)
    ap.add_argument('--nclasses', type=int, default=2)
    ap.add_argument('--nchannels', type=int, default=4)
    a = ap.parse_args()
    if a.reverse:
        npz_to_keras(*a.paths[:3])
    else:
        keras_to_npz(a.paths[0], a.paths[1], a.build, a.nclasses, a.nchannels)
