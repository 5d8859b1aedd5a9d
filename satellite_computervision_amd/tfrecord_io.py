"""Earth-Engine TFRecord tile IO around the prediction path, without TensorFlow.

On-disk format (what `tf.data.TFRecordDataset(files, compression_type='GZIP')` reads and `tf.io.TFRecordWriter` writes,
utils/prediction_tools.py:221, 404; utils/processing.py:416):
  record  = uint64 length | uint32 masked_crc32c(length) | payload | uint32 masked_crc32c(payload)      (little endian)
  masked  = ((crc >> 15) | (crc << 17)) + 0xa282ead8   (mod 2^32)
  payload = serialized tf.train.Example: Example{1: Features{1: map<string, Feature>}},
            Feature{1: BytesList{1: bytes}, 2: FloatList{1: packed float32}, 3: Int64List{1: packed varint}}
The whole file may be GZIP-compressed.  The protobuf wire format is hand-parsed (no generated classes needed); the CRC
runs in libsatcv (`satcv_crc32c`).  Mirrors of the reference's callers:
  make_pred_dataset            utils/prediction_tools.py:159-226   (FixedLenFeature float32 [H+buf, W+buf] per band)
  make_array_predictions       utils/prediction_tools.py:293-373   (mixer.json mosaic; pinned by a fixture of the real body)
  write_tfrecord_predictions   utils/prediction_tools.py:375-445   (b1..bC float lists of the cropped patch)
"""
import gzip
import os
import json
import struct
from os.path import join

import numpy as np

from ._lib import lib

_MASK_DELTA = 0xa282ead8


def crc32c(data, crc=0):
    buf = bytes(data)
    return int(lib.satcv_crc32c(buf, len(buf), crc))


def masked_crc(data):
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + _MASK_DELTA) & 0xffffffff


# ------------------------------------------------------------------ record framing
def _open(path, mode):
    if 'r' in mode:
        with open(path, 'rb') as f:
            magic = f.read(2)
        return gzip.open(path, 'rb') if magic == b'\x1f\x8b' else open(path, 'rb')
    return open(path, mode)


def read_records(path, check_crc=True):
    """Yield the payload bytes of every record of a (possibly GZIP-compressed) TFRecord file."""
    with _open(path, 'rb') as f:
        while True:
            head = f.read(12)
            if not head:
                return
            if len(head) != 12:
                raise IOError(f'{path}: truncated record header')
            (length,), (lcrc,) = struct.unpack('<Q', head[:8]), struct.unpack('<I', head[8:])
            if check_crc and masked_crc(head[:8]) != lcrc:
                raise IOError(f'{path}: corrupt record length')
            data = f.read(length)
            tail = f.read(4)
            if len(data) != length or len(tail) != 4:
                raise IOError(f'{path}: truncated record')
            if check_crc and masked_crc(data) != struct.unpack('<I', tail)[0]:
                raise IOError(f'{path}: corrupt record payload')
            yield data


class TFRecordWriter:
    """tf.io.TFRecordWriter(path) semantics (uncompressed unless compression='GZIP')."""

    def __init__(self, path, compression=None):
        self._f = gzip.open(path, 'wb') if compression == 'GZIP' else open(path, 'wb')

    def write(self, payload):
        head = struct.pack('<Q', len(payload))
        self._f.write(head + struct.pack('<I', masked_crc(head)) + payload + struct.pack('<I', masked_crc(payload)))

    def close(self):
        self._f.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


# ------------------------------------------------------------------ tf.train.Example wire format
def _varint(n):
    out = bytearray()
    n &= (1 << 64) - 1
    while True:
        b = n & 0x7f
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _read_varint(buf, pos):
    shift = val = 0
    while True:
        b = buf[pos]
        pos += 1
        val |= (b & 0x7f) << shift
        if not b & 0x80:
            return val, pos
        shift += 7


def _ld(field, payload):
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def encode_example(features):
    """{name: ndarray} -> serialized tf.train.Example.  float arrays -> FloatList, integer arrays -> Int64List, bytes -> BytesList."""
    entries = b''
    for key in features:
        v = features[key]
        if isinstance(v, (bytes, bytearray)):
            feat = _ld(1, _ld(1, bytes(v)))
        else:
            a = np.asarray(v)
            if a.dtype.kind == 'f':
                feat = _ld(2, _ld(1, np.ascontiguousarray(a, '<f4').tobytes()))
            else:
                feat = _ld(3, _ld(1, b''.join(_varint(int(x)) for x in a.reshape(-1))))
        entries += _ld(1, _ld(1, key.encode()) + _ld(2, feat))
    return _ld(1, entries)


# ------------------------------------------------------------------ TensorBoard event files (tensorflow.Event wire format)
def encode_event(wall_time, step=0, scalars=None, file_version=None):
    """Serialized tensorflow.Event: wall_time (1, double), step (2, int64), file_version (3) or summary (5) holding one
    Summary.Value {tag (1), simple_value (2, float)} per scalar -- what tf.summary.scalar / the Keras TensorBoard callback write."""
    ev = struct.pack('<Bd', (1 << 3) | 1, float(wall_time)) + _varint((2 << 3) | 0) + _varint(int(step))
    if file_version is not None:
        ev += _ld(3, file_version.encode())
    if scalars:
        vals = b''.join(_ld(1, _ld(1, tag.encode()) + struct.pack('<Bf', (2 << 3) | 5, float(v))) for tag, v in scalars.items())
        ev += _ld(5, vals)
    return ev


def decode_event(buf):
    """Inverse of encode_event -> dict(wall_time, step, file_version, scalars)."""
    out = dict(wall_time=None, step=0, file_version=None, scalars={})
    pos = 0
    while pos < len(buf):
        tag, pos = _read_varint(buf, pos)
        f, wt = tag >> 3, tag & 7
        if wt == 1:
            (v,) = struct.unpack('<d', buf[pos:pos + 8]); pos += 8
            if f == 1:
                out['wall_time'] = v
        elif wt == 0:
            v, pos = _read_varint(buf, pos)
            if f == 2:
                out['step'] = v
        elif wt == 2:
            n, pos = _read_varint(buf, pos)
            body = buf[pos:pos + n]; pos += n
            if f == 3:
                out['file_version'] = bytes(body).decode()
            elif f == 5:
                for f1, _, val in _fields(body):
                    if f1 == 1:
                        tg, sv = None, None
                        for f2, wt2, x in _fields(val):
                            if f2 == 1:
                                tg = bytes(x).decode()
                            elif f2 == 2 and wt2 == 5:
                                (sv,) = struct.unpack('<f', x)
                        out['scalars'][tg] = sv
        elif wt == 5:
            pos += 4
    return out


class EventFileWriter:
    """events.out.tfevents.<time>.<host> under `logdir`: TFRecord framing, first record = file_version 'brain.Event:2'."""

    def __init__(self, logdir):
        import socket, time
        os.makedirs(logdir, exist_ok=True)
        self.path = os.path.join(logdir, f'events.out.tfevents.{int(time.time())}.{socket.gethostname()}.{os.getpid()}.v2')
        self._w = TFRecordWriter(self.path)
        self._time = time.time
        self._w.write(encode_event(self._time(), 0, file_version='brain.Event:2'))

    def scalars(self, step, values):
        self._w.write(encode_event(self._time(), step, scalars=values))
        self._w._f.flush()

    def close(self):
        self._w.close()


def _fields(buf):
    pos = 0
    while pos < len(buf):
        tag, pos = _read_varint(buf, pos)
        wt = tag & 7
        if wt == 2:
            n, pos = _read_varint(buf, pos)
            yield tag >> 3, wt, buf[pos:pos + n]
            pos += n
        elif wt == 0:
            v, pos = _read_varint(buf, pos)
            yield tag >> 3, wt, v
        elif wt == 5:
            yield tag >> 3, wt, buf[pos:pos + 4]
            pos += 4
        elif wt == 1:
            yield tag >> 3, wt, buf[pos:pos + 8]
            pos += 8
        else:
            raise ValueError(f'unsupported protobuf wire type {wt}')


def decode_example(payload):
    """serialized tf.train.Example -> {name: float32 ndarray | int64 ndarray | [bytes]}"""
    out = {}
    buf = memoryview(payload)
    for f1, _, features in _fields(buf):
        if f1 != 1:
            continue
        for f2, _, entry in _fields(features):
            if f2 != 1:
                continue
            key, feat = None, None
            for f3, _, val in _fields(entry):
                if f3 == 1:
                    key = bytes(val).decode()
                elif f3 == 2:
                    feat = val
            value = None
            for kind, _, lst in _fields(feat if feat is not None else b''):
                if kind == 2:                                        # FloatList: packed (or repeated fixed32)
                    chunks = [bytes(v) for f, wt, v in _fields(lst) if f == 1]
                    value = np.frombuffer(b''.join(chunks), '<f4')
                elif kind == 3:
                    vals = []
                    for f, wt, v in _fields(lst):
                        if f != 1:
                            continue
                        if wt == 0:
                            vals.append(v)
                        else:
                            pos = 0
                            while pos < len(v):
                                x, pos = _read_varint(v, pos)
                                vals.append(x)
                    value = np.array(vals, dtype=np.uint64).astype(np.int64)
                elif kind == 1:
                    value = [bytes(v) for f, wt, v in _fields(lst) if f == 1]
            out[key] = value
    return out


# ------------------------------------------------------------------ mirrors of the reference's callers
def rescale_tensor(img, axes=[2], epsilon=1e-8, moments=None, splits=None):
    """utils/processing.py:281-322 (NumPy; same arithmetic as utils/array_tools.rescale_array)."""
    def rescale(x):
        if moments:
            mn = np.array([t[0] for t in moments], dtype='float32')
            mx = np.array([t[1] for t in moments], dtype='float32')
        else:
            mn = x.min(axis=tuple(axes), keepdims=True)
            mx = x.max(axis=tuple(axes), keepdims=True)
        return (x - mn) / ((mx - mn) + epsilon)
    if splits:
        parts = np.split(img, np.cumsum(splits)[:-1], axis=2)
        return np.concatenate([rescale(p) for p in parts], axis=2)
    return rescale(img)


def normalize_tensor(x, axes=[2], epsilon=1e-8, moments=None, splits=None):
    """utils/processing.py:225-279: (x - mean) / sqrt(variance + epsilon) with moments over `axes` (tf.nn.moments: population
    variance) or from a list of (mean, variance) tuples; with `splits` the first sum(splits) channels are standardised group by
    group and the remaining channels pass through."""
    def normalize(t):
        if moments:
            mean = np.array([tpl[0] for tpl in moments], dtype='float32')
            variance = np.array([tpl[1] for tpl in moments], dtype='float32')
        else:
            mean = t.mean(axis=tuple(axes), keepdims=True)
            variance = t.var(axis=tuple(axes), keepdims=True)
        return (t - mean) / np.sqrt(variance + epsilon)
    if splits:
        n = sum(splits)
        parts = np.split(x[:, :, 0:n], np.cumsum(splits)[:-1], axis=2)
        return np.concatenate([normalize(p) for p in parts] + [x[:, :, n:]], axis=2)
    return normalize(x)


def calc_ndvi(input):
    """utils/processing.py:116-127: (B8 - B4) / (1e-8 + B8 + B4) from a dictionary of band arrays."""
    nir, red = input.get('B8'), input.get('B4')
    return (nir - red) / (1e-8 + (nir + red))


# ---- training / evaluation datasets over Earth-Engine TFRecords (utils/processing.py:129-183, 335-454), NumPy on the host.  The
# random draws (tf.random.uniform, tf.image.random_flip_*) cannot be reproduced bit for bit without TensorFlow: the same
# distributions are drawn from a seedable NumPy generator (set_seed).
_RNG = np.random.default_rng()


def set_seed(seed):
    global _RNG
    _RNG = np.random.default_rng(seed)


class FixedLenFeature:
    """tf.io.FixedLenFeature(shape, dtype) as used for the `ftDict` argument (one [H, W] float list per band)."""

    def __init__(self, shape, dtype='float32', default_value=None):
        self.shape, self.dtype = tuple(shape), dtype


def aug_tensor_color(img):
    """utils/processing.py:129-152: per-channel contrast about the channel mean and brightness of the mean, both U(0.95, 1.05)."""
    n_ch = img.shape[-1]
    ch_mean = img.mean(axis=(0, 1), keepdims=True)
    contra_mul = _RNG.uniform(0.95, 1.05, (1, 1, n_ch)).astype(np.float32)
    bright_mul = _RNG.uniform(0.95, 1.05, (1, 1, n_ch)).astype(np.float32)
    return (img - ch_mean) * contra_mul + ch_mean * bright_mul


def aug_tensor_morph(img):
    """utils/processing.py:169-183: random left-right flip, random up-down flip, rot90 by a random k in 0..3 (counter-clockwise)."""
    x = img[:, ::-1] if _RNG.random() < 0.5 else img
    x = x[::-1] if _RNG.random() < 0.5 else x
    return np.squeeze(np.rot90(x, int(_RNG.integers(0, 4)), axes=(0, 1)))


def to_tuple(inputs, features, response, axes=[2], splits=None, one_hot=None, moments=None, **kwargs):
    """utils/processing.py:335-392: dict of [H, W] arrays -> (features HWC, labels HWC).  Continuous bands are colour-augmented and
    rescaled, one-hot features and the response are appended, the stack is flipped / rotated as one, labels above 1 become 1."""
    for fxn in kwargs.values():              # custom preprocessing functions receive and return the dictionary
        inputs = fxn(inputs)
    if type(response) == dict:
        key, depth = list(response.keys())[0], list(response.values())[0]
        res = np.squeeze((inputs.get(key).astype(np.uint8)[..., None] == np.arange(depth)).astype(np.float32))
    else:
        res = np.expand_dims(inputs.get(response), axis=2).astype(np.float32)
    if one_hot:
        featList = [inputs.get(key) for key in features if key not in one_hot.keys()]
        hotList = [(inputs.get(key).astype(np.uint8)[..., None] == np.arange(val)).astype(np.float32) for key, val in one_hot.items() if key in features]
    else:
        featList = [inputs.get(key) for key in features]
    bands = np.transpose(np.stack(featList, axis=0), [1, 2, 0]).astype(np.float32)
    bands = aug_tensor_color(bands)
    bands = rescale_tensor(bands, axes=axes, moments=moments, splits=splits)
    stacked = np.concatenate([bands] + (hotList if one_hot else []) + [res], axis=2)
    stacked = aug_tensor_morph(stacked)
    nres = res.shape[2]
    feats, labels = stacked[:, :, :-nres], stacked[:, :, -nres:]
    labels = np.where(labels > 1.0, 1.0, labels)
    return np.ascontiguousarray(feats, dtype=np.float32), np.ascontiguousarray(labels, dtype=np.float32)


class Dataset:
    """The tf.data chain the reference builds -- map / shuffle(buffer) / batch / repeat -- as a re-iterable Python object that
    Model.fit / evaluate / predict accept (an iterable of (features, labels) batches)."""

    def __init__(self, source):
        self._source = source                 # callable -> iterator of elements

    def __iter__(self):
        return self._source()

    def map(self, fn, num_parallel_calls=None):
        return Dataset(lambda: (fn(*e) if isinstance(e, tuple) else fn(e) for e in self._source()))

    def shuffle(self, buffer_size):
        """tf.data shuffle semantics: a buffer of `buffer_size` elements, one drawn at random as each new element arrives."""
        def gen():
            buf = []
            for e in self._source():
                buf.append(e)
                if len(buf) > buffer_size:
                    yield buf.pop(int(_RNG.integers(0, len(buf))))
            while buf:
                yield buf.pop(int(_RNG.integers(0, len(buf))))
        return Dataset(gen)

    def batch(self, n, drop_remainder=False):
        def gen():
            cur = []
            for e in self._source():
                cur.append(e)
                if len(cur) == n:
                    yield tuple(np.stack(c) for c in zip(*cur)) if isinstance(cur[0], tuple) else np.stack(cur)
                    cur = []
            if cur and not drop_remainder:
                yield tuple(np.stack(c) for c in zip(*cur)) if isinstance(cur[0], tuple) else np.stack(cur)
        return Dataset(gen)

    def repeat(self, count=None):
        def gen():
            i = 0
            while count is None or i < count:
                empty = True
                for e in self._source():
                    empty = False
                    yield e
                if empty:
                    return
                i += 1
        return Dataset(gen)

    def take(self, n):
        import itertools
        return Dataset(lambda: itertools.islice(self._source(), n))


def get_dataset(files, ftDict, features, response, axes=[2], splits=None, one_hot=None, moments=None, **kwargs):
    """utils/processing.py:394-419: GZIP TFRecords -> parse_single_example(ftDict) -> to_tuple."""
    files = [files] if isinstance(files, str) else list(files)

    def records():
        for path in files:
            for payload in read_records(path):
                ex = decode_example(payload)
                dic = {}
                for k, spec in ftDict.items():
                    if k not in ex:
                        raise KeyError(f'{path}: feature {k} not in the record')
                    shape = tuple(getattr(spec, 'shape', spec))
                    dic[k] = np.asarray(ex[k], dtype=np.float32).reshape(shape)
                yield to_tuple(dic, features, response, axes, splits, one_hot, moments, **kwargs)
    return Dataset(records)


def get_training_dataset(files, ftDict, features, response, buff, batch=16, repeat=True, axes=[2], splits=None, one_hot=None, moments=None, **kwargs):
    """utils/processing.py:421-441: shuffle(buff).batch(batch)[.repeat()]."""
    dataset = get_dataset(files, ftDict, features, response, axes, splits, one_hot, moments, **kwargs)
    return dataset.shuffle(buff).batch(batch).repeat() if repeat else dataset.shuffle(buff).batch(batch)


def get_eval_dataset(files, ftDict, features, response, axes=[2], splits=None, one_hot=None, moments=None, **kwargs):
    """utils/processing.py:443-454: batch(1)."""
    return get_dataset(files, ftDict, features, response, axes, splits, one_hot, moments, **kwargs).batch(1)


def make_pred_dataset(file_list, features, kernel_shape=[256, 256], kernel_buffer=[128, 128], axes=[2], splits=None, moments=None,
                      one_hot=None, **kwargs):
    """utils/prediction_tools.py:159-226: generator of (1, H+buf, W+buf, C) float32 batches, files in sorted order."""
    file_list = sorted(file_list)
    shape = (kernel_shape[0] + kernel_buffer[0], kernel_shape[1] + kernel_buffer[1])

    def gen():
        for path in file_list:
            for payload in read_records(path):
                dic = {k: v.reshape(shape).astype(np.float32) for k, v in decode_example(payload).items() if k in features}
                missing = [k for k in features if k not in dic]
                if missing:
                    raise KeyError(f'{path}: features {missing} not in the record')
                feat = [dic[k] for k in features if not (one_hot and k in one_hot)]
                bands = np.transpose(np.stack(feat, axis=0), [1, 2, 0])
                bands = rescale_tensor(bands, axes=axes, moments=moments, splits=splits)
                for fxn in kwargs.values():
                    bands = np.concatenate([bands, np.expand_dims(fxn(dic), 2)], axis=2)
                if one_hot:
                    hot = [(dic[k].astype(np.uint8)[..., None] == np.arange(d)).astype(np.float32) for k, d in one_hot.items()]
                    bands = np.concatenate([bands] + hot, axis=2)
                yield bands[None].astype(np.float32)
    return gen()


def make_array_predictions(imageDataset, model, jsonFile, kernel_shape=[256, 256], kernel_buffer=[128, 128]):
    """utils/prediction_tools.py:293-373: run the model over the patches and rebuild the mosaic described by the Earth-Engine
    mixer file (row-major, `patchesPerRow` columns); the buffer is cropped from every patch exactly as coded (x/y naming of the
    reference kept)."""
    with open(jsonFile) as f:
        mixer = json.load(f)
    patches, cols = mixer['totalPatches'], mixer['patchesPerRow']
    predictions = model.predict(imageDataset, steps=patches, verbose=1)
    if type(predictions) == list:
        predictions = np.concatenate([p if p.ndim == 4 else p[..., None] for p in predictions], axis=3)
    x_buffer, y_buffer = int(kernel_buffer[0] / 2), int(kernel_buffer[1] / 2)
    x_size, y_size = kernel_shape[0] + y_buffer, kernel_shape[1] + x_buffer
    rows_out, row = None, None
    for x, prediction in enumerate(predictions, start=1):
        patch = prediction[y_buffer:y_size, x_buffer:x_size, :]
        row = patch if x % cols == 1 or cols == 1 else np.append(row, patch, axis=1)
        if x % cols == 0:
            rows_out = row if x <= cols else np.append(rows_out, row, axis=0)
    return rows_out


def write_tfrecord_predictions(predictions, pred_path, out_image_base, kernel_shape=[256, 256], kernel_buffer=[128, 128]):
    """utils/prediction_tools.py:375-445: one uncompressed TFRecord of tf.train.Examples with float lists b1..bC of the patch
    cropped by the buffer."""
    if type(predictions) == list:
        predictions = np.concatenate([p if p.ndim == 4 else p[..., None] for p in predictions], axis=3)
    C = predictions.shape[-1]
    out_image_file = join(pred_path, f'{out_image_base}.tfrecords')
    x_buffer, y_buffer = int(kernel_buffer[0] / 2), int(kernel_buffer[1] / 2)
    x_size, y_size = x_buffer + kernel_shape[1], y_buffer + kernel_shape[0]
    with TFRecordWriter(out_image_file) as w:
        for prediction in predictions:
            patch = prediction[y_buffer:y_size, x_buffer:x_size, :]
            w.write(encode_example({f'b{i + 1}': np.ndarray.flatten(patch[:, :, i]).astype(np.float32) for i in range(C)}))
    return out_image_file
