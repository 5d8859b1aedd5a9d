"""Device-side tile input pipeline: `UNETDataGenerator` of the reference (utils/processing.py:456-755) with the per-pixel
work (rescale, NaN mask channel, centre trim, colour augmentation, label merging + one-hot, flip/rot90 augmentation) done by
the HIP kernels of csrc/input_pipeline.hip instead of NumPy.  Same constructor arguments and Keras `Sequence` protocol
(`__len__`, `__getitem__`, `on_epoch_end`); `__getitem__` returns DEVICE tensors (features NHWC fp32, one-hot labels fp32)
that `Model.fit` consumes without a host round trip.  File entries may be `.npy` paths (np.load, as the reference) or
in-memory arrays.

Random draws use Python's `random` in the reference's order (per colour-augmented source: contrast, brightness; then
vertical flip, horizontal flip, rot90 count), so `random.seed(k)` reproduces the reference's augmentation choices; the
N(0,1) replacement values of masked pixels come from a counter-based device generator (the reference's np.random.randn
stream cannot be reproduced on the device)."""
import copy
import ctypes as C
import os
import random
from pathlib import Path

import numpy as np
import torch

from . import ops
from ._lib import lib, check, TileDesc

_KIND = {np.dtype('uint8'): 0, np.dtype('uint16'): 1, np.dtype('float32'): 2, np.dtype('int16'): 3, np.dtype('float64'): 4,
         np.dtype('int32'): 5, np.dtype('int64'): 6}


# ---- file-list helpers that feed the generators (utils/processing.py:26-114); pinned by tests/golden/file_helpers_reference.json
def get_file_id(f, delim='_', parts=slice(3, 5), flag=False):
    """Identifier of a tile file: the `parts` slice of its stem split on `delim`, as a tuple (utils/processing.py:26-45)."""
    return tuple(Path(f).stem.split(delim)[parts])


def match_files(urls, vars, delim='_', parts=slice(3, 5), subset=None, flatdirectory=False):
    """utils/processing.py:47-89: per variable of `vars` (skipping those whose 'files' is None) the sorted files whose identifier
    occurs for EVERY variable (and in `subset`, when given).  A file belongs to variable k if its path contains '/k/' -- or
    '_k_' with flatdirectory.  Returns a deep copy of `vars` with the 'files' lists filled in."""
    out = copy.deepcopy(vars)
    mark = '_{}_' if flatdirectory else '/{}/'
    per_var = {k: [u for u in urls if mark.format(k) in u] for k, v in out.items() if v['files'] is not None}
    common = set.intersection(*[{get_file_id(f, delim, parts) for f in fl} for fl in per_var.values()])
    if subset:
        common &= set(subset)
    for k, fl in per_var.items():
        out[k].update({'files': sorted(f for f in fl if get_file_id(f, delim, parts) in common)})
    return out


def split_files(files, labels=['label', 'lu', 'naip', 'lidar', 's2'], delim='_', parts=slice(3, 5)):
    """utils/processing.py:91-114: one list per label (a path COMPONENT here, not a substring) of the files whose identifier is
    present for every label, in the order of `files`."""
    of = [[f for f in files if lab in Path(f).parts] for lab in labels]
    common = set.intersection(*[{get_file_id(f, delim, parts) for f in fl} for fl in of])
    return [[f for f in fl if get_file_id(f, delim, parts) in common] for fl in of]


def merge_lut(trans, device):
    """merge_classes (utils/array_tools.py:26-44) as a 256-entry table: rules test the ORIGINAL value, the last one wins."""
    lut = np.full(256, -1, np.int32)
    for x, y in trans or []:
        if 0 <= int(x) < 256:
            lut[int(x)] = int(y)
    return torch.from_numpy(lut).to(device)


def _load(item):
    if isinstance(item, (str, os.PathLike)):
        return np.load(item)
    return np.asarray(item)


def _stack_chw(items):
    arrays = [_load(f) for f in items]
    assert len(arrays) > 0 and all(a.ndim == 3 for a in arrays), 'all arrays not 3D'
    chw = [np.moveaxis(a, -1, 0) if a.shape[-1] < a.shape[0] else a for a in arrays]          # utils/processing.py:549
    batch = np.stack(chw, axis=0)
    if batch.dtype not in _KIND:
        batch = batch.astype(np.float64)
    return np.ascontiguousarray(batch)


def device_source(batch_chw, unet_dim, dst, coff, *, rescale_val=0.0, add_nan_mask=False, to_fit=True, color=None, morph=(0, 0, 0), seed=0):
    """One source of one batch -> channels [coff, coff + C (+1)) of the NHWC fp32 tensor `dst`.  color = (contra_mul, bright_mul)
    or None; morph = (flip_v, flip_h, rot)."""
    src = torch.from_numpy(batch_chw).to(dst.device, non_blocking=True)
    n, c, hin, win = batch_chw.shape
    d = TileDesc(src=src.data_ptr(), src_kind=_KIND[batch_chw.dtype], n=n, c=c, hin=hin, win=win, h=unet_dim[0], w_=unet_dim[1],
                 rescale=float(rescale_val or 0.0), nan_mask=int(add_nan_mask) if add_nan_mask in (0, 1, 2) else 1, replace=int(bool(to_fit)), seed=int(seed),
                 flip_v=int(bool(morph[0])), flip_h=int(bool(morph[1])), rot=int(morph[2]) % 4,
                 dst=dst.data_ptr(), ldc=dst.shape[-1], coff=coff)
    st = ops.stream_ptr()
    keep = [src]
    if color is not None:
        mean = torch.empty(n * c, dtype=torch.float64, device=dst.device)
        check(lib.satcv_tile_channel_mean(C.byref(d), mean.data_ptr(), st))
        d.ch_mean, d.contra_mul, d.bright_mul = mean.data_ptr(), float(color[0]), float(color[1])
        keep.append(mean)
    check(lib.satcv_tile_ingest(C.byref(d), st))
    return keep


def device_labels(lc_b1hw, unet_dim, n_classes, dst, coff, lut, lu_b1hw=None, lu_lut=None, morph=(0, 0, 0)):
    lc = torch.from_numpy(np.ascontiguousarray(lc_b1hw)).to(dst.device, non_blocking=True)
    lu = torch.from_numpy(np.ascontiguousarray(lu_b1hw)).to(dst.device, non_blocking=True) if lu_b1hw is not None else None
    n, _, hin, win = lc_b1hw.shape
    check(lib.satcv_label_onehot(lc.data_ptr(), _KIND[lc_b1hw.dtype], lut.data_ptr() if lut is not None else None,
                                 lu.data_ptr() if lu is not None else None, _KIND[lu_b1hw.dtype] if lu is not None else 0,
                                 lu_lut.data_ptr() if lu is not None else None, n, hin, win, unet_dim[0], unet_dim[1], n_classes,
                                 int(bool(morph[0])), int(bool(morph[1])), int(morph[2]) % 4, dst.data_ptr(), dst.shape[-1], coff, ops.stream_ptr()))
    return [lc, lu]


class UNETDataGenerator:
    """Same arguments as utils/processing.py:460-468.  `splits` / `moments` are accepted and unused, as in the reference's
    __getitem__."""

    # (attribute, rescale_val, nan mask follows self.mask, colour augmentation when fitting) in the reference's order (:716-744)
    _SOURCES = (('s2files', 10000.0, False, True), ('naipfiles', 255.0, False, True), ('hagfiles', 100, True, False),
                ('demfiles', 2000.0, True, False), ('ssurgofiles', 0.0, False, False), ('lidarfiles', 100, True, False))

    def __init__(self, labelfiles=None, s2files=None, naipfiles=None, hagfiles=None, lidarfiles=None, lufiles=None, demfiles=None,
                 ssurgofiles=None, to_fit=True, batch_size=32, unet_dim=(256, 256), n_channels=4, n_classes=8, shuffle=True,
                 splits=None, moments=None, lc_transitions=[(12, 3), (11, 3), (10, 3), (9, 8), (255, 0)],
                 lu_transitions=[(82, 9), (84, 10)], device=None):
        self.s2files, self.naipfiles, self.hagfiles, self.demfiles = s2files, naipfiles, hagfiles, demfiles
        self.ssurgofiles, self.lidarfiles, self.labelfiles, self.lufiles = ssurgofiles, lidarfiles, labelfiles, lufiles
        self.to_fit, self.batch_size, self.unet_dim = to_fit, batch_size, tuple(unet_dim)
        self.n_channels, self.n_classes, self.shuffle = n_channels, n_classes, shuffle
        self.splits, self.moments = splits, moments
        self.lc_trans, self.lu_trans = lc_transitions, lu_transitions
        self.device = device or torch.device('cuda', torch.cuda.current_device())
        first = labelfiles if labelfiles is not None else next((getattr(self, a) for a, *_ in self._SOURCES if getattr(self, a)), [])
        self.indexes = np.arange(len(first))
        self.mask = bool(to_fit)                                   # :498-500
        self.on_epoch_end()
        if self.shuffle:
            np.random.shuffle(self.indexes)
        self._lut = merge_lut(self.lc_trans, self.device)
        self._lu_lut = merge_lut(self.lu_trans, self.device)
        self._batch_counter = 0

    def __len__(self):
        return int(np.floor(len(self.indexes) / self.batch_size))

    def on_epoch_end(self):
        self.indexes = np.arange(len(self.indexes))
        if self.shuffle:
            np.random.shuffle(self.indexes)

    def __getitem__(self, index):
        idx = self.indexes[index * self.batch_size:(index + 1) * self.batch_size]
        plan = []
        for attr, rescale, masked, colored in self._SOURCES:
            files = getattr(self, attr)
            if not files:
                continue
            batch = _stack_chw([files[k] for k in idx])
            color = None
            if colored and self.to_fit:                            # aug_array_color draws (utils/array_tools.py:180-182)
                color = (random.uniform(1 - 0.05, 1 + 0.05), random.uniform(1 - 0.05, 1 + 0.05))
            plan.append((batch, rescale, masked and self.mask, color))
        nch = sum(b.shape[1] + (1 if m else 0) for b, _, m, _ in plan)
        morph = (0, 0, 0)
        if self.to_fit:                                            # aug_array_morph draws (utils/array_tools.py:202-207)
            morph = (random.uniform(0, 1) < 0.5, random.uniform(0, 1) < 0.5, random.randint(0, 3))
        h, w = self.unet_dim
        ho, wo = (w, h) if morph[2] % 2 else (h, w)
        n = len(idx)
        x = torch.empty(n, ho, wo, nch, dtype=torch.float32, device=self.device)
        keep, off = [], 0
        self._batch_counter += 1
        for batch, rescale, masked, color in plan:
            keep += device_source(batch, self.unet_dim, x, off, rescale_val=rescale, add_nan_mask=masked, to_fit=self.to_fit, color=color,
                                  morph=morph, seed=(self._batch_counter << 20) + off)
            off += batch.shape[1] + (1 if masked else 0)
        if not self.to_fit:
            torch.cuda.current_stream().synchronize()
            return x
        lc = np.stack([_load(self.labelfiles[k]) for k in idx], axis=0)
        assert lc.shape[1] == 1, 'labels must be (1, H, W)'
        if lc.dtype not in _KIND:
            lc = lc.astype(np.int64)
        lu = None
        if self.lufiles:
            lu = np.stack([_load(self.lufiles[k]) for k in idx], axis=0)
            if lu.dtype not in _KIND:
                lu = lu.astype(np.float64)
        y = torch.empty(n, ho, wo, self.n_classes, dtype=torch.float32, device=self.device)
        keep += device_labels(lc, self.unet_dim, self.n_classes, y, 0, self._lut if self.lc_trans else None, lu, self._lu_lut, morph)
        torch.cuda.current_stream().synchronize()                  # the staged host batches may be released
        return x[..., :self.n_channels], y


class SiameseDataGenerator(UNETDataGenerator):
    """utils/processing.py:757-893: two dates of the same tiles for make_siamese_unet -- `[features_before, features_after], labels`.
    Both dates: x / 10000, centre trim, optional validity mask (1 where no band of the pixel is NaN or below -1; NaN elements
    become U[0,1) draws -- the reference's np.random stream is not reproduced), colour augmentation per date when fitting; labels
    are binary (values above 1 become 1; class ids 0..255) and multiplied by the mask of BOTH dates; one flip / rot90 draw for the
    whole stack.  Random draws in the reference's order: colour (before), colour (after), flips + rotation."""

    def __init__(self, beforefiles, afterfiles, add_nan_mask: bool, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.beforefiles, self.afterfiles = beforefiles, afterfiles
        self.mask = add_nan_mask
        if self.labelfiles is None:
            self.indexes = np.arange(len(beforefiles))
        if self.shuffle:
            np.random.shuffle(self.indexes)
        lut = np.full(256, -1, np.int32)
        lut[2:] = 1
        self._binary_lut = torch.from_numpy(lut).to(self.device)

    def _date(self, files, idx, color, morph, seed):
        batch = _stack_chw([files[k] for k in idx])
        c = batch.shape[1]
        h, w = self.unet_dim
        ho, wo = (w, h) if morph[2] % 2 else (h, w)
        t = torch.empty(len(idx), ho, wo, c + (1 if self.mask else 0), dtype=torch.float32, device=self.device)
        keep = device_source(batch, self.unet_dim, t, 0, rescale_val=10000.0, add_nan_mask=2 if self.mask else 0, to_fit=True, color=color, morph=morph, seed=seed)
        return t, c, keep

    def __getitem__(self, index):
        idx = self.indexes[index * self.batch_size:(index + 1) * self.batch_size]
        cb = ca = None
        if self.to_fit:
            cb = (random.uniform(1 - 0.05, 1 + 0.05), random.uniform(1 - 0.05, 1 + 0.05))
            ca = (random.uniform(1 - 0.05, 1 + 0.05), random.uniform(1 - 0.05, 1 + 0.05))
        morph = (0, 0, 0)
        if self.to_fit:
            morph = (random.uniform(0, 1) < 0.5, random.uniform(0, 1) < 0.5, random.randint(0, 3))
        self._batch_counter += 1
        tb, c, k1 = self._date(self.beforefiles, idx, cb, morph, (self._batch_counter << 21) + 1)
        ta, _, k2 = self._date(self.afterfiles, idx, ca, morph, (self._batch_counter << 21) + (1 << 20))
        if not self.to_fit:
            torch.cuda.current_stream().synchronize()
            return [tb[..., :c].contiguous(), ta[..., :c].contiguous()]
        lc = np.stack([np.squeeze(_load(self.labelfiles[k])) for k in idx], axis=0)[:, None]       # (B, 1, H, W)
        assert lc.ndim == 4, 'labels must be 2-D (or (1, H, W)) arrays'
        if lc.dtype not in _KIND:
            lc = lc.astype(np.int64)
        y2 = torch.empty(tb.shape[0], tb.shape[1], tb.shape[2], 2, dtype=torch.float32, device=self.device)
        k3 = device_labels(lc, self.unet_dim, 2, y2, 0, self._binary_lut, morph=morph)
        labels = y2[..., 1:2]
        if self.mask:
            labels = labels * torch.minimum(tb[..., c:c + 1], ta[..., c:c + 1])
        torch.cuda.current_stream().synchronize()
        return [tb[..., :self.n_channels].contiguous(), ta[..., :self.n_channels].contiguous()], labels.contiguous()


# ------------------------------------------------------------------------------------------------ time-series generators
# LSTMDataGenerator / LSTMAutoencoderGenerator / HybridDataGenerator (utils/processing.py:895-1187) feed the ConvLSTM2D models of
# lstm_tools.py.  The sequences are small (6 x 32 x 32 x 6 per tile), so their host-side preparation stays NumPy -- the helpers below
# are pinned by the reference's own function bodies (tests/golden/timeseries_reference.npz) -- and the device work starts at
# satcv_ingest_seq; the U-Net half of the hybrid generator goes through the device pipeline above.
def normalize_timeseries(arr, maxval=10000, axis=-1, e=0.00001):
    """utils/processing.py:185-193: arr / maxval with NaN -> 0"""
    normalized = np.asarray(arr) / maxval
    return np.where(np.isnan(normalized), 0.0, normalized)


def rearrange_timeseries(arr, nbands, time_dim=1):
    """utils/processing.py:195-218: rotate the sequence to a random start (random.randint, the reference's generator), split off the
    last image's first `nbands` bands as the label; re-draw while a label image sums to zero.  Returns (feats, labels, starttime)."""
    timesteps = arr.shape[time_dim]
    starttime = random.randint(0, timesteps - 1)
    rearranged = np.concatenate([arr[:, starttime:timesteps], arr[:, 0:starttime]], axis=1)
    feats = rearranged[:, 0:-1]
    labels = rearranged[:, -1, :, :, 0:nbands]
    if 0.0 in np.sum(labels, axis=(1, 2, 3)):
        feats, labels, starttime = rearrange_timeseries(arr, nbands)
    return feats, labels, starttime


def sin_cos(t, freq=6):
    """utils/processing.py:220-223"""
    import math
    theta = 2 * math.pi * (t / freq)
    return (math.sin(theta), math.cos(theta))


def make_harmonics(times, timesteps, dims):
    """utils/array_tools.py:12-24: (B, H, W, 2) planes of sin / cos of each start time"""
    xys = [sin_cos(time, timesteps) for time in times]
    return np.stack([np.stack([np.full(dims, x), np.full(dims, y)], axis=-1) for x, y in xys], axis=0)


class LSTMDataGenerator:
    """utils/processing.py:895-972.  Files hold (T, C, H, W) arrays (.npy paths or in-memory arrays); a batch is centre-trimmed to `dim`,
    cut to `n_timesteps`, moved to (B, T, H, W, C) and divided by 10000 (NaN -> 0).  to_fit: (features, label) = a random rotation of the
    sequence with its last image's first n_channels bands as the target (the reference's rearrange_timeseries; its trailing
    `array_tools.split_timeseries(rearranged)` call receives a tuple and cannot run -- the split it describes is what rearrange_timeseries
    already returns)."""

    def __init__(self, files=None, to_fit=True, batch_size=32, dim=(256, 256), n_channels=4, n_timesteps=6, shuffle=True):
        self.files, self.to_fit, self.batch_size, self.dim = files, to_fit, batch_size, tuple(dim)
        self.n_channels, self.n_timesteps, self.shuffle = n_channels, n_timesteps, shuffle
        self.on_epoch_end()

    def __len__(self):
        return int(np.floor(len(self.files) / self.batch_size))

    def on_epoch_end(self):
        self.indexes = np.arange(len(self.files))
        if self.shuffle:
            np.random.shuffle(self.indexes)

    def _load_numpy_data(self, files_temp):
        return [_load(f) for f in files_temp]

    def _batch(self, index, steps):
        idx = self.indexes[index * self.batch_size:(index + 1) * self.batch_size]
        files_temp = [self.files[k] for k in idx]
        batch = np.stack(self._load_numpy_data(files_temp), axis=0)                      # (B, T, C, H, W)
        t0, t1 = (batch.shape[3] - self.dim[0]) // 2, (batch.shape[4] - self.dim[1]) // 2
        cut = batch[:, 0:steps, :, t0:t0 + self.dim[0], t1:t1 + self.dim[1]]
        return normalize_timeseries(np.moveaxis(cut, 2, 4), axis=1), files_temp

    def __getitem__(self, index):
        normalized, _ = self._batch(index, self.n_timesteps)
        if self.to_fit:
            feats, labels, _ = rearrange_timeseries(normalized, self.n_channels)
            return feats.astype(np.float32), labels.astype(np.float32)
        return normalized.astype(np.float32)


class LSTMAutoencoderGenerator(LSTMDataGenerator):
    """utils/processing.py:974-1049: n_timesteps + 1 images per tile; to_fit: ([features, harmonics], [reversed features, next image],
    sample weights or None); the file stem's third `_` field is the start date of the series."""

    def __init__(self, harmonics=True, sample_weights=False, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.add_harmonics, self.sample_weights = harmonics, sample_weights

    def __getitem__(self, index):
        normalized, files_temp = self._batch(index, self.n_timesteps + 1)
        starts = [int(Path(str(f)).stem.split('_')[2]) for f in files_temp] if self.add_harmonics else None
        harmonics = None
        if self.to_fit:
            feats, y, start = rearrange_timeseries(normalized, self.n_channels)
            temporal_y = np.flip(feats, axis=1)
            weights = [None, abs(feats[:, -1] - y) / (feats[:, -1] + y)] if self.sample_weights else None
            if self.add_harmonics:
                harmonics = make_harmonics([s_ + start - self.n_timesteps for s_ in starts], self.n_timesteps, self.dim)
            return ([feats.astype(np.float32), None if harmonics is None else harmonics.astype(np.float32)],
                    [np.ascontiguousarray(temporal_y, dtype=np.float32), y.astype(np.float32)], weights)
        if self.add_harmonics:
            harmonics = make_harmonics(starts, self.n_timesteps, self.dim)
        return [normalized.astype(np.float32), None if harmonics is None else harmonics.astype(np.float32)]


class HybridDataGenerator(UNETDataGenerator):
    """utils/processing.py:1051-1187: the U-Net sources (NAIP, DEM, HAG, lidar, SSURGO) through the device pipeline of UNETDataGenerator,
    the Sentinel-2 / Sentinel-1 SEQUENCES ((T, C, h, w) files, divided by 10000 / -50, the Sentinel-2 one colour-augmented when fitting)
    concatenated along the band axis.  Returns [unet features (device), lstm features (B, T, h, w, c)] (+ one-hot labels)."""

    def __init__(self, s1files=None, lstm_dim=(6, 32, 32, 6), lc_transitions=[(12, 3), (11, 3), (10, 3), (9, 8), (255, 0)],
                 lu_transitions=[(82, 9), (84, 10)], unet_dim=(600, 600), *args, **kwargs):
        self._s2seq = kwargs.pop('s2files', None)
        super().__init__(*args, unet_dim=unet_dim, lc_transitions=lc_transitions, lu_transitions=lu_transitions, **kwargs)
        self.s1files, self.lstm_dim, self.n_timesteps = s1files, tuple(lstm_dim), lstm_dim[0]

    def _get_lstm_data(self, files_temp, rescale_val=1.0):
        arrays = [_load(f) for f in files_temp]
        want = (self.lstm_dim[0], self.lstm_dim[3], self.lstm_dim[1], self.lstm_dim[2])
        assert len(arrays) > 0, 'No Array Found'
        assert all(a.shape[0] >= want[0] and a.shape[2] >= want[2] and a.shape[3] >= want[3] for a in arrays), [a.shape for a in arrays]
        batch = np.stack(arrays, axis=0)
        t0, t1 = (batch.shape[3] - self.lstm_dim[1]) // 2, (batch.shape[4] - self.lstm_dim[2]) // 2
        cut = batch[:, 0:self.n_timesteps, :, t0:t0 + self.lstm_dim[1], t1:t1 + self.lstm_dim[2]]
        return normalize_timeseries(np.moveaxis(cut, 2, 4), maxval=rescale_val, axis=1)

    def __getitem__(self, index):
        idx = self.indexes[index * self.batch_size:(index + 1) * self.batch_size]
        seqs = []
        if self._s2seq:
            s2 = self._get_lstm_data([self._s2seq[k] for k in idx], 10000.0)
            if self.to_fit:                      # aug_array_color (utils/array_tools.py:159-190): per-batch contrast / brightness about the band means
                contra, bright = random.uniform(1 - 0.05, 1 + 0.05), random.uniform(1 - 0.05, 1 + 0.05)
                axes = tuple(range(1, s2.ndim - 1))
                mean = s2.mean(axis=axes, keepdims=True)
                s2 = (s2 - mean) * contra + mean * bright
            seqs.append(s2)
        if self.s1files:
            seqs.append(self._get_lstm_data([self.s1files[k] for k in idx], -50.0))
        lstm = np.concatenate(seqs, axis=-1).astype(np.float32)
        rest = super().__getitem__(index)
        if self.to_fit:
            xu, y = rest
            return [xu, lstm], y
        return [rest, lstm]
