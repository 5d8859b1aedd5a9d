"""CPU checks of the host layer: the C-ABI library loads and exports every symbol declared in
include/satcv.h (no compute calls), the Keras-style graph builder mirrors the reference's
structure, chip-index helpers reproduce the reference-generated fixtures, and the N>1 logic
(flat-gradient all-reduce, chip sharding) works under gloo with world_size 2."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def test_library_exports_every_declared_symbol():
    from satellite_computervision_amd import _lib
    hdr = open(os.path.join(ROOT, 'include', 'satcv.h')).read()
    declared = set(re.findall(r'\b(satcv_[a-z0-9_]+)\s*\(', hdr))
    assert declared, 'no declarations parsed'
    so = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(so, name), f'{name} declared in satcv.h but not exported'
    assert declared == set(_lib.EXPORTED_SYMBOLS), declared ^ set(_lib.EXPORTED_SYMBOLS)
    _lib.lib.satcv_version.restype = ctypes.c_char_p
    assert b'gfx950' in _lib.lib.satcv_version()


def test_argument_validation_without_gpu():
    """bad descriptors are rejected on the host with a message (no launch is attempted)."""
    from satellite_computervision_amd import _lib
    d = _lib.ConvDesc()
    rc = _lib.lib.satcv_conv2d_igemm(ctypes.byref(d), None)
    assert rc == -1 and b'null pointer' in _lib.lib.satcv_last_error()
    with pytest.raises(_lib.SatcvError):
        _lib.check(rc)


def test_unet_graph_structure_matches_reference():
    from satellite_computervision_amd import model_tools as mt
    mt.reset_uids()
    m = mt.get_unet_model(2, 4)
    assert m.count_params() == 18536898
    trainable = sum(p.size for p in m.param_specs if 'moving' not in p.kind)
    assert trainable == 18524930
    ops_ = [n.op for n in m.nodes]
    assert ops_.count('cba') == 16 and ops_.count('pool') == 5 and ops_.count('convT') == 5 and ops_.count('concat_bn_relu') == 5
    assert [t.name for t in m.outputs] == ['probs', 'classes']
    # conv_block as coded: one conv per encoder level, BN moving stats updated twice (Q1/Q2)
    enc = [n for n in m.nodes if n.op == 'cba'][:6]
    assert all(n.attrs['bn_updates'] == 2 for n in enc)
    dec = [n for n in m.nodes if n.op == 'cba'][6:]
    assert all(n.attrs['bn_updates'] == 1 for n in dec)
    # Keras-style automatic names: cba2 of every conv_block consumes a name although it is never built
    assert enc[1].layer.name == 'conv2d_2'
    m13 = mt.get_unet_model(2, 13, head_name='lc_')
    assert m13.count_params() == 18539490 and m13.outputs[1].name == 'lc_classes'
    with pytest.raises(AssertionError):
        mt.build_unet_layers(mt.Input([None, None, 4]), filters=[32, 64], factors=[2])
    dbl = mt.get_unet_model(1, 6, double_conv=True)
    assert [n.op for n in dbl.nodes].count('cba') == 22


def test_chip_helpers_match_reference_fixtures():
    from satellite_computervision_amd import prediction_tools as pt
    z = np.load(os.path.join(GOLD, 'tiling_reference.npz'))
    for key in z['cases']:
        _, h, w, c, buff, kernel = str(key).split('_')
        got = pt.generate_chip_indices(np.zeros((int(h), int(w), int(c)), np.float32), int(buff), int(kernel))
        assert np.array_equal(np.asarray(got, np.int64).reshape(-1, 2), z[str(key)]), key

    class Fake:
        def predict(self, x, **kw):
            return 2.0 * x[..., :1]
    arr = z['pc_arr']
    t = pt.predict_chips(arr, [tuple(i) for i in z['pc_idx']], np.zeros(arr.shape[:2]), Fake(), 32, 16, batch_size=5)
    assert np.array_equal(t, z['pc_template'])
    assert np.array_equal(np.stack(pt.extract_chips(z['ec_arr'], 16, 32)), z['ec_chips'])


def test_graph_builders_of_the_atrous_family_and_helpers():
    """Graph construction is pure Python (no GPU): node lists, Keras layer names and parameter counts of get_acnn_model /
    get_acnn_model2 / get_autoencoder; normalize_confusion_matrix against the reference-generated fixture."""
    from satellite_computervision_amd import model_tools as mt
    mt.reset_uids()
    m = mt.get_acnn_model(3, 16, 4, 3)
    assert [n.op for n in m.nodes] == ['input', 'cba', 'raw', 'cba', 'add_relu', 'cba', 'cba', 'add_relu', 'cba', 'head']
    names = [ps.name for ps in m.param_specs]
    assert names[:2] == ['Conv2D_0_1/kernel', 'Conv2D_0_1/bias'] and 'BN_1_2/gamma' not in names and 'BN_2_2/gamma' in names
    assert m.output_names == ['probabilities']
    m2 = mt.get_acnn_model2(2, 4, 16, 2)
    assert m2.count_params() == 7842 and m2.output_names == ['probs']
    ae = mt.get_autoencoder(6, filters=[32, 64], factors=[2, 2])
    assert ae.output_names == ['continuous'] and [n.op for n in ae.nodes][-1] == 'head'
    with pytest.raises(NameError):
        mt.get_acnn_model(3, 16, 4, 1)
    z = np.load(os.path.join(GOLD, 'tiling_reference.npz'))
    assert np.array_equal(mt.normalize_confusion_matrix(z['ncm_in']), z['ncm_out'])
    with pytest.raises(RuntimeError):
        mt.predict_chunk(np.zeros((4, 8, 8), np.float32), 'https://account.blob.core.windows.net/c/model.h5')


def test_loss_spec_resolution():
    from satellite_computervision_amd import model_tools as mt
    m = mt.get_unet_model(2, 4, filters=[32], factors=[2])
    m.compile(optimizer=mt.Adam(9e-4), loss=lambda yt, yp: mt.weighted_bce(yt, yp, 20), metrics=['categorical_accuracy', mt.MeanIoU(2)])
    assert m._loss.kind == 'weighted_bce' and m._loss.weights.tolist() == [20.0]
    assert m.metrics_names == ['loss', 'categorical_accuracy', 'mean_io_u']
    assert abs(float(m.optimizer.learning_rate.numpy()) - 9e-4) < 1e-9


WORKER = r'''
import os, sys, torch, numpy as np
sys.path.insert(0, os.environ["REPO"])
import torch.distributed as dist
from satellite_computervision_amd import parallel
rank, world = parallel.init_from_env("gloo")
assert world == 2
g = torch.arange(1000, dtype=torch.float32) * (rank + 1)
sync = parallel.GradSync(g.numel(), bucket_bytes=1024)
assert len(sync.bounds) == 4 and sync.bounds[0][0] == 0 and sync.bounds[-1] == (744, 1000)        # cut from the end
sync.ready_above(g, 0)                                   # CPU tensors: no early launch, everything goes in __call__
sync(g)
assert torch.equal(g, torch.arange(1000, dtype=torch.float32) * 3)
g2 = torch.ones(1000) * (rank + 1)
sync(g2)                                                 # state resets between steps
assert torch.equal(g2, torch.full((1000,), 3.0))
w = torch.full((7,), float(rank))
parallel.broadcast_state([w], 0)
assert torch.equal(w, torch.zeros(7))
chips = [(y, x) for y in range(3) for x in range(3)]
mine = parallel.shard_list(chips, rank, world)
allc = [None, None]
dist.all_gather_object(allc, mine)
assert sorted(allc[0] + allc[1]) == sorted(chips) and not set(allc[0]) & set(allc[1])
t = torch.zeros(4, 4); t[rank] = rank + 1
parallel.reduce_templates(t)
assert t[0].eq(1).all() and t[1].eq(2).all() and t[2:].eq(0).all()
s = torch.tensor([1.0, 2.0]) * (rank + 1)
parallel.allreduce_mean_(s)
assert torch.allclose(s, torch.tensor([1.5, 3.0])) and parallel.world_size() == 2
dist.barrier(); dist.destroy_process_group()
print("OK", rank)
'''


def test_data_parallel_logic_gloo_world2(tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(WORKER)
    env = dict(os.environ, REPO=ROOT, MASTER_ADDR='127.0.0.1', MASTER_PORT='29613', WORLD_SIZE='2')
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
        assert 'OK' in o


def test_descriptor_layouts_match_the_header(tmp_path):
    """Every descriptor struct of include/satcv.h has the same size and field offsets as its ctypes mirror in _lib.py (compiled
    with the host C compiler), and the binding stub printed in INTEGRATION.md lists the same fields."""
    import subprocess
    from satellite_computervision_amd import _lib
    pairs = {'satcv_conv_desc': _lib.ConvDesc, 'satcv_pack_job': _lib.PackJob, 'satcv_tile_desc': _lib.TileDesc, 'satcv_wgrad_desc': _lib.WgradDesc,
             'satcv_bnbwd_desc': _lib.BnBwdDesc, 'satcv_head_desc': _lib.HeadDesc, 'satcv_bwdf_desc': _lib.BwdfDesc, 'satcv_reduce_job': _lib.ReduceJob, 'satcv_ctbf_desc': _lib.CtbfDesc}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "satcv.h"', 'int main(void) {']
    for cname, cls in pairs.items():
        lines.append(f'  printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['  return 0;', '}']
    src = tmp_path / 'layout.c'
    src.write_text('\n'.join(lines))
    exe = tmp_path / 'layout'
    subprocess.run(['gcc', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)], check=True)
    got = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, cls in pairs.items():
        assert int(got[cname]) == ctypes.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got[f'{cname}.{fname}']) == getattr(cls, fname).offset, f'{cname}.{fname}'
    doc = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    blk = doc[doc.index('class ConvDesc'):doc.index('lib.satcv_conv2d_igemm.argtypes')]
    assert re.findall(r"\('(\w+)',", blk) == [f[0] for f in _lib.ConvDesc._fields_]


def test_file_list_helpers_match_reference_fixture():
    """get_file_id / match_files / split_files (utils/processing.py:26-114) against outputs of the reference's own function bodies
    (tests/golden/make_reference_fixtures.py)."""
    import json
    from satellite_computervision_amd import processing as P
    z = json.load(open(os.path.join(GOLD, 'file_helpers_reference.json')))
    for u, want in z['ids'][:-1]:
        assert list(P.get_file_id(u)) == want
    assert list(P.get_file_id('a-b-c-d-e-f.npy', '-', slice(1, 4))) == z['ids'][-1][1]
    urls, flat = z['urls'], z['flat']
    spec = {'naip': {'files': []}, 's2': {'files': []}, 'label': {'files': []}, 'lidar': {'files': None}}
    assert P.match_files(urls, spec) == z['match'] and spec['naip']['files'] == []              # the argument is not modified
    got = P.match_files(urls, {'naip': {'files': [], 'bands': 4}, 'label': {'files': []}}, subset={('000', '001'), ('001', '002'), ('002', '003')})
    assert got == z['match_subset']
    assert P.match_files(flat, {'naip': {'files': []}, 's2': {'files': []}, 'label': {'files': []}}, parts=slice(3, 5), flatdirectory=True) == z['match_flat']
    assert P.split_files(urls, labels=['label', 'naip', 's2']) == z['split']


def test_lstm_builders_are_reachable_under_the_reference_module_name():
    """a caller of utils/model_tools.py:666-920, 1016 finds the ConvLSTM2D builders under model_tools (they live in lstm_tools)"""
    from satellite_computervision_amd import model_tools as mt, lstm_tools as lt
    for name in ('build_lstm_layers', 'build_lstm_layers2', 'get_lstm_model', 'get_lstm_autoencoder', 'get_hybrid_model', 'get_hierarchical_model'):
        assert getattr(mt, name) is getattr(lt, name)
    with pytest.raises(AttributeError):
        mt.no_such_builder


def test_library_was_built_from_the_sources_in_the_tree():
    """the in-tree libsatcv.so is what every GPU run loads: its build stamp must equal the digest of the current sources and flags
    (a source edit that does not compile, or a forgotten rebuild, would otherwise ship an old library to the GPU box unnoticed)"""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('_satcv_build_t', os.path.join(root, 'satellite_computervision_amd', 'build.py'))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    deps = [os.path.join(b.CSRC, f) for f in os.listdir(b.CSRC) if f.endswith(('.hip', '.hpp'))]
    deps.append(os.path.join(root, 'include', 'satcv.h'))
    stamp = os.path.join(b.OBJDIR, 'stamp')
    if not os.path.exists(stamp):
        pytest.skip('no build stamp (library built elsewhere)')
    assert open(stamp).read() == b._digest(deps), 'libsatcv.so is older than csrc/: run python -m satellite_computervision_amd.build'


def test_host_side_of_the_c_abi_under_asan_ubsan(tmp_path):
    """SURVEY section 5, sanitizers (CPU build only: GPU AddressSanitizer is not available on this pool): every source of the library compiled
    `--cuda-host-only -fsanitize=address,undefined` (satellite_computervision_amd/build.py::build_host_asan), driven by tests/asan/host_abi_driver.c --
    descriptor validation with null / zeroed / absurd descriptors, ~430 000 tile / split / weight-gradient plans through the dry-run and workspace
    queries, the reduce / pack job queries, the option table, CRC-32C against a bitwise restatement.  Any sanitizer report aborts the driver.
    (Its first run found signed overflows of n * h * w in the planners for absurd extents: descriptors beyond 2^31 pixels are now refused.)"""
    import subprocess
    import importlib.util
    spec = importlib.util.spec_from_file_location('_satcv_build', os.path.join(ROOT, 'satellite_computervision_amd', 'build.py'))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    so = b.build_host_asan()
    exe = tmp_path / 'host_abi_driver'
    clang = '/opt/rocm/lib/llvm/bin/clang'
    subprocess.run([clang, '-fsanitize=address,undefined', '-fno-sanitize-recover=undefined', '-g', '-O1', '-I', os.path.join(ROOT, 'include'),
                    os.path.join(ROOT, 'tests', 'asan', 'host_abi_driver.c'), '-o', str(exe), '-L', os.path.dirname(so), '-lsatcv_hostasan',
                    '-Wl,-rpath,' + os.path.dirname(so)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS='detect_leaks=0:abort_on_error=0', UBSAN_OPTIONS='print_stacktrace=1'),
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert 'runtime error' not in r.stderr and 'AddressSanitizer' not in r.stderr, r.stderr[-4000:]
    assert ' 0 failures' in r.stdout, r.stdout
