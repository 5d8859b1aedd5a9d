"""Builds libsatcv.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

    python -m satellite_computervision_amd.build

One translation unit per .hip file, compiled in parallel, linked into
satellite_computervision_amd/libsatcv.so.  hipcc cross-compiles without a GPU.
"""
import hashlib
import os
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OUT = os.path.join(HERE, 'libsatcv.so')
OBJDIR = os.path.join(HERE, 'csrc', '_obj')
SOURCES = ['api.hip', 'comm.hip', 'conv_igemm.hip', 'conv_igemm_fast.hip', 'conv_igemm_m16.hip', 'conv_igemm_m16p.hip', 'conv_igemm_ws.hip', 'conv_thin_roles.hip', 'conv_transpose_thin.hip', 'conv_bwd_fused.hip', 'convt_bwd_fused.hip', 'conv_wgrad.hip', 'convlstm.hip', 'elementwise.hip', 'input_pipeline.hip']
EXTRA = []
FLAGS = ['--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-Wall', '-Wno-unused-function',
         '-Wno-unused-variable', '-Wno-pass-failed']


def _digest(paths):
    h = hashlib.sha256()
    for p in sorted(paths):
        h.update(open(p, 'rb').read())
    h.update(' '.join(FLAGS).encode())
    h.update(' '.join(EXTRA).encode())
    return h.hexdigest()


def _stamp_commit():
    """the commit this tree was built at, kept beside the library (the GPU box gets no .git): bench.py reports it next to
    the commit its PMC traffic summary was measured at."""
    try:
        root = os.path.dirname(HERE)
        r = subprocess.run(['git', '-C', root, 'rev-parse', '--short', 'HEAD'], capture_output=True, text=True, timeout=10)
        if r.returncode == 0 and r.stdout.strip():
            dirty = subprocess.run(['git', '-C', root, 'status', '--porcelain', '--untracked-files=no'], capture_output=True, text=True, timeout=10).stdout.strip()
            open(os.path.join(HERE, '_build_commit.txt'), 'w').write(r.stdout.strip() + ('+dirty' if dirty else ''))
    except Exception:
        pass


def build(force=False, verbose=True, extra_flags=(), out=None, objdir=None):
    global OUT, OBJDIR
    _stamp_commit()
    if out:
        OUT = out
    if objdir:
        OBJDIR = objdir
    os.makedirs(OBJDIR, exist_ok=True)
    EXTRA[:] = list(extra_flags)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(('.hip', '.hpp'))]
    deps.append(os.path.join(os.path.dirname(HERE), 'include', 'satcv.h'))
    stamp = os.path.join(OBJDIR, 'stamp')
    dig = _digest(deps)
    if not force and os.path.exists(OUT) and os.path.exists(stamp) and open(stamp).read() == dig:
        return OUT
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')

    def compile_one(src):
        obj = os.path.join(OBJDIR, src.replace('.hip', '.o'))
        cmd = [hipcc] + FLAGS + EXTRA + ['-c', os.path.join(CSRC, src), '-o', obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'hipcc failed for {src}:\n{r.stderr[-6000:]}')
        if verbose and r.stderr.strip():
            sys.stderr.write(r.stderr[-2000:])
        return obj

    with ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', OUT] + objs + ['-ldl']
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'link failed:\n{r.stderr[-4000:]}')
    open(stamp, 'w').write(dig)
    return OUT


def build_host_asan(verbose=False):
    """HOST side of the C ABI under AddressSanitizer + UndefinedBehaviorSanitizer (sanitizers on the CPU build only: GPU ASan is
    not available on this pool).  Every source is compiled with `--cuda-host-only` -- descriptor validation, tile / split planning, workspace and
    job queries, the CRC-32C of the TFRecord framing; kernels are not emitted, nothing can launch -- and linked into
    csrc/_obj_asan/libsatcv_hostasan.so for tests/asan/host_abi_driver.c (tests/test_host_cpu.py::test_host_side_of_the_c_abi_under_asan_ubsan)."""
    objdir = os.path.join(HERE, 'csrc', '_obj_asan')
    out = os.path.join(objdir, 'libsatcv_hostasan.so')
    os.makedirs(objdir, exist_ok=True)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(('.hip', '.hpp'))] + [os.path.join(os.path.dirname(HERE), 'include', 'satcv.h')]
    h = hashlib.sha256()
    for p in sorted(deps):
        h.update(open(p, 'rb').read())
    stamp = os.path.join(objdir, 'stamp')
    if os.path.exists(out) and os.path.exists(stamp) and open(stamp).read() == h.hexdigest():
        return out
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    flags = ['--offload-arch=gfx950', '--cuda-host-only', '-O1', '-g', '-fPIC', '-std=c++17', '-fsanitize=address,undefined', '-fno-sanitize-recover=undefined',
             '-fno-omit-frame-pointer', '-Wno-unused-function', '-Wno-unused-variable', '-Wno-pass-failed']

    def one(src):
        obj = os.path.join(objdir, src.replace('.hip', '.o'))
        r = subprocess.run([hipcc] + flags + ['-c', os.path.join(CSRC, src), '-o', obj], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'hipcc (host ASan) failed for {src}:\n{r.stderr[-4000:]}')
        return obj
    with ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(one, SOURCES))
    # a host-only object still refers to the device binary its full build would embed (`__hip_fatbin_<hash>`): define empty ones.  The driver
    # interposes the __hipRegister* entry points of the runtime, so these bytes are never parsed
    nm = subprocess.run(['nm', '--undefined-only'] + objs, capture_output=True, text=True).stdout
    syms = sorted({l.split()[-1] for l in nm.splitlines() if '__hip_fatbin' in l})
    stub = os.path.join(objdir, 'fatbin_stubs.c')
    open(stub, 'w').write(''.join(f'const char {sy}[16] = {{0}};\n' for sy in syms))
    subprocess.run(['gcc', '-fPIC', '-c', stub, '-o', stub[:-2] + '.o'], check=True)
    objs.append(stub[:-2] + '.o')
    r = subprocess.run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-fsanitize=address,undefined', '-o', out] + objs + ['-ldl'], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'link (host ASan) failed:\n{r.stderr[-4000:]}')
    open(stamp, 'w').write(h.hexdigest())
    return out


if __name__ == '__main__':
    ab = [a for a in sys.argv[1:] if a.startswith('-D')]
    if ab:      # profiling variant: python -m ...build -DSATCV_ABLATE=2 -> libsatcv_<tag>.so
        tag = ''.join(c for c in '_'.join(ab) if c.isalnum() or c == '_')
        print(build(force=True, extra_flags=ab, out=os.path.join(HERE, f'libsatcv{tag}.so'), 
                    objdir=os.path.join(tempfile.gettempdir(), 'satcv_obj' + tag)))      # (objects of a variant build stay OUT of the tree: everything in-tree ships to every GPU lease)
    elif '--host-asan' in sys.argv:
        print(build_host_asan())
    else:
        print(build(force='--force' in sys.argv))
