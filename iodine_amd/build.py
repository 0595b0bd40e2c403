"""Build libiodine_hip.so (gfx950 only) in-tree with hipcc.

``python -m iodine_amd.build`` or ``iodine_amd.build.build()``.  hipcc cross-compiles
without a GPU, so this runs in the build container; the resulting .so travels to the GPU
box with the source snapshot (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, 'libiodine_hip.so')
SOURCES = ['kernels_conv.hip', 'kernels_convws.hip', 'kernels_out.hip', 'kernels_pixel.hip', 'kernels_misc.hip', 'kernels_train.hip', 'kernels_wgrad32.hip', 'kernels_refine.hip', 'kernels_refbwd.hip', 'kernels_refws.hip', 'kernels_refl0.hip', 'kernels_pack.hip', 'kernels_generic.hip', 'kernels_gens2.hip', 'kernels_genl0.hip', 'iodine_api.cpp']
HEADERS = [os.path.join(CSRC, 'common.h'), os.path.join(CSRC, 'pixel_terms.h'), os.path.join(CSRC, 'pack_bodies.h'), os.path.join(ROOT, 'include', 'iodine_hip.h')]
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function',
         '-Wno-unused-result'] + os.environ.get('IODINE_EXTRA_HIPCC_FLAGS', '').split()   # e.g. -DIODINE_TILE_PROF (tools only)


def _hipcc() -> str:
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError('hipcc not found')


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src: str, obj: str, verbose: bool):
    cmd = [_hipcc(), *FLAGS, '-x', 'hip', '-c', src, '-o', obj]
    if verbose:
        print(' '.join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'hipcc failed for {src}:\n{r.stdout}\n{r.stderr}')
    if verbose and r.stderr.strip():
        print(r.stderr, file=sys.stderr)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 and link libiodine_hip.so.  Returns its path."""
    objdir = os.path.join(CSRC, 'build')
    os.makedirs(objdir, exist_ok=True)
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    jobs = []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s) + '.o')
        if force or _stale(o, [s] + HEADERS):
            jobs.append((s, o))
    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(lambda so: _compile(so[0], so[1], verbose), jobs))
    objs = [os.path.join(objdir, os.path.basename(s) + '.o') for s in srcs]
    if force or jobs or _stale(LIB, objs):
        cmd = [_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB, *objs]
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'link failed:\n{r.stdout}\n{r.stderr}')
    return LIB


def source_digest() -> str:
    """sha256 over the kernel / host sources the library is built from (csrc/* and the ABI header).  Measurements kept
    under profiles/ (PMC traffic per launch) carry this digest; bench.py quotes them only while it still matches."""
    import hashlib
    h = hashlib.sha256()
    for s in sorted(SOURCES) + ['common.h', 'pixel_terms.h', 'pack_bodies.h']:
        p = os.path.join(CSRC, s)
        if os.path.exists(p):
            h.update(s.encode())
            h.update(open(p, 'rb').read())
    h.update(open(os.path.join(ROOT, 'include', 'iodine_hip.h'), 'rb').read())
    return h.hexdigest()


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
