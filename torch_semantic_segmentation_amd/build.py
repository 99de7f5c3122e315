"""Build libtss_hip.so for gfx950 with hipcc (in-tree, next to this file).

    python -m torch_semantic_segmentation_amd.build [--force]

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so travels with the
working tree to the GPU box.  Objects are rebuilt only when a source or header is newer.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
INCLUDE = os.path.join(os.path.dirname(HERE), 'include')
LIB = os.path.join(HERE, 'libtss_hip.so')
SOURCES = ['convgemm.hip', 'pwfast.hip', 'pwbwd.hip', 'pwsweep.hip', 'bneck.hip', 'conv3x3.hip', 'fc1d.hip', 'fcg.hip', 'sconv.hip', 'atrous.hip', 'wstat.hip', 'stem.hip', 'wgrad.hip', 'dwconv.hip', 'dwroll.hip', 'updw.hip', 'pointwise.hip', 'xchg.hip', 'resample.hip', 'ppm.hip', 'loss.hip', 'ohem.hip', 'hostio.hip', 'gate.hip', 'zoo.hip', 'ssnbt.hip', 'prof.cpp']
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-Wall', '-Wno-unused-function',
         '-I' + INCLUDE, '-I' + CSRC]


def _deps_mtime():
    files = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    files.append(os.path.join(INCLUDE, 'tss_hip.h'))
    return max(os.path.getmtime(f) for f in files)


def _compile(src, force):
    path = os.path.join(CSRC, src)
    timing = os.environ.get('TSS_TIMING') == '1'     # debug variant (phase timers): its own object files, never mixed
    obj = os.path.join(CSRC, os.path.splitext(src)[0] + ('.timing.o' if timing else '.o'))
    if (not force and os.path.exists(obj)
            and os.path.getmtime(obj) >= max(os.path.getmtime(path), _deps_mtime())):
        return obj, False
    extra = ['-DTSS_TIMING'] if timing else []
    cmd = [HIPCC] + FLAGS + extra + (['-x', 'hip'] if src.endswith('.cpp') else []) + ['-c', path, '-o', obj]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError('hipcc failed for %s:\n%s\n%s' % (src, res.stdout, res.stderr))
    if res.stderr.strip():
        sys.stderr.write(res.stderr)
    return obj, True


def build(force=False, verbose=True):
    with ThreadPoolExecutor(max_workers=4) as pool:
        results = list(pool.map(lambda s: _compile(s, force), SOURCES))
    objs = [o for o, _ in results]
    variant = 'timing' if os.environ.get('TSS_TIMING') == '1' else 'release'
    stamp = os.path.join(CSRC, '.variant')
    same = os.path.exists(stamp) and open(stamp).read().strip() == variant
    if any(changed for _, changed in results) or not os.path.exists(LIB) or not same:
        cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError('link failed:\n%s\n%s' % (res.stdout, res.stderr))
        open(stamp, 'w').write(variant)
        if verbose:
            print('built', LIB, '(%s)' % variant)
    elif verbose:
        print('up to date', LIB)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
