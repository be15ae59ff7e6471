"""Build libpamnet_hip.so for gfx950 with hipcc (cross-compiles without a GPU).  In-tree artefact:
physics-aware-multiplex-gnn_amd/pamnet_amd/libpamnet_hip.so (git-ignored, travels to the GPU box with the snapshot)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
REPO = os.path.dirname(PKG)
CSRC = os.path.join(PKG, 'csrc')
INCLUDE = os.path.join(REPO, 'include')
LIB = os.path.join(HERE, 'libpamnet_hip.so')
OBJ = os.path.join(CSRC, 'build')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-I' + INCLUDE, '-I' + CSRC,
         '-Wno-unused-result', '-ffp-contract=on']


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError('hipcc not found')


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    deps += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile every .hip under csrc/ and link the shared library.  Returns the library path."""
    if not force and not _stale():
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()

    def compile_one(src):
        obj = os.path.join(OBJ, os.path.basename(src)[:-4] + '.o')
        hdr_t = max(os.path.getmtime(os.path.join(d, f)) for d in (CSRC, INCLUDE) for f in os.listdir(d)
                    if f.endswith('.h'))
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t):
            cmd = [hipcc] + FLAGS + ['-c', src, '-o', obj]
            if verbose:
                print(' '.join(cmd), flush=True)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError('hipcc failed for %s:\n%s' % (src, r.stderr))
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, sources()))
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('link failed:\n' + r.stderr)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
