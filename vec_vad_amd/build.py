"""Build libvecvad_hip.so (all hand-written gfx950 kernels) in-tree with hipcc.

    python -m vec_vad_amd.build            # rebuild if any source is newer than the .so

hipcc cross-compiles for gfx950 without a GPU; the .so is git-ignored but travels to the GPU box with gpurun.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(CSRC, 'libvecvad_hip.so')


def sources():
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')))


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, '*.h')) + [os.path.join(HERE, '..', 'include', 'vecvad_hip.h')]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objs = []
    procs = []
    os.makedirs(os.path.join(CSRC, 'build'), exist_ok=True)
    for src in sources():
        obj = os.path.join(CSRC, 'build', os.path.basename(src) + '.o')
        cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-c', src, '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((subprocess.Popen(cmd), src))
        objs.append(obj)
    for p, src in procs:
        if p.wait() != 0:
            raise RuntimeError('hipcc failed on %s' % src)
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print(LIB)
