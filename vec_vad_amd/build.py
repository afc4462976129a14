"""Build libvecvad_hip.so (all hand-written gfx950 kernels) in-tree with hipcc.

    python -m vec_vad_amd.build            # rebuild whatever is out of date
    python -m vec_vad_amd.build --force    # recompile everything

Staleness is decided by CONTENT, not by mtimes: every object file carries a manifest entry = sha256 of its source, of every
header it can include (csrc/*.h, include/vecvad_hip.h) and of the compiler command line; the library's entry is the hash of
the object hashes.  A source whose hash differs from the manifest is recompiled, whatever the timestamps say, so a stale
binary cannot be tested silently.  hipcc cross-compiles for gfx950 without a GPU; the .so and the manifest are git-ignored but
travel to the GPU box with gpurun.
"""
import glob
import hashlib
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(CSRC, 'libvecvad_hip.so')
MANIFEST = os.path.join(CSRC, 'build', 'manifest.json')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-inline-asm'] + os.environ.get('VV_HIPCC_EXTRA', '').split()


def sources():
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')))


def headers():
    return sorted(glob.glob(os.path.join(CSRC, '*.h'))) + [os.path.normpath(os.path.join(HERE, '..', 'include', 'vecvad_hip.h'))]


def _sha(paths, extra=''):
    h = hashlib.sha256(extra.encode())
    for p in paths:
        h.update(os.path.basename(p).encode())
        with open(p, 'rb') as f:
            h.update(f.read())
    return h.hexdigest()


def _obj(src):
    return os.path.join(CSRC, 'build', os.path.basename(src) + '.o')


def wanted(srcs=None):
    """{object path: content hash it must have been built from} + the library's hash."""
    hh = _sha(headers(), ' '.join(FLAGS))
    objs = {_obj(s): _sha([s], hh) for s in (sources() if srcs is None else srcs)}
    lib = hashlib.sha256(''.join(objs[k] for k in sorted(objs)).encode()).hexdigest()
    return objs, lib


def _load_manifest():
    try:
        with open(MANIFEST) as f:
            return json.load(f)
    except Exception:
        return {}


def needs_build():
    objs, lib = wanted()
    m = _load_manifest()
    return not os.path.exists(LIB) or m.get('lib') != lib


def build(force=False, verbose=True):
    srcs = sources()                    # one snapshot of the source list for compile AND link
    objs, libhash = wanted(srcs)
    m = {} if force else _load_manifest()
    have = m.get('objs', {})
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    os.makedirs(os.path.join(CSRC, 'build'), exist_ok=True)
    procs = []
    for src in srcs:
        obj = _obj(src)
        key = os.path.basename(obj)
        if not force and os.path.exists(obj) and have.get(key) == objs[obj]:
            continue
        cmd = [hipcc] + FLAGS + ['-c', src, '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((subprocess.Popen(cmd), src, key, objs[obj]))
    failed = []
    for p, src, key, h in procs:
        if p.wait() != 0:
            failed.append(src)
            have.pop(key, None)
        else:
            have[key] = h
    with open(MANIFEST, 'w') as f:       # objects that did compile are remembered even when another source failed
        json.dump({'objs': have, 'lib': None if (failed or procs) else m.get('lib')}, f, indent=1)
    if failed:
        raise RuntimeError('hipcc failed on %s' % ', '.join(failed))
    if procs or not os.path.exists(LIB) or m.get('lib') != libhash:
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + [_obj(s) for s in srcs]
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
        with open(MANIFEST, 'w') as f:
            json.dump({'objs': have, 'lib': libhash}, f, indent=1)
    elif verbose:
        print('libvecvad_hip.so is up to date (content hash %s)' % libhash[:16], flush=True)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print(LIB)
