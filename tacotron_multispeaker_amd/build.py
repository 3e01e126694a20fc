"""Builds the gfx950 HIP kernels + C-ABI into tacotron_multispeaker_amd/lib/libtaco_hip.so with plain hipcc
(cross-compiles without a GPU).  `python -m tacotron_multispeaker_amd.build` or __graft_entry__.build()."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
LIB = os.path.join(LIBDIR, 'libtaco_hip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = (['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC'] + (['-DTACO_STAMP'] if os.environ.get('TACO_STAMP') else [])
         + os.environ.get('HIPCC_EXTRA', '').split())        # developer experiments (scripts/dev_*.sh)


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    inc = os.path.join(os.path.dirname(HERE), 'include')
    deps = (sources() + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(('.hpp', '.h'))]
            + [os.path.join(inc, f) for f in os.listdir(inc) if f.endswith('.h')])
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    objs = []
    procs = []
    for src in sources():
        obj = os.path.join(LIBDIR, os.path.basename(src)[:-4] + '.o')
        objs.append(obj)
        cmd = [HIPCC] + FLAGS + ['-c', src, '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, pr in procs:
        if pr.wait() != 0:
            raise RuntimeError('hipcc failed: ' + ' '.join(cmd))
    cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
