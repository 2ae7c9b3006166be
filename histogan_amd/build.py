"""In-tree build of libhistogan_hip.so (hipcc, gfx950 only).

    python -m histogan_amd.build [--force] [--verbose]

The .so lands next to the sources' package (histogan_amd/libhistogan_hip.so) so it
travels with the repo snapshot to the GPU box; it is git-ignored.
"""
import hashlib
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, 'csrc')
ROOT = os.path.dirname(PKG)
# experiment builds: HG_LIB_TAG=foo HG_CFLAGS='-DHG_BWD_WAVES=2' -> libhistogan_hip_foo.so (loaded when HG_LIB_TAG=foo)
TAG = os.environ.get('HG_LIB_TAG', '')
EXTRA = os.environ.get('HG_CFLAGS', '').split()
LIB = os.path.join(PKG, 'libhistogan_hip' + ('_' + TAG if TAG else '') + '.so')
STAMP = os.path.join(PKG, '.libhistogan_hip' + ('_' + TAG if TAG else '') + '.stamp')
ARCH = 'gfx950'


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))


def _digest():
    hsh = hashlib.sha256()
    files = sources() + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h'))
    files += sorted(os.path.join(ROOT, 'include', f) for f in os.listdir(os.path.join(ROOT, 'include')))
    hsh.update(' '.join(EXTRA).encode())
    for f in files:
        hsh.update(f.encode())
        with open(f, 'rb') as fh:
            hsh.update(fh.read())
    return hsh.hexdigest()


def hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return 'hipcc'


def build(force=False, verbose=False):
    """Compile every .hip under csrc/ into one shared library. Returns the path."""
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(STAMP):
        with open(STAMP) as f:
            if f.read().strip() == dig:
                return LIB
    jobs = []
    for src in sources():
        obj = os.path.join(CSRC, os.path.basename(src)[:-4] + ('_' + TAG if TAG else '') + '.o')
        cmd = [hipcc(), f'--offload-arch={ARCH}', '-O3', '-std=c++17', '-fPIC', '-Wall',
               '-Wno-unused-function', *EXTRA, '-I', os.path.join(ROOT, 'include'), '-c', src, '-o', obj]
        if verbose:
            cmd.insert(1, '-Rpass-analysis=kernel-resource-usage')
            print(' '.join(cmd), flush=True)
        jobs.append((cmd, obj))
    # the translation units are independent: compile them side by side (hg_conv.hip alone is ~1 min)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as pool:
        list(pool.map(lambda j: subprocess.check_call(j[0]), jobs))
    objs = [obj for _, obj in jobs]
    cmd = [hipcc(), f'--offload-arch={ARCH}', '-shared', '-fPIC', '-o', LIB] + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(STAMP, 'w') as f:
        f.write(dig)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose='--verbose' in sys.argv))
