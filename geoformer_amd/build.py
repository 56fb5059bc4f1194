"""In-tree build of libgeoformer_hip.so: hipcc --offload-arch=gfx950 over geoformer_amd/csrc/*.hip.

No cmake, no torch headers: the library is a plain C-ABI shared object (include/geoformer_hip.h).
Objects are cached per source under csrc/_obj keyed on content hashes, so rebuilding after an edit
recompiles only what changed.
"""
import hashlib
import re
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(CSRC, '_obj')
LIB = os.path.join(HERE, 'libgeoformer_hip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-value'] + os.environ.get('HIPCC_EXTRA', '').split()


def _headers_digest():
    h = hashlib.sha256()
    for d in (CSRC, os.path.join(os.path.dirname(HERE), 'include')):
        for f in sorted(os.listdir(d)):
            if f.endswith('.h'):
                h.update(open(os.path.join(d, f), 'rb').read())
    h.update(' '.join(FLAGS).encode())
    return h.hexdigest()


def _file_flags(body):
    """Per-file compiler flags: a `// hipcc-flags: ...` line among the first lines of the source."""
    for line in body.split(b'\n')[:5]:
        if line.startswith(b'// hipcc-flags:'):
            return line[len(b'// hipcc-flags:'):].decode().split()
    return []


def _compile(src, hdig, verbose):
    body = open(src, 'rb').read()
    extra = _file_flags(body)
    for inc in re.findall(rb'#include "([\w.]+\.hip)"', body):       # a source that includes another one is rebuilt with it
        body += open(os.path.join(CSRC, inc.decode()), 'rb').read()
    tag = hashlib.sha256(body + hdig.encode()).hexdigest()[:16]
    obj = os.path.join(OBJ, os.path.basename(src) + '.' + tag + '.o')
    if not os.path.exists(obj):
        for old in os.listdir(OBJ):
            if old.startswith(os.path.basename(src) + '.'):
                os.remove(os.path.join(OBJ, old))
        cmd = [HIPCC, *FLAGS, *extra, '-c', src, '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return obj


def build(verbose=True, jobs=4):
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))
    hdig = _headers_digest()
    with ThreadPoolExecutor(jobs) as ex:
        objs = list(ex.map(lambda s: _compile(s, hdig, verbose), srcs))
    stamp = hashlib.sha256(' '.join(objs).encode()).hexdigest()
    stamp_file = os.path.join(OBJ, 'link.stamp')
    if not (os.path.exists(LIB) and os.path.exists(stamp_file) and open(stamp_file).read() == stamp):
        cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB, *objs]
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        open(stamp_file, 'w').write(stamp)
    return LIB


if __name__ == '__main__':
    print(build(verbose='-q' not in sys.argv))
