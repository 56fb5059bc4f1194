"""MIOpen (the backbone's convolutions) plumbing: the repo ships the plain-text user find-db with the tuned
algorithm picks for the bench / eval shapes on gfx950 (`geoformer_amd/miopen_db/*.udb.txt, *.ufdb.txt`), so
MIOpen's immediate mode picks them without a multi-minute search.  Round 6: the find-db also holds the problems of the fp32 parity leg
and of the two training steps (forward, backward-data and backward-weights of the backbone's convolutions at 640x480x4 and 640x640x8:
without them the FIRST training step of a process spent ~160 s inside MIOpen - `tools/r06_miopen_db2.sh` is the run that found them),
and MIOpen's compiled-kernel cache of those picks (`miopen_db/cache/*.ukdb`, a build artefact: git-ignored, it travels with the working
tree like the built .so; without it the first use of a pick compiles its kernel).

`use_shipped_find_db()` must run before the first convolution of the process.  It points MIOPEN_USER_DB_PATH at
a PRIVATE per-process copy of the shipped files: several processes (one per GPU) or threads opening the same
user-db under MIOpen's file locks were seen to leave a handle on slow fallback algorithms for the whole run, and
a private copy also keeps the repository files unmodified.  `--tune`-style searches call `save_find_db()` to
copy the extended db back.
"""
import atexit
import glob
import os
import shutil
import tempfile

SHIPPED = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'miopen_db')
_private = [None]


def use_shipped_find_db(force=False):
    """Returns the directory MIOpen will use, or None if the caller already configured MIOPEN_USER_DB_PATH."""
    if 'MIOPEN_USER_DB_PATH' in os.environ and not force:
        return None
    d = tempfile.mkdtemp(prefix=f'geoformer_miopen_{os.getpid()}_')
    for f in glob.glob(os.path.join(SHIPPED, '*.txt')):
        shutil.copy(f, d)
    os.environ['MIOPEN_USER_DB_PATH'] = d
    if 'MIOPEN_CUSTOM_CACHE_DIR' not in os.environ:
        cache = os.path.join(d, 'cache')
        os.makedirs(cache, exist_ok=True)
        for f in glob.glob(os.path.join(SHIPPED, 'cache', '*.ukdb')):      # the compiled kernels of the shipped picks
            shutil.copy(f, cache)
        os.environ['MIOPEN_CUSTOM_CACHE_DIR'] = cache
    _private[0] = d
    atexit.register(shutil.rmtree, d, True)
    return d


def save_find_db():
    """Copies the (extended) private find-db back over the shipped one (after a search)."""
    if _private[0] is None:
        return
    for f in glob.glob(os.path.join(_private[0], '*.txt')):
        shutil.copy(f, SHIPPED)
