"""Build libemx.so (HIP/gfx950) in-tree.  hipcc cross-compiles without a GPU."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "emx.hip")
LIB = os.path.join(HERE, "libemx.so")
DEPS = [SRC] + [os.path.join(HERE, "csrc", f) for f in ("emx_kernels.hpp", "emx_rng.hpp", "mt19937_legacy.hpp")] + [
    os.path.join(os.path.dirname(HERE), "include", "emx.h")]
# -ffp-contract=off: the proposal arithmetic must round like NumPy's separate multiply/subtract.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-Wno-unused-value", "-fPIC", "-shared"]


HASHFILE = LIB + ".srchash"


def _source_hash():
    import hashlib
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for d in DEPS:
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def stale():
    """True when libemx.so is missing or was built from different sources / flags.  Content based
    (not mtime based) so that a snapshot copied to another box is not rebuilt needlessly."""
    if not os.path.exists(LIB) or not os.path.exists(HASHFILE):
        return True
    try:
        return open(HASHFILE).read().strip() != _source_hash()
    except OSError:
        return True


def build(force=False, verbose=False):
    if not force and not stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc] + FLAGS + [SRC, "-o", LIB]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stderr[-4000:])
    with open(HASHFILE, "w") as f:
        f.write(_source_hash() + "\n")
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
