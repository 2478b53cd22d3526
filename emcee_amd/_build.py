"""Build libemx.so (HIP/gfx950) in-tree.  hipcc cross-compiles without a GPU."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRCS = [os.path.join(HERE, "csrc", f) for f in ("emx.hip", "emx_small.hip", "emx_aux.hip", "emx_hot.hip", "emx_wide.hip", "emx_mtdev.hip", "emx_slab.hip", "emx_pvalu.hip", "emx_pmix.hip", "emx_pslab.hip", "emx_podd.hip")]     # translation units, built in parallel
SRC = SRCS[0]
# EMX_BUILD_FLAVOUR=exp: the experiments flavour (-DEMX_EXPERIMENTS=1: the timing experiments' skip-phase switches, tuning "ablate",
# tools/ablate.py) as libemx_exp.so beside the product library; load it with EMX_LIB=<path>.  The product is built without them.
FLAVOUR = os.environ.get("EMX_BUILD_FLAVOUR", "")
LIB = os.path.join(HERE, "libemx_exp.so" if FLAVOUR == "exp" else "libemx.so")
HOST_SRCS = [os.path.join(HERE, "csrc", f) for f in ("emx_mtpipe.cpp", "emx_mtjump.cpp")]       # plain host C++ (threads, SIMD clones): no device pass
# every header of csrc/ (globbed: round 5's emx_planlog.hpp / emx_logtab.hpp were missing from a hand-kept list, so an edit of the
# kernels' logarithm rebuilt nothing) + the C ABI
DEPS = SRCS + HOST_SRCS + sorted(os.path.join(HERE, "csrc", f) for f in os.listdir(os.path.join(HERE, "csrc")) if f.endswith((".hpp", ".h"))) + [
    os.path.join(os.path.dirname(HERE), "include", "emx.h")]
HOST_FLAGS = ["-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-pthread", "-fvisibility=hidden"]
# -ffp-contract=off: the proposal arithmetic must round like NumPy's separate multiply/subtract.
# -fvisibility=hidden: the library exports the C ABI of include/emx.h and nothing else (the extern "C" regions push default visibility)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-Wno-unused-value", "-Wno-constant-logical-operand", "-fPIC", "-fvisibility=hidden"] + (
    ["-DEMX_EXPERIMENTS=1"] if FLAVOUR == "exp" else [])
# emx_hot.hip (the headline kernel alone): the ILP instruction scheduler, +2.2 % there (csrc/emx_launch.hpp says why not everywhere)
EXTRA_FLAGS = {"emx_hot.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]}


HASHFILE = LIB + ".srchash"


def _source_hash():
    import hashlib
    h = hashlib.sha256((" ".join(FLAGS + HOST_FLAGS) + repr(sorted(EXTRA_FLAGS.items()))).encode())
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


OBJCACHE = os.path.join(HERE, ".objcache")      # objects keyed by the hash of (translation unit, every header, flags): an edit of one
                                                # .hip file recompiles that file alone (git-ignored and gpurun-ignored: the .so travels)


def _unit_hash(src, flags):
    import hashlib
    h = hashlib.sha256((" ".join(flags)).encode())
    for d in [src] + [x for x in DEPS if x.endswith((".hpp", ".h"))]:
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:20]


def build(force=False, verbose=False):
    if not force and not stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    hostcxx = "/opt/rocm/lib/llvm/bin/clang++"
    if not os.path.exists(hostcxx):
        hostcxx = shutil.which("amdclang++") or shutil.which("clang++") or shutil.which("g++")
    os.makedirs(OBJCACHE, exist_ok=True)
    objs, procs = [], []
    for src in SRCS + HOST_SRCS:
        host = src in HOST_SRCS
        flags = HOST_FLAGS if host else FLAGS + EXTRA_FLAGS.get(os.path.basename(src), [])
        obj = os.path.join(OBJCACHE, "%s.%s.o" % (os.path.basename(src), _unit_hash(src, flags)))
        objs.append(obj)
        if os.path.exists(obj) and not force:
            continue
        for old in os.listdir(OBJCACHE):                  # one object per translation unit
            if old.startswith(os.path.basename(src) + "."):
                os.remove(os.path.join(OBJCACHE, old))
        cmd = [hostcxx if host else hipcc] + flags + ["-c", src, "-o", obj + ".tmp"]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)))
    failed = None
    for cmd, obj, pr in procs:
        _, err = pr.communicate()
        if pr.returncode != 0:
            failed = failed or err
        else:
            os.replace(obj + ".tmp", obj)
    if failed is not None:
        raise RuntimeError("hipcc failed:\n" + failed[-4000:])
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB, "-ldl", "-pthread"]
    if verbose:
        print(" ".join(link))
    r = subprocess.run(link, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc link failed:\n" + r.stderr[-4000:])
    with open(HASHFILE, "w") as f:
        f.write(_source_hash() + "\n")
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
