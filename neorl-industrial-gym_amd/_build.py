"""Build libnig.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree."""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "csrc", "nig_kernels.hip")
DEPS = [SRC, os.path.join(_HERE, "csrc", "nig_envs.hpp"), os.path.join(_HERE, "csrc", "nig_detmath.hpp"),
        os.path.join(os.path.dirname(_HERE), "include", "nig.h")]
LIB = os.path.join(_HERE, "libnig.so")

# -ffp-contract=off: NumPy evaluates a*b+c with two roundings; the parity bar is bit-level.
# -fno-slp-vectorize: hipcc packs neighbouring scalar f32 adds/muls into v_pk_* pairs; the moves that
# gather their operands cost more issue slots than the packing saves (fused rollout -3 % with SLP on).
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared",
               "-std=c++17", "-Wall", "-Wno-unused-function"]


def find_hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    return None


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in DEPS if os.path.exists(d))


def build(force=False, verbose=False):
    """Compile csrc/ -> libnig.so.  Cross-compiles without a GPU."""
    if not force and not stale():
        return LIB
    hipcc = find_hipcc()
    if hipcc is None:
        raise RuntimeError("hipcc not found: cannot build libnig.so (no CPU fallback exists)")
    cmd = [hipcc] + HIPCC_FLAGS + ["-o", LIB, SRC]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB
