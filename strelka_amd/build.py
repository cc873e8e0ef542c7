"""In-tree build of the HIP C-ABI library (libstrelka_hip.so) for gfx950.  hipcc cross-compiles without a GPU."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB = os.environ.get("SKH_LIB") or os.path.join(HERE, "libstrelka_hip.so")  # SKH_LIB: A/B builds of the same HIP source
SRC = os.path.join(HERE, "csrc", "strelka_hip.hip")
DEPS = [SRC] + [os.path.join(HERE, "csrc", f) for f in ("skh_device.h", "skh_bvh.h", "skh_kernels.h")] + [
    os.path.join(ROOT, "include", "strelka_hip.h")]
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17"]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def needs_build():
    return not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in DEPS)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    cmd = [hipcc()] + FLAGS + ["-o", LIB, SRC]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force=True, verbose=True)
