"""In-tree build of the HIP C-ABI library (libstrelka_hip.so) for gfx950.  hipcc cross-compiles without a GPU."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB = os.environ.get("SKH_LIB") or os.path.join(HERE, "libstrelka_hip.so")  # SKH_LIB: A/B builds of the same HIP source
SRC = os.path.join(HERE, "csrc", "strelka_hip.hip")
DEPS = [SRC] + [os.path.join(HERE, "csrc", f) for f in ("skh_device.h", "skh_bvh.h", "skh_kernels.h", "skh_trace_body.inc", "skh_libm.h")] + [
    os.path.join(ROOT, "include", "strelka_hip.h"), os.path.abspath(__file__)]  # (this file: the flags)
# -fno-slp-vectorize: the SLP pass pairs scalar float ops into v_pk_* and pays for it in v_mov packing and registers (k_shade
#   128 -> 112 VGPRs, k_trace 80 + 4 spilled -> 77; kitchen +3 %, hair +13 %); max-ilp scheduling: another +0.5 %.  Neither changes
#   a result: no fast-math, no contraction.
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-slp-vectorize", "-mllvm", "-amdgpu-sched-strategy=max-ilp",
         "-fPIC", "-shared", "-std=c++17"]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def needs_build():
    if os.environ.get("SKH_LIB") and os.path.exists(LIB):
        return False  # an A/B variant somebody built with its own -D flags: never rebuilt (with the default flags) behind their back
    return not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in DEPS)


def build_variant(out_path, defines, verbose=False):
    """Same source, extra -D flags (tests use it to exercise rarely-taken paths, e.g. a tiny LDS stack)."""
    cmd = [hipcc()] + FLAGS + ["-D" + d for d in defines] + ["-o", out_path, SRC]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out_path


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    # several ranks of one job may get here at the same time (bench.py under torch.distributed.run): one compiles, into a
    # temporary file that is renamed into place, the others wait on the lock and find the library up to date
    import fcntl

    with open(LIB + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if force or needs_build():
                tmp = "%s.tmp.%d" % (LIB, os.getpid())
                cmd = [hipcc()] + FLAGS + ["-o", tmp, SRC]
                if verbose:
                    print(" ".join(cmd))
                subprocess.check_call(cmd)
                os.replace(tmp, LIB)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB


HOST_DIR = os.path.join(HERE, "host")
HOST_LIB = os.path.join(HOST_DIR, "liboka_hip.so")
HOST_TEST = os.path.join(HOST_DIR, "host_test")


def build_host(force=False, verbose=False):
    """g++ build of the C++ host mirror (oka::HipRender above the C ABI) and its driver program."""
    build(force=False, verbose=verbose)
    integ = os.path.join(ROOT, "integration")
    srcs = [os.path.join(HOST_DIR, f) for f in ("oka_mirror.cpp", "oka_mirror.h", "oka_render.h", "host_test.cpp")] + [
        os.path.join(integ, f) for f in ("HipRender.cpp", "HipRender.h", "SkSceneDump.h")] + [LIB]
    if not force and os.path.exists(HOST_TEST) and all(os.path.getmtime(s) <= os.path.getmtime(HOST_TEST) for s in srcs):
        return HOST_TEST
    libdir = os.path.dirname(LIB)
    common = ["g++", "-std=c++17", "-O2", "-fPIC", "-Wall"]
    # liboka_hip.so = the stand-in types (oka_mirror.cpp) + the backend exactly as a Strelka tree compiles it (integration/HipRender.cpp,
    # here WITHOUT -DSKH_WITH_STRELKA_HEADERS: glm / MDL SDK / OpenUSD headers are not in this image)
    cmds = [common + ["-shared", "-o", HOST_LIB, os.path.join(HOST_DIR, "oka_mirror.cpp"), os.path.join(integ, "HipRender.cpp"),
                      "-I" + os.path.join(ROOT, "include"), "-L" + libdir, "-lstrelka_hip", "-Wl,-rpath,$ORIGIN/.."],
            common + ["-o", HOST_TEST, os.path.join(HOST_DIR, "host_test.cpp"), "-L" + HOST_DIR, "-loka_hip", "-L" + libdir,
                      "-lstrelka_hip", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,$ORIGIN/.."]]
    # (pytest -n runs several workers: one builds, under the lock, into temporary files that are renamed into place; the others wait and find
    # the files up to date -- a reader must never see a half-written liboka_hip.so)
    import fcntl

    def stale():
        return force or not os.path.exists(HOST_TEST) or any(os.path.getmtime(s) > os.path.getmtime(HOST_TEST) for s in srcs)

    with open(HOST_LIB + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if stale():
                tmp_lib, tmp_test = "%s.tmp.%d" % (HOST_LIB, os.getpid()), "%s.tmp.%d" % (HOST_TEST, os.getpid())
                cmds[0][cmds[0].index(HOST_LIB)] = tmp_lib
                cmds[1][cmds[1].index(HOST_TEST)] = tmp_test
                if verbose:
                    print(" ".join(cmds[0]))
                subprocess.check_call(cmds[0])
                os.replace(tmp_lib, HOST_LIB)
                if verbose:
                    print(" ".join(cmds[1]))
                subprocess.check_call(cmds[1])
                os.replace(tmp_test, HOST_TEST)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return HOST_TEST


if __name__ == "__main__":
    build(force=True, verbose=True)
    build_host(force=True, verbose=True)
