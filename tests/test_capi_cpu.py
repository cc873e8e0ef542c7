"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a GPU, exports every symbol
include/strelka_hip.h declares, and fails LOUDLY (no fallback) when no device is present."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from strelka_amd import build, capi

    build.build()  # hipcc cross-compiles gfx950 without a GPU
    return capi.load()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "strelka_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(skh_[a-z_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(lib):
    from strelka_amd import capi

    syms = declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/strelka_hip.h but not exported"
    assert sorted(capi.SYMBOLS) == syms  # the Python binding covers the whole ABI
    # the library reports the version of the header it was built from
    header = open(os.path.join(ROOT, "include", "strelka_hip.h")).read()
    assert lib.skh_abi_version() == int(re.search(r"#define SKH_ABI_VERSION (\d+)", header).group(1)) == 5


def test_record_sizes_match_the_reference_layouts(ork):
    from strelka_amd import scene as S

    # Vertex 32 B (scene.h:80-89), Instance 64 B, UniformLight 112 B (Lights.h:5-14), material block 64 B, params 176 B
    assert (S.VERTEX.itemsize, S.MESH.itemsize, S.CURVE.itemsize, S.INSTANCE.itemsize) == (32, 16, 24, 64)
    assert (S.LIGHT.itemsize, S.MATERIAL.itemsize, S.FRAME_PARAMS.itemsize, S.RAY.itemsize, S.HIT.itemsize) == (112, 64, 176, 32, 20)
    for which, dt in enumerate([S.VERTEX, S.MESH, S.CURVE, S.INSTANCE, S.LIGHT, S.MATERIAL, S.FRAME_PARAMS, S.RAY, S.HIT]):
        assert ork.ork_sizeof(which) == dt.itemsize


def test_no_gpu_means_loud_failure_not_fallback(lib):
    import torch

    from strelka_amd import capi

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    h = C.c_void_p()
    assert lib.skh_create(0, C.byref(h)) != 0 and not h
    with pytest.raises(capi.SkhError, match="no CPU fallback"):
        capi.Context(0)
    # null-handle calls are rejected, not crashed
    assert lib.skh_resize(None, 16, 16) != 0
    assert lib.skh_last_error(None) == b"null context"


def test_the_library_is_rebuilt_when_any_file_it_includes_changes():
    """`strelka_amd/build.py` rebuilds libstrelka_hip.so when a file of `DEPS` is newer than it.  Round 6 moved `k_trace`'s loop body into a new file and a STALE
    library went to the GPU box with it (19 false occluders in a test of code that was already fixed): every quoted `#include` reachable from the one translation
    unit, and the C ABI header, must be in `DEPS`."""
    from strelka_amd import build

    seen, todo = set(), [build.SRC]
    while todo:
        f = todo.pop()
        if f in seen:
            continue
        seen.add(f)
        for m in re.finditer(r'^\s*#\s*include\s+"([^"]+)"', open(f, errors="ignore").read(), re.M):
            for base in (os.path.dirname(f), os.path.join(ROOT, "include")):
                cand = os.path.normpath(os.path.join(base, m.group(1)))
                if os.path.exists(cand):
                    todo.append(cand)
                    break
    deps = {os.path.normpath(d) for d in build.DEPS}
    missing = sorted(seen - deps)
    assert not missing, missing
    assert len(seen) >= 6  # the .hip, four headers, the loop body


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under strelka_amd/ (or the C-ABI sources) may reference it."""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "strelka_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".inc")):
                t = open(os.path.join(dirpath, f), errors="ignore").read()
                if re.search(r"orklib|liboracle|oracle/|ork_[a-z]", t):
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad
