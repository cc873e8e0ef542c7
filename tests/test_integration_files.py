"""N1 shipped as files: integration/HipRender.{h,cpp} is the adapter a Strelka maintainer compiles -- against the real
headers with -DSKH_WITH_STRELKA_HEADERS, against strelka_amd/host/oka_mirror.h in this repository's own build -- and
integration/strelka_hip.patch is the change to the reference tree (RenderFactory's eCompute branch, the CMake switch, the .skscene
exporter hooks).  Here: the patch applies to the reference as it is (dry run, when the tree is present), the adapter sources keep the
header switch, and the exporter header writes the file oka::Scene::saveDump writes."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INTEG = os.path.join(ROOT, "integration")
REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(REF) or shutil.which("patch") is None, reason="needs the reference tree and patch(1)")
def test_patch_applies_to_the_reference_tree():
    r = subprocess.run(["patch", "--dry-run", "-p1", "-d", REF, "-i", os.path.join(INTEG, "strelka_hip.patch")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    touched = [l.split()[-1] for l in r.stdout.splitlines() if l.startswith("checking file")]
    assert touched == ["src/HdStrelka/RenderDelegate.cpp", "src/HdStrelka/RenderPass.cpp", "src/app/main.cpp", "src/render/CMakeLists.txt",
                       "src/render/render.cpp"]
    assert "FAILED" not in r.stdout and "fuzz" not in r.stdout  # exact context: the reference files as they are


def test_patch_adds_the_ecompute_branch_and_nothing_unconditional():
    text = open(os.path.join(INTEG, "strelka_hip.patch"), newline="").read()
    added = [l[1:] for l in text.splitlines() if l.startswith("+") and not l.startswith("+++")]
    assert any("RenderType::eCompute" in l for l in added) and any("new HipRender()" in l for l in added)
    assert any("skhDumpScene" in l for l in added) and any("STRELKA_DUMP_SKSCENE" in l for l in added)
    removed = [l for l in text.splitlines() if l.startswith("-") and not l.startswith("---")]
    assert not removed  # the reference's own lines all stay: every addition sits behind STRELKA_WITH_HIP
    assert text.count("STRELKA_WITH_HIP") >= 7


def test_adapter_sources_keep_the_header_switch_and_no_mirror_types_leak():
    h = open(os.path.join(INTEG, "HipRender.h")).read()
    c = open(os.path.join(INTEG, "HipRender.cpp")).read()
    assert "#ifdef SKH_WITH_STRELKA_HEADERS" in h and "<render/render.h>" in h and "<scene/scene.h>" in h and "oka_mirror.h" in h
    # the adapter must read the same against glm and against the mirror: no mirror-only spellings
    for bad in (".m[", "float4x4", ".inverse()", ".transposed()"):
        assert bad not in c, bad
    assert "skh_copy_aov" in c and "OptixRender.cpp:1022-1043" in c  # the AOV hand-back after the last sample
    cm = open(os.path.join(INTEG, "strelka_hip.cmake")).read()
    assert "-ffp-contract=off" in cm and "SKH_WITH_STRELKA_HEADERS" in cm and "HipRender.cpp" in cm
    from strelka_amd import build

    for flag in build.FLAGS:
        if flag not in ("-fPIC", "-shared") and not flag.startswith("--offload-arch"):
            assert flag in cm, flag  # the Strelka-side build compiles the kernels with this repository's flags


def test_exporter_header_writes_the_same_file_as_the_mirror_scene(tmp_path):
    from strelka_amd import build, scene_io

    exe = build.build_host()
    out = subprocess.run([exe, "cpu", str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    a = open(os.path.join(tmp_path, "scene.skscene"), "rb").read()
    b = open(os.path.join(tmp_path, "scene_exporter.skscene"), "rb").read()
    assert a == b and len(a) > 1000
    sc = scene_io.load_scene(os.path.join(tmp_path, "scene_exporter.skscene"))
    assert len(sc.arrays()["instances"]) == 5
