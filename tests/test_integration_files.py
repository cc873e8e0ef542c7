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


def test_cpp_material_translation_equals_the_python_statement(tmp_path):
    """integration/SkhMaterials.h (what HipRender.cpp uses inside the Strelka tree to turn MaterialDescription {file, name, params} into the
    64-byte skh_material) against strelka_amd/scene_io.py::material_from_description on the same descriptions: the default material,
    OmniPBR with and without textures, OmniGlass (clear / frosted / with glass_ior), UsdPreviewSurface (opaque / glass), hair with colour
    and with explicit absorption, an unknown material.  The header is instantiated with a local look-alike of the reference's two structs
    (tests/cpp/skhmaterials_main.cpp): the reference's own headers need glm, which this image lacks."""
    import struct

    import numpy as np

    from strelka_amd import scene as S, scene_io

    T = {"float": 0, "int": 1, "bool": 2, "float2": 3, "float3": 4, "float4": 5, "texture": 6}

    def P(name, typ, value):
        return {"name": name, "type": typ, "value": value}

    cases = [
        {"file": "default.mdl", "name": "default_material", "params": [P("diffuse_color", "float3", [0.3, 0.5, 0.7])]},
        {"file": "default.mdl", "name": "default_material", "params": []},
        {"file": "OmniPBR.mdl", "name": "OmniPBR", "params": [P("diffuse_color_constant", "float3", [0.9, 0.1, 0.2]), P("reflection_roughness_constant", "float", 0.35),
                                                               P("metallic_constant", "float", 1.0)]},
        {"file": "OmniPBR.mdl", "name": "OmniPBR", "params": [P("diffuse_color_constant", "float3", [0.5, 0.5, 0.5]), P("diffuse_texture", "texture", "wood.png"),
                                                               P("normalmap_texture", "texture", "wood_n.png")]},
        {"file": "OmniPBR.mdl", "name": "OmniPBR", "params": []},
        {"file": "OmniGlass.mdl", "name": "OmniGlass", "params": [P("enable_opacity", "bool", True), P("thin_walled", "bool", False), P("frosting_roughness", "float", 0.0)]},
        {"file": "OmniGlass.mdl", "name": "OmniGlass", "params": [P("glass_color", "float3", [0.9, 1.0, 0.95]), P("glass_ior", "float", 1.33), P("frosting_roughness", "float", 0.4)]},
        {"file": "", "name": "UsdPreviewSurface", "params": [P("diffuseColor", "float3", [0.2, 0.4, 0.6]), P("roughness", "float", 0.25), P("metallic", "float", 0.5)]},
        {"file": "", "name": "preview_glass", "params": [P("diffuseColor", "float3", [1.0, 1.0, 1.0]), P("opacity", "float", 0.1), P("ior", "float", 1.45)]},
        {"file": "hair.mdl", "name": "hair_material", "params": [P("diffuse_color", "float3", [0.35, 0.2, 0.1]), P("roughness_R", "float", 0.25), P("roughness_azimuthal", "float", 0.4)]},
        {"file": "hair.mdl", "name": "hair_material", "params": [P("absorption_coefficient", "float3", [0.4, 0.9, 1.8]), P("cuticle_angle", "float", 0.05), P("diffuse_reflection_weight", "float", 0.2),
                                                                  P("diffuse_reflection_tint", "float3", [0.5, 0.4, 0.3]), P("ior", "float", 1.5)]},
        {"file": "something.mdl", "name": "unknown_thing", "params": [P("foo", "int", 3)]},
    ]
    lines = []
    for c in cases:
        lines.append("D %s|%s|%d" % (c["file"], c["name"], len(c["params"])))
        for p_ in c["params"]:
            v = p_["value"]
            if p_["type"] == "texture":
                raw = v.encode()
            elif p_["type"] == "bool":
                raw = bytes([1 if v else 0])
            elif p_["type"] == "int":
                raw = struct.pack("<i", v)
            else:
                raw = np.asarray(v, np.float32).tobytes()
            lines.append("P %d %s %s" % (T[p_["type"]], p_["name"], raw.hex() or "-"))
    exe = str(tmp_path / "skhmat")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-o", exe, os.path.join(ROOT, "tests", "cpp", "skhmaterials_main.cpp")])
    out = subprocess.run([exe], input="\n".join(lines).encode(), capture_output=True, timeout=60)
    assert out.returncode == 0, out.stderr
    got = np.frombuffer(out.stdout, dtype=S.MATERIAL)
    assert len(got) == len(cases)
    for k, c in enumerate(cases):
        want = scene_io.material_from_description(c)
        assert got[k]["type"] == want["type"], (k, c["name"])
        for f in ("base_color", "roughness", "metallic", "specular", "ior", "reserved"):
            assert np.allclose(got[k][f], want[f], rtol=2e-6, atol=1e-7), (k, c["name"], f, got[k][f], want[f])
    # texture ids: the caller numbers the textures it loaded; a material without texture parameters gets none
    assert (got[3]["base_color_texture"], got[3]["normal_texture"]) == (1, 2) and got[2]["base_color_texture"] == 0


LOOKALIKE = os.path.join(ROOT, "tests", "cpp", "strelka_lookalike")


def test_real_header_branches_go_through_a_compiler():
    """The -DSKH_WITH_STRELKA_HEADERS branches of integration/HipRender.{h,cpp} (texture loading through stb_image, materials through
    SkhMaterials.h, <render/render.h> ... includes) compile -- syntax and types -- against tests/cpp/strelka_lookalike/: forwarding headers
    that stand where the Strelka tree's would be found and switch this repository's stand-in to the REFERENCE's shape of
    Scene::MaterialDescription {file, name, params[MaterialManager::Param]} (scene.h:65-78, materialmanager.h:33-48).  glm / MDL / OpenUSD are
    not in this image, so this is as far as a compiler can take those branches here; nothing is linked."""
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-DSKH_WITH_STRELKA_HEADERS", "-I", LOOKALIKE, "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "integration", "HipRender.cpp")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr


def test_exporter_real_header_branch_round_trips_through_scene_io(tmp_path):
    """integration/SkSceneDump.h with the reference's material descriptions (the branch a Strelka build runs: materials travel as the MDSC JSON
    section, not as argument blocks): tests/cpp/skscene_mdsc_main.cpp fills three descriptions -- default with a float3, OmniPBR with float /
    float3 / a texture path holding quotes, OmniGlass with float / bool / int / float2 / float4 parameters -- and dumps; scene_io.load_scene reads
    the file back and must hold exactly what material_from_description makes of the same descriptions."""
    import numpy as np

    from strelka_amd import scene_io

    exe, dump = str(tmp_path / "mdsc"), str(tmp_path / "m.skscene")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-DSKH_WITH_STRELKA_HEADERS", "-I", LOOKALIKE, "-I", os.path.join(ROOT, "include"), "-o", exe,
                           os.path.join(ROOT, "tests", "cpp", "skscene_mdsc_main.cpp")])
    subprocess.check_call([exe, dump], timeout=60)
    got = scene_io.load_scene(dump).arrays()["materials"]

    def P(name, typ, value):
        return {"name": name, "type": typ, "value": value}

    want = [
        {"file": "default.mdl", "name": "default_material", "params": [P("diffuse_color", "float3", [0.25, 0.5, 0.75])]},
        {"file": "OmniPBR.mdl", "name": "OmniPBR", "params": [P("diffuse_color_constant", "float3", [0.9, 0.1, 0.2]), P("reflection_roughness_constant", "float", 0.35),
                                                               P("metallic_constant", "float", 1.0)]},
        {"file": "OmniGlass.mdl", "name": "OmniGlass", "params": [P("glass_ior", "float", 1.33), P("frosting_roughness", "float", 0.4), P("thin_walled", "bool", True),
                                                                   P("depth", "int", 7)]},
    ]
    assert len(got) == 3
    for k, d in enumerate(want):
        w = scene_io.material_from_description(d)
        assert got[k]["type"] == w["type"]
        for f in ("base_color", "roughness", "metallic", "specular", "ior"):
            assert np.array_equal(got[k][f], w[f]), (k, f, got[k][f], w[f])  # %.9g round-trips a float32 exactly
    assert got[1]["base_color_texture"] == 0  # (its texture file does not exist beside the dump: the constant colour stays, OptixRender.cpp:1195-1199)
