"""Flat binary scene dump (".skscene"): the arrays ``oka::Scene`` hands the renderer, written verbatim.

SURVEY.md section 8(f) N2: the reference reaches its scenes through OpenUSD / tinygltf, neither of which exists on the GPU
box.  Where Strelka builds, a ~40-line exporter (`Scene::saveDump` in strelka_amd/host is the same
code against this repository's mirror of the headers) writes what `OptiXRender::render()` uploads on its first frame
(OptixRender.cpp:876-888, getters of include/scene/scene.h:199-216 and :229-327); `bench.py --scene file.skscene` and
`load_scene()` consume it without any third-party dependency.

Layout (little endian)::

    header   : char magic[8] = "SKSCENE\\0"; u32 version = 1; u32 section_count
    section  : char tag[4]; u32 elem_size; u64 count; u8 data[elem_size * count]; zero padding to a multiple of 8

    tag   element                                        reference source
    VERT  32 B vertex {float3 pos; u32 tangent, normal, uv; 2 pad}   Scene::getVertices()            scene.h:80-89
    INDX  u32 mesh-local index                                        Scene::getIndices()
    MESH  4 x u32 {index_offset, index_count, vertex_offset, vertex_count}   Scene::getMeshes()      scene.h:21-27
    CPTS  float3 control point            Scene::getCurvesPoint()
    CWID  float radius                    Scene::getCurvesWidths()
    CVCN  u32 control points per strand   Scene::getCurvesVertexCounts()
    CURV  6 x u32                         Scene::getCurves()                                         scene.h:29-42
    INST  64 B {float m[12] (3x4 row-major = glm::float3x4(glm::rowMajor4(transform)), OptixRender.cpp:438);
                u32 type, geom_id, material_id, light_id}             Scene::getInstances()          scene.h:44-60
    LGHT  112 B UniformLight              Scene::getLights()                                         Lights.h / scene.h:146-155
    MATL  64 B skh_material (the fixed argument block MaterialDescription maps to)
    MDSC  u8 JSON text: the original MaterialDescription list [{"file", "name", "params": [{"name", "type", "value"}]}]
          (optional; `materials_from_descriptions` turns it into MATL when MATL is absent)
    CAMR  96 B {float view[16] (world -> view, row-major); float fov_deg, znear, zfar; u32 pad[5]}   Camera::matrices.view, fov
    TXDS  4 x u32 {offset (texels), width, height, 0} per texture; materials refer to texture k as id k + 1 (0 = none)
    TXEL  u32 RGBA8 texel (R in the low byte), rows top to bottom as stbi_load returns them (OptixRender.cpp:1191-1264)
Unknown tags are skipped, so the format can grow.
"""
import json
import struct

import numpy as np

from . import scene as S

MAGIC = b"SKSCENE\0"
VERSION = 1
CAMERA = np.dtype([("view", np.float32, 16), ("fov", np.float32), ("znear", np.float32), ("zfar", np.float32),
                   ("pad", np.uint32, 5)])
assert CAMERA.itemsize == 96

_SECTIONS = [  # tag, key in Scene.arrays(), dtype
    (b"VERT", "vertices", S.VERTEX),
    (b"INDX", "indices", np.dtype(np.uint32)),
    (b"MESH", "meshes", S.MESH),
    (b"CPTS", "curve_points", np.dtype((np.float32, 3))),
    (b"CWID", "curve_radii", np.dtype(np.float32)),
    (b"CVCN", "curve_vertex_counts", np.dtype(np.uint32)),
    (b"CURV", "curves", S.CURVE),
    (b"INST", "instances", S.INSTANCE),
    (b"LGHT", "lights", S.LIGHT),
    (b"MATL", "materials", S.MATERIAL),
]


def _camera_record(cam):
    rec = np.zeros(1, CAMERA)
    rec["view"][0] = np.asarray(cam.view, np.float64).astype(np.float32).reshape(16)
    rec["fov"], rec["znear"], rec["zfar"] = cam.fov, cam.znear, cam.zfar
    return rec


def save_scene(path, arrays, camera=None, material_descriptions=None):
    """arrays: the dict of Scene.arrays(); camera: scene.Camera (optional); material_descriptions: list of dicts (optional)."""
    sections = []
    for tag, key, dt in _SECTIONS:
        if key in arrays and arrays[key] is not None:
            a = np.ascontiguousarray(arrays[key], dtype=dt.base if dt.shape else dt)
            if dt.shape:
                a = a.reshape((-1,) + dt.shape)
            sections.append((tag, dt.itemsize, len(a), a.tobytes()))
    if arrays.get("textures"):
        desc, texels = S.pack_textures(arrays["textures"])
        sections.append((b"TXDS", S.TEXTURE_DESC.itemsize, len(desc), desc.tobytes()))
        sections.append((b"TXEL", 4, len(texels), texels.tobytes()))
    if material_descriptions is not None:
        text = json.dumps(material_descriptions).encode()
        sections.append((b"MDSC", 1, len(text), text))
    if camera is not None:
        sections.append((b"CAMR", CAMERA.itemsize, 1, _camera_record(camera).tobytes()))
    with open(path, "wb") as f:
        f.write(MAGIC + struct.pack("<II", VERSION, len(sections)))
        for tag, esz, cnt, data in sections:
            assert len(data) == esz * cnt
            f.write(tag + struct.pack("<IQ", esz, cnt) + data + b"\0" * (-len(data) % 8))


class LoadedCamera(S.Camera):
    """oka::Camera restored from its view matrix (position / orientation are not needed by the renderer)."""

    def __init__(self, view, fov, znear, zfar):
        super().__init__(fov=fov, znear=znear, zfar=zfar, name="dumped camera")
        self.view = np.asarray(view, np.float64).reshape(4, 4)

    def updateViewMatrix(self):
        pass


class LoadedScene:
    """Duck-types strelka_amd.scene.Scene for the consumers (arrays(), getCamera())."""

    def __init__(self, arrays, cameras, material_descriptions=None):
        self._arrays = arrays
        self.mCameras = cameras
        self.material_descriptions = material_descriptions

    def arrays(self):
        return self._arrays

    def getCamera(self, index=0):
        return self.mCameras[index]


def load_scene(path):
    with open(path, "rb") as f:
        blob = f.read()
    if blob[:8] != MAGIC:
        raise ValueError(f"{path}: not a .skscene file")
    version, nsec = struct.unpack_from("<II", blob, 8)
    if version != VERSION:
        raise ValueError(f"{path}: version {version}, this reader knows {VERSION}")
    known = {tag: (key, dt) for tag, key, dt in _SECTIONS}
    arrays = {key: np.zeros((0,) + dt.shape, dt.base if dt.shape else dt) for _, key, dt in _SECTIONS}
    cameras, descs = [], None
    tex_desc, texels = np.zeros(0, S.TEXTURE_DESC), np.zeros(0, np.uint32)
    off = 16
    for _ in range(nsec):
        if off + 16 > len(blob):
            raise ValueError(f"{path}: truncated section header")
        tag = blob[off:off + 4]
        esz, cnt = struct.unpack_from("<IQ", blob, off + 4)
        off += 16
        nbytes = esz * cnt
        if off + nbytes > len(blob):
            raise ValueError(f"{path}: section {tag!r} runs past the end of the file")
        data = blob[off:off + nbytes]
        off += nbytes + (-nbytes % 8)
        if tag in known:
            key, dt = known[tag]
            if esz != dt.itemsize:
                raise ValueError(f"{path}: section {tag!r} has {esz}-byte elements, expected {dt.itemsize}")
            a = np.frombuffer(data, dtype=dt.base if dt.shape else dt).copy()
            arrays[key] = a.reshape((-1,) + dt.shape) if dt.shape else a
        elif tag == b"CAMR":
            for rec in np.frombuffer(data, dtype=CAMERA):
                cameras.append(LoadedCamera(rec["view"], float(rec["fov"]), float(rec["znear"]), float(rec["zfar"])))
        elif tag == b"MDSC":
            descs = json.loads(bytes(data).decode())
        elif tag == b"TXDS" and esz == S.TEXTURE_DESC.itemsize:
            tex_desc = np.frombuffer(data, dtype=S.TEXTURE_DESC).copy()
        elif tag == b"TXEL" and esz == 4:
            texels = np.frombuffer(data, dtype=np.uint32).copy()
        # unknown tags: skipped
    arrays["textures"] = []
    for k, d in enumerate(tex_desc):
        n = int(d["width"]) * int(d["height"])
        if n == 0 or int(d["offset"]) + n > len(texels):
            raise ValueError(f"{path}: texture {k} reaches outside the texel section")
        arrays["textures"].append(texels[int(d["offset"]):int(d["offset"]) + n].view(np.uint8).reshape(int(d["height"]), int(d["width"]), 4).copy())
    if len(arrays["materials"]) == 0:
        arrays["materials"] = materials_from_descriptions(descs or [])
    if not cameras:
        cameras.append(S.Camera())
    validate(arrays)
    return LoadedScene(arrays, cameras, descs)


def validate(arrays):
    """The range checks the reference never makes (it trusts its own loaders); a dump comes from outside."""
    nv, ni = len(arrays["vertices"]), len(arrays["indices"])
    for k, m in enumerate(arrays["meshes"]):
        if int(m["index_offset"]) + int(m["index_count"]) > ni or int(m["vertex_offset"]) + int(m["vertex_count"]) > nv:
            raise ValueError(f"mesh {k} reaches outside the index / vertex buffers")
        if m["index_count"] % 3:
            raise ValueError(f"mesh {k}: index count {m['index_count']} is not a multiple of 3")
    nm, nc, nl = len(arrays["meshes"]), len(arrays["curves"]), len(arrays["lights"])
    for k, i in enumerate(arrays["instances"]):
        t, g = int(i["type"]), int(i["geom_id"])
        if t not in (S.INSTANCE_MESH, S.INSTANCE_LIGHT, S.INSTANCE_CURVE):
            raise ValueError(f"instance {k}: unknown type {t}")
        if (t == S.INSTANCE_CURVE and g >= nc) or (t != S.INSTANCE_CURVE and g >= nm):
            raise ValueError(f"instance {k}: geometry {g} does not exist")
        if t == S.INSTANCE_LIGHT and int(i["light_id"]) >= nl:
            raise ValueError(f"instance {k}: light {int(i['light_id'])} does not exist")
    npts, nw, nvc = len(arrays["curve_points"]), len(arrays["curve_radii"]), len(arrays["curve_vertex_counts"])
    nt = len(arrays.get("textures") or [])
    for k, m in enumerate(arrays["materials"]):
        if int(m["base_color_texture"]) > nt or int(m["normal_texture"]) > nt:
            raise ValueError(f"material {k} refers to a texture that does not exist")
    for k, c in enumerate(arrays["curves"]):
        if (int(c["points_start"]) + int(c["points_count"]) > npts or int(c["widths_start"]) + int(c["widths_count"]) > nw or
                int(c["vertex_counts_start"]) + int(c["vertex_counts_count"]) > nvc):
            raise ValueError(f"curve set {k} reaches outside the control-point / radius / count buffers")


# ---- MaterialDescription -> skh_material (what the HipRender adapter does inside Strelka: INTEGRATION.md section 1) ----
def _param(desc, name, default=None):
    for p in desc.get("params", []):
        if p.get("name") == name:
            return p.get("value", default)
    return default


def material_from_description(desc):
    """One reference MaterialDescription {file, name, params[{name, type, value}]} -> one MATERIAL record.
    default.mdl::default_material.diffuse_color (OptixRender.cpp:1090-1097, RenderPass.cpp:222-245) -> diffuse;
    OmniPBR.{diffuse_color_constant, reflection_roughness_constant, metallic_constant} (gltfloader.cpp:304-352) -> PBR;
    OmniGlass (gltfloader.cpp:354-406: enable_opacity, thin_walled, frosting_roughness; glass_ior when present) -> glass;
    UsdPreviewSurface parameter sets (HdStrelka's eMaterialX descriptions) -> PBR / glass; names containing "hair" -> the hair
    BSDF slot.  Unknown materials fall back to the default diffuse 0.8 grey."""
    m = np.zeros(1, S.MATERIAL)[0]
    m["base_color"], m["roughness"], m["specular"], m["ior"] = 0.8, 0.5, 0.5, 1.5
    name = (desc.get("name") or "") + " " + (desc.get("file") or "")
    low = name.lower()
    pnames = {p.get("name") for p in desc.get("params", [])}
    if pnames & {"diffuseColor", "useSpecularWorkflow", "specularColor", "clearcoat", "emissiveColor"}:
        # UsdPreviewSurface: HdStrelkaMaterial copies the node's parameters under their USD names (Material.cpp:52-150) and hands
        # the network to MaterialX -> MDL (RenderPass.cpp:164-172, type eMaterialX).  Spec defaults: diffuseColor 0.18,
        # roughness 0.5, metallic 0, ior 1.5, opacity 1; an opacity below 0.5 is treated as glass.
        opacity = float(_param(desc, "opacity", 1.0))
        m["type"] = S.MAT_GLASS if opacity < 0.5 else S.MAT_PBR
        m["base_color"] = _param(desc, "diffuseColor", (0.18, 0.18, 0.18))
        m["roughness"] = float(_param(desc, "roughness", 0.5))
        m["metallic"] = float(_param(desc, "metallic", 0.0))
        m["ior"] = float(_param(desc, "ior", 1.5))
    elif "omniglass" in low or "glass" in low:
        m["type"] = S.MAT_GLASS
        m["base_color"] = _param(desc, "glass_color", (1.0, 1.0, 1.0))
        m["roughness"] = float(_param(desc, "frosting_roughness", 0.0))
        m["ior"] = float(_param(desc, "glass_ior", 1.491))  # OmniGlass.mdl default
    elif "omnipbr" in low or "pbr" in low:
        m["type"] = S.MAT_PBR
        m["base_color"] = _param(desc, "diffuse_color_constant", (0.2, 0.2, 0.2))  # OmniPBR.mdl default
        m["roughness"] = float(_param(desc, "reflection_roughness_constant", 0.5))
        m["metallic"] = float(_param(desc, "metallic_constant", 0.0))
    elif "hair" in low:
        # a hair material = df::chiang_hair_bsdf behind the `hair` slot (mdlPtxCodeGen.cpp:143-155); parameter names as in the
        # MDL specification of that distribution function, colour -> absorption by Chiang et al. 2016 eq. 9 when no
        # absorption_coefficient is given
        m["type"] = S.MAT_HAIR
        rn = float(_param(desc, "roughness_azimuthal", _param(desc, "roughness", 0.3)))
        sig = _param(desc, "absorption_coefficient", None)
        if sig is None:
            sig = S.hair_sigma_a_from_color(_param(desc, "diffuse_color", _param(desc, "color", (0.35, 0.2, 0.1))), rn)
        m["base_color"] = _param(desc, "diffuse_reflection_tint", (1.0, 1.0, 1.0))
        m["roughness"] = float(_param(desc, "roughness_R", _param(desc, "roughness", 0.3)))
        m["metallic"] = float(_param(desc, "roughness_TT", 0.0))
        m["specular"] = float(_param(desc, "roughness_TRT", 0.0))
        m["ior"] = float(_param(desc, "ior", 1.55))
        m["reserved"] = (sig[0], sig[1], sig[2], rn, float(_param(desc, "cuticle_angle", 0.035)), float(_param(desc, "diffuse_reflection_weight", 0.0)))
    else:
        m["type"] = S.MAT_DIFFUSE
        m["base_color"] = _param(desc, "diffuse_color", (0.8, 0.8, 0.8))
    return m


def materials_from_descriptions(descs):
    out = np.zeros(max(1, len(descs)), S.MATERIAL)
    if not descs:
        out[0]["base_color"] = 0.8  # material 0 = default.mdl::default_material
    for k, d in enumerate(descs):
        out[k] = material_from_description(d)
    return out
