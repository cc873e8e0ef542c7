"""glTF 2.0 -> oka::Scene, following the reference's loader semantics (src/sceneloader/gltfloader.cpp) -- SURVEY.md 8(f) N2.

No third-party dependency (the reference uses tinygltf + nlohmann/json + glm): JSON through the standard library, buffers
from external files, data: URIs or the GLB binary chunk.  What the reference does, including its quirks, is kept:

* `loadMaterials` (:407-420): alphaMode "OPAQUE" -> OmniPBR {diffuse_color_constant = baseColorFactor.rgb,
  reflection_roughness_constant, metallic_constant; diffuse_texture / normalmap_texture uris} (:304-352), anything else ->
  OmniGlass {enable_opacity, thin_walled = false, frosting_roughness = roughnessFactor} (:354-406).  The scene's material
  list is model.materials in order; a primitive without a material uses index 0 (:134-138).
* lights: `<model>_light.json` next to the file -> rect lights {position, orientation (Euler degrees), width, height, color,
  intensity} (:597-641); otherwise ONE default distant light: orientation (-45, 15, 0), half angle 5 deg, intensity 100000,
  white (:664-678).
* cameras (:422-451): perspective only, fov = yfov * (180 / 3.1415926); a default camera when the file has none.  A camera
  node sets position = translation * scale and orientation = conjugate(rotation) of its global transform (:276-291).
* nodes (:222-263, :265-302): local = T * R * S (translation scaled by globalScale = 1) or the column-major `matrix`;
  global = parent * local; every primitive of a mesh node becomes its own mesh + instance (:95-208).
* vertices (:140-152): pos * globalScale, normal = packNormal(normalize(n)), uv = packUV(uv or 0); tangent: the reference
  computes ONE tangent from the LAST triangle of the primitive and stores it in that triangle's three vertices only
  (`computeTangent`, :64-93, using its own inconsistent `unpackUV`), every other vertex keeps tangent 0 -- reproduced.
  Deviation: a primitive without NORMAL makes the reference normalise a zero vector (NaN -> undefined cast); here such
  vertices get area-weighted geometric normals instead.
* indices u8 / u16 / u32 only, triangles only (:155-200); non-indexed primitives are rejected (the reference asserts).
Animations (:453-546) are outside the render() hot path and are not loaded.
"""
import base64
import binascii
import json
import math
import os
import struct
import sys
import zlib

import numpy as np

from . import scene as S
from . import scene_io

_COMPONENT = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5125: np.uint32, 5126: np.float32}
_NCOMP = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT2": 4, "MAT3": 9, "MAT4": 16}


class GltfError(ValueError):
    pass


def _read_model(path):
    with open(path, "rb") as f:
        blob = f.read()
    glb_bin = None
    if blob[:4] == b"glTF":
        _, _, total = struct.unpack_from("<4sII", blob, 0)
        off, doc = 12, None
        while off + 8 <= min(total, len(blob)):
            clen, ctype = struct.unpack_from("<II", blob, off)
            chunk = blob[off + 8:off + 8 + clen]
            if ctype == 0x4E4F534A:
                doc = json.loads(chunk.decode())
            elif ctype == 0x004E4942 and glb_bin is None:
                glb_bin = chunk
            off += 8 + clen + (-clen % 4)
        if doc is None:
            raise GltfError(f"{path}: GLB without a JSON chunk")
    else:
        doc = json.loads(blob.decode())
    base = os.path.dirname(os.path.abspath(path))
    buffers = []
    for b in doc.get("buffers", []):
        uri = b.get("uri")
        if uri is None:
            if glb_bin is None:
                raise GltfError(f"{path}: buffer without uri outside a GLB container")
            data = glb_bin
        elif uri.startswith("data:"):
            data = base64.b64decode(uri.split(",", 1)[1])
        else:
            with open(os.path.join(base, uri), "rb") as f:
                data = f.read()
        if len(data) < b.get("byteLength", 0):
            raise GltfError(f"{path}: buffer shorter than its byteLength")
        buffers.append(data)
    return doc, buffers


def _accessor(doc, buffers, index):
    acc = doc["accessors"][index]
    view = doc["bufferViews"][acc["bufferView"]]
    dt = np.dtype(_COMPONENT[acc["componentType"]])
    ncomp = _NCOMP[acc["type"]]
    stride = view.get("byteStride") or dt.itemsize * ncomp
    start = view.get("byteOffset", 0) + acc.get("byteOffset", 0)
    count = acc["count"]
    data = buffers[view["buffer"]]
    if count and start + stride * (count - 1) + dt.itemsize * ncomp > len(data):
        raise GltfError(f"accessor {index} reads past its buffer")
    return np.ndarray((count, ncomp), dt, data, start, (stride, dt.itemsize)).copy()


def pack_tangent(t):
    """packTangent (gltfloader.cpp:45-52): 10-10-10 bits over [-10, 10]."""
    t = np.asarray(t, np.float32).reshape(-1, 3)
    q = ((t + np.float32(10.0)) / np.float32(20.0) * np.float32(511.99999)).astype(np.uint32)
    return (q[:, 0] + (q[:, 1] << np.uint32(10)) + (q[:, 2] << np.uint32(20))).astype(np.uint32)


def _unpack_uv_loader(val):
    """the loader's own unpackUV (gltfloader.cpp:54-61) -- NOT the inverse of packUV; only used by computeTangent"""
    val = np.uint32(val)
    y = np.float32((val & np.uint32(0xFFFF0000)) >> np.uint32(16)) / np.float32(16383.99999) * np.float32(10.0) - np.float32(5.0)
    x = np.float32(val & np.uint32(0x0000FFFF)) / np.float32(16383.99999) * np.float32(10.0) - np.float32(5.0)
    return np.array([x, y], np.float32)


def _compute_tangent(vb, ib):
    """computeTangent (gltfloader.cpp:64-93): the last triangle only."""
    i0, i1, i2 = (int(ib[-3]), int(ib[-2]), int(ib[-1]))
    uv0, uv1, uv2 = (_unpack_uv_loader(vb["uv"][i]) for i in (i0, i1, i2))
    p0, p1, p2 = (vb["pos"][i].astype(np.float32) for i in (i0, i1, i2))
    d1, d2 = p1 - p0, p2 - p0
    e1, e2 = uv1 - uv0, uv2 - uv0
    tangent = np.array([0.0, 0.0, 1.0], np.float32)
    d = np.float32(e1[0] * e2[1] - e1[1] * e2[0])
    if abs(d) > 1e-6:
        r = np.float32(1.0) / d
        tangent = ((d1 * e2[1] - d2 * e1[1]) * r).astype(np.float32)
    packed = pack_tangent(np.clip(tangent, -10.0, 10.0))[0]
    for i in (i0, i1, i2):
        vb["tangent"][i] = packed


def _local_transform(node):
    if node.get("matrix"):
        return np.asarray(node["matrix"], np.float64).reshape(4, 4).T  # glTF / glm::make_mat4: column-major
    t = np.asarray(node.get("translation", (0.0, 0.0, 0.0)), np.float32).astype(np.float64)
    s = np.asarray(node.get("scale", (1.0, 1.0, 1.0)), np.float32).astype(np.float64)
    q = node.get("rotation")  # glTF order x, y, z, w
    rot = S.quat_to_mat4((q[3], q[0], q[1], q[2])) if q else np.eye(4)
    return S.translate(t) @ rot @ S.scale(s)


def _geometric_normals(pos, ib):
    n = np.zeros_like(pos, dtype=np.float64)
    tri = ib.reshape(-1, 3)
    fn = np.cross(pos[tri[:, 1]] - pos[tri[:, 0]], pos[tri[:, 2]] - pos[tri[:, 0]])
    for k in range(3):
        np.add.at(n, tri[:, k], fn)
    l = np.linalg.norm(n, axis=1, keepdims=True)
    return np.where(l > 0, n / np.maximum(l, 1e-30), np.array([0.0, 0.0, 1.0]))


class GltfScene(S.Scene):
    """oka::Scene filled by the loader, plus the reference-style material descriptions it produced."""

    def __init__(self):
        super().__init__()
        self.material_descriptions = []
        self.texture_ids = {}  # uri -> texture id (1-based), loaded relative to the model like resource/searchPath

    def arrays(self):
        arr = super().arrays()
        mats = scene_io.materials_from_descriptions(self.material_descriptions)
        # eTexture parameters (OptixRender.cpp:1352-1377): each uri becomes a texture resource of the material
        for k, d in enumerate(self.material_descriptions):
            for p in d.get("params", []):
                if p.get("type") == "texture" and p.get("value") in self.texture_ids:
                    if p["name"] == "diffuse_texture":
                        mats[k]["base_color_texture"] = self.texture_ids[p["value"]]
                    elif p["name"] == "normalmap_texture":
                        mats[k]["normal_texture"] = self.texture_ids[p["value"]]
        arr["materials"] = mats
        return arr


def _load_materials(doc, sc):
    images, textures = doc.get("images", []), doc.get("textures", [])

    def tex_uri(info):
        if not info or info.get("index", -1) < 0:
            return None
        src = textures[info["index"]].get("source", -1)
        if not 0 <= src < len(images):
            return None
        if images[src].get("uri") is None and "bufferView" in images[src]:
            return f"bufferView:{images[src]['bufferView']}"  # image stored inside a buffer (.glb); tinygltf decodes those too
        return images[src].get("uri")

    for m in doc.get("materials", []):
        pbr = m.get("pbrMetallicRoughness", {})
        base = pbr.get("baseColorFactor", [1.0, 1.0, 1.0, 1.0])
        rough, metal = float(pbr.get("roughnessFactor", 1.0)), float(pbr.get("metallicFactor", 1.0))
        if m.get("alphaMode", "OPAQUE") == "OPAQUE":
            params = [{"name": "diffuse_color_constant", "type": "float3", "value": [float(base[0]), float(base[1]), float(base[2])]},
                      {"name": "reflection_roughness_constant", "type": "float", "value": rough},
                      {"name": "metallic_constant", "type": "float", "value": metal}]
            for key, info in (("diffuse_texture", pbr.get("baseColorTexture")), ("normalmap_texture", m.get("normalTexture"))):
                uri = tex_uri(info)
                if uri is not None:
                    params.append({"name": key, "type": "texture", "value": uri})
            sc.material_descriptions.append({"file": "OmniPBR.mdl", "name": "OmniPBR", "params": params})
        else:
            sc.material_descriptions.append({"file": "OmniGlass.mdl", "name": "OmniGlass", "params": [
                {"name": "enable_opacity", "type": "bool", "value": True}, {"name": "thin_walled", "type": "bool", "value": False},
                {"name": "frosting_roughness", "type": "float", "value": rough}]})


def _load_textures(path, sc, doc=None, buffers=None):
    """The reference resolves texture uris against `resource/searchPath` and loads them with stb_image; here: PNG and
    baseline JPEG (strelka_amd/png.py, jpeg.py), from files next to the model, data: URIs or bufferViews.  A texture that
    cannot be read is reported and left out (the reference logs an error and binds an empty texture,
    OptixRender.cpp:1195-1199) -- the material then keeps its constant colour."""
    from . import jpeg, png

    def decode(blob, name):
        return jpeg.decode_jpeg(blob, name) if blob[:2] == b"\xff\xd8" else png.decode_png(blob, name)

    base = os.path.dirname(os.path.abspath(path))
    for d in sc.material_descriptions:
        for p in d.get("params", []):
            if p.get("type") != "texture" or p["value"] in sc.texture_ids:
                continue
            uri = p["value"]
            try:
                if uri.startswith("data:"):
                    img = decode(base64.b64decode(uri.split(",", 1)[1]), "data: uri")
                elif uri.startswith("bufferView:") and doc is not None:
                    view = doc["bufferViews"][int(uri.split(":")[1])]
                    start = view.get("byteOffset", 0)
                    img = decode(bytes(buffers[view["buffer"]][start:start + view["byteLength"]]), uri)
                else:
                    with open(os.path.join(base, uri), "rb") as f:
                        img = decode(f.read(), uri)
            except (OSError, ValueError, KeyError, IndexError, struct.error, zlib.error, binascii.Error) as e:
                # (ValueError covers PngError / JpegError; the rest: truncated headers, corrupt deflate data, bad base64)
                print(f"[gltf] unable to load texture {uri[:60]}: {e}", file=sys.stderr)
                continue
            sc.texture_ids[uri] = sc.addTexture(img)


def _load_lights(path, sc):
    light_file = path[:path.rfind(".")] + "_light.json"
    if os.path.exists(light_file):
        with open(light_file) as f:
            doc = json.load(f)
        for l in doc["lights"]:
            sc.createLight({"type": 0, "useXform": False, "position": tuple(l["position"]), "orientation": tuple(l["orientation"]),
                            "width": float(l["width"]), "height": float(l["height"]), "color": tuple(l["color"]),
                            "intensity": float(l["intensity"])})
        return True
    sc.createLight({"type": 3, "useXform": False, "position": (0.0, 0.0, 0.0), "orientation": (-45.0, 15.0, 0.0),
                    "halfAngle": 10.0 * 0.5 * (math.pi / 180.0), "intensity": 100000.0, "color": (1.0, 1.0, 1.0), "radius": 0.0})
    return False


def _load_cameras(doc, sc):
    for c in doc.get("cameras", []):
        if c.get("type") == "perspective":
            p = c["perspective"]
            sc.addCamera(S.Camera(fov=float(np.float32(p["yfov"]) * np.float32(180.0 / 3.1415926)), znear=p.get("znear", 0.1),
                                  zfar=p.get("zfar", 1000.0), name=c.get("name", "")))
    if not sc.mCameras:
        sc.addCamera(S.Camera())


def _process_primitive(doc, buffers, sc, prim, transform):
    if prim.get("mode", 4) != 4:
        raise GltfError("only triangle primitives are supported (gltfloader.cpp:95-208)")
    if "POSITION" not in prim.get("attributes", {}) or prim.get("indices", -1) < 0:
        raise GltfError("a primitive needs POSITION and indices (gltfloader.cpp:98,158)")
    att = prim["attributes"]
    pos = _accessor(doc, buffers, att["POSITION"]).astype(np.float32)
    ib = _accessor(doc, buffers, prim["indices"]).reshape(-1)
    if ib.dtype not in (np.uint8, np.uint16, np.uint32):
        raise GltfError(f"index component type {ib.dtype} not supported")
    ib = ib.astype(np.uint32)
    if len(ib) == 0 or len(ib) % 3 or (len(ib) and int(ib.max()) >= len(pos)):
        raise GltfError("index count must be a non-zero multiple of 3 and stay inside the vertex range")
    if "NORMAL" in att:
        n = _accessor(doc, buffers, att["NORMAL"]).astype(np.float32)
        l = np.sqrt((n * n).sum(1, keepdims=True, dtype=np.float32))
        n = n / np.where(l > 0, l, np.float32(1.0))
    else:
        n = _geometric_normals(pos.astype(np.float64), ib).astype(np.float32)
    uv = _accessor(doc, buffers, att["TEXCOORD_0"]).astype(np.float32) if "TEXCOORD_0" in att else np.zeros((len(pos), 2), np.float32)
    vb = np.zeros(len(pos), S.VERTEX)
    vb["pos"] = pos  # * globalScale (= 1)
    vb["normal"] = S.pack_normals(n)
    vb["uv"] = S.pack_uv(np.clip(uv, -10.0, 10.0))
    _compute_tangent(vb, ib)
    mat = prim.get("material", -1)
    mesh_id = sc.createMesh(vb, ib)
    sc.createInstance(S.INSTANCE_MESH, mesh_id, 0 if mat is None or mat < 0 else mat, transform)


def _process_node(doc, buffers, sc, index, base):
    node = doc["nodes"][index]
    glob = base @ _local_transform(node)
    if node.get("mesh", -1) >= 0:
        for prim in doc["meshes"][node["mesh"]].get("primitives", []):
            _process_primitive(doc, buffers, sc, prim, glob)
    elif node.get("camera", -1) >= 0 and node["camera"] < len(sc.mCameras):
        # glm::decompose + conjugate (gltfloader.cpp:276-291)
        m3 = glob[:3, :3]
        scale = np.linalg.norm(m3, axis=0)
        rot = m3 / np.where(scale > 0, scale, 1.0)
        if np.linalg.det(rot) < 0:
            scale, rot = -scale, -rot
        cam = sc.mCameras[node["camera"]]
        r = np.eye(4)
        r[:3, :3] = rot.T  # conjugate(rotation): world -> camera
        cam.rotation = r
        cam.position = glob[:3, 3] * scale
        cam.updateViewMatrix()
    for child in node.get("children", []):
        _process_node(doc, buffers, sc, child, glob)


def load_gltf(path):
    """GltfLoader::loadGltf (gltfloader.cpp:643-689).  Returns a Scene (strelka_amd.scene API + material_descriptions)."""
    doc, buffers = _read_model(path)
    sc = GltfScene()
    _load_materials(doc, sc)
    _load_textures(path, sc, doc, buffers)
    _load_lights(path, sc)
    _load_cameras(doc, sc)
    scenes = doc.get("scenes", [])
    scene_id = doc.get("scene", 0 if scenes else -1)
    if scene_id < 0 or scene_id >= len(scenes):
        raise GltfError(f"{path}: no default scene")
    for root in scenes[scene_id].get("nodes", []):
        _process_node(doc, buffers, sc, root, np.eye(4))
    return sc
