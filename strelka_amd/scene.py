"""Host-side mirror of the reference's scene model (``oka::Scene``, include/scene/scene.h, src/scene/scene.cpp).

Same names and argument meaning as the reference's C++ API (createMesh / createInstance / createCurve /
createLight / addMaterial / addCamera), producing the flat arrays the renderer uploads verbatim
(SURVEY.md section 8b "Inputs read from oka::Scene") in the C-ABI layouts of include/strelka_hip.h.
numpy only -- no GPU work happens here.
"""
import math

import numpy as np

# ---- C-ABI record layouts (include/strelka_hip.h) -------------------------------------------------------
VERTEX = np.dtype([("pos", np.float32, 3), ("tangent", np.uint32), ("normal", np.uint32), ("uv", np.uint32),
                   ("pad0", np.float32), ("pad1", np.float32)])  # scene.h:80-89, 32 B
MESH = np.dtype([("index_offset", np.uint32), ("index_count", np.uint32), ("vertex_offset", np.uint32),
                 ("vertex_count", np.uint32)])  # scene.h:21-27
CURVE = np.dtype([("vertex_counts_start", np.uint32), ("vertex_counts_count", np.uint32), ("points_start", np.uint32),
                  ("points_count", np.uint32), ("widths_start", np.uint32), ("widths_count", np.uint32)])  # scene.h:29-42
INSTANCE = np.dtype([("transform", np.float32, 12), ("type", np.uint32), ("geom_id", np.uint32),
                     ("material_id", np.uint32), ("light_id", np.uint32)])  # scene.h:44-60, 64 B
LIGHT = np.dtype([("points", np.float32, (4, 4)), ("color", np.float32, 4), ("normal", np.float32, 4),
                  ("type", np.int32), ("half_angle", np.float32), ("pad0", np.float32), ("pad1", np.float32)])  # 112 B
MATERIAL = np.dtype([("type", np.uint32), ("base_color", np.float32, 3), ("roughness", np.float32),
                     ("metallic", np.float32), ("specular", np.float32), ("ior", np.float32),
                     ("base_color_texture", np.uint32), ("normal_texture", np.uint32),  # texture ids: 1-based, 0 = none
                     ("reserved", np.float32, 6)])  # 64 B
TEXTURE_DESC = np.dtype([("offset", np.uint32), ("width", np.uint32), ("height", np.uint32), ("pad", np.uint32)])
FRAME_PARAMS = np.dtype([("view_to_world", np.float32, 16), ("clip_to_view", np.float32, 16),
                         ("subframe_index", np.uint32), ("samples_this_launch", np.uint32), ("spp_total", np.uint32),
                         ("max_depth", np.uint32), ("rect_light_sampling_method", np.uint32),
                         ("exposure", np.float32, 3), ("enable_accumulation", np.uint32), ("debug", np.uint32),
                         ("shadow_ray_tmin", np.float32), ("material_ray_tmin", np.float32)])  # 176 B
RAY = np.dtype([("origin", np.float32, 3), ("tmin", np.float32), ("dir", np.float32, 3), ("tmax", np.float32)])
HIT = np.dtype([("t", np.float32), ("instance_id", np.uint32), ("prim_id", np.uint32), ("u", np.float32),
                ("v", np.float32)])
assert VERTEX.itemsize == 32 and INSTANCE.itemsize == 64 and LIGHT.itemsize == 112
assert MATERIAL.itemsize == 64 and FRAME_PARAMS.itemsize == 176 and RAY.itemsize == 32 and HIT.itemsize == 20

INSTANCE_MESH, INSTANCE_LIGHT, INSTANCE_CURVE = 0, 1, 2  # oka::Instance::Type
MAT_DIFFUSE, MAT_PBR, MAT_GLASS, MAT_HAIR = 0, 1, 2, 3
NO_ID = 0xFFFFFFFF


def pack_normals(n):
    """packNormals (scene.cpp:111-117 == HdStrelka/RenderPass.cpp:53-59): 10-10-10 bits, fp32 arithmetic."""
    n = np.asarray(n, np.float32).reshape(-1, 3)
    q = ((n + np.float32(1.0)) / np.float32(2.0) * np.float32(511.99999)).astype(np.uint32)
    return (q[:, 0] + (q[:, 1] << np.uint32(10)) + (q[:, 2] << np.uint32(20))).astype(np.uint32)


def pack_uv(uv):
    """packUV (HdStrelka/RenderPass.cpp:61-67): 16-16 bits over [-10, 10]."""
    uv = np.asarray(uv, np.float32).reshape(-1, 2)
    q = ((uv + np.float32(10.0)) / np.float32(20.0) * np.float32(16383.99999)).astype(np.uint32)
    return (q[:, 0] + (q[:, 1] << np.uint32(16))).astype(np.uint32)


def translate(t):
    m = np.eye(4, dtype=np.float64)
    m[:3, 3] = t
    return m


def scale(s):
    return np.diag([s[0], s[1], s[2], 1.0]).astype(np.float64)


def rotate(axis, angle_rad):
    a = np.asarray(axis, np.float64)
    a = a / np.linalg.norm(a)
    c, s = math.cos(angle_rad), math.sin(angle_rad)
    x, y, z = a
    r = np.array([[c + x * x * (1 - c), x * y * (1 - c) - z * s, x * z * (1 - c) + y * s, 0],
                  [y * x * (1 - c) + z * s, c + y * y * (1 - c), y * z * (1 - c) - x * s, 0],
                  [z * x * (1 - c) - y * s, z * y * (1 - c) + x * s, c + z * z * (1 - c), 0], [0, 0, 0, 1]], np.float64)
    return r


def quat_from_euler_deg(e):
    """glm::quat(glm::radians(eulerDegrees)) as used by Scene::getTransform(desc) (scene.h:331-343)."""
    ex, ey, ez = [math.radians(v) * 0.5 for v in e]
    cx, cy, cz = math.cos(ex), math.cos(ey), math.cos(ez)
    sx, sy, sz = math.sin(ex), math.sin(ey), math.sin(ez)
    w = cx * cy * cz + sx * sy * sz
    x = sx * cy * cz - cx * sy * sz
    y = cx * sy * cz + sx * cy * sz
    z = cx * cy * sz - sx * sy * cz
    return np.array([w, x, y, z], np.float64)


def quat_to_mat4(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y), 0],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x), 0],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y), 0], [0, 0, 0, 1]], np.float64)


class Camera:
    """oka::Camera (include/scene/camera.h, src/scene/camera.cpp): first-person, view dir -Z, reverse-Z projection."""

    def __init__(self, position=(0, 0, 10), fov=45.0, znear=0.1, zfar=1000.0, name="Default camera"):
        self.name = name
        self.fov = float(fov)
        self.znear, self.zfar = float(znear), float(zfar)
        self.position = np.asarray(position, np.float64)
        self.rotation = np.eye(4, dtype=np.float64)  # rotM (world -> camera rotation)
        self.view = None
        self.inv_perspective = None
        self.updateViewMatrix()

    def lookAt(self, eye, target, up=(0, 1, 0)):
        eye, target, up = [np.asarray(v, np.float64) for v in (eye, target, up)]
        f = target - eye
        f /= np.linalg.norm(f)
        s = np.cross(f, up)
        s /= np.linalg.norm(s)
        u = np.cross(s, f)
        r = np.eye(4)
        r[0, :3], r[1, :3], r[2, :3] = s, u, -f
        self.rotation = r
        self.position = eye
        self.updateViewMatrix()

    def updateViewMatrix(self):  # camera.cpp:10-23 (firstperson: view = rotM * transM)
        self.view = self.rotation @ translate(-self.position)

    def updateAspectRatio(self, aspect):  # camera.cpp:125-131 + 61-118: hand-written reverse-Z inverse, fp32
        n, f = np.float32(self.zfar), np.float32(self.znear)  # swapped for reverse z
        focal = np.float32(1.0) / np.float32(math.tan(np.float32(np.float32(self.fov) * np.float32(0.017453292519943295)) / np.float32(2.0)))
        x = np.float32(focal / np.float32(aspect))
        y = focal
        A = np.float32(n / (f - n))
        B = np.float32(f * A)
        one = np.float32(1.0)
        self.inv_perspective = np.array([[one / x, 0, 0, 0], [0, one / y, 0, 0], [0, 0, 0, -1.0], [0, 0, one / B, A / B]],
                                        np.float32)

    def view_to_world_rowmajor(self):  # OptixRender.cpp:953: transpose(inverse(view)) of a column-major glm matrix
        return np.linalg.inv(self.view).astype(np.float32).reshape(16)

    def clip_to_view_rowmajor(self):  # OptixRender.cpp:954
        return self.inv_perspective.astype(np.float32).reshape(16)


class Scene:
    """oka::Scene: flat CPU arrays the renderer uploads verbatim (scene.h:199-216)."""

    def __init__(self):
        self.mVertices = []  # list of VERTEX arrays
        self.mIndices = []
        self.mMeshes = []
        self.mCurves = []
        self.mCurvePoints = []
        self.mCurveWidths = []
        self.mCurveVertexCounts = []
        self.mInstances = []
        self.mLights = []
        self.mLightDesc = []
        self.mMaterials = []
        self.mTextures = []
        self.mCameras = []
        self._nverts = 0
        self._nidx = 0
        self._npoints = 0
        self._nvc = 0
        self.mRectLightMeshId = -1
        self.mSphereLightMeshId = -1
        self.mDiskLightMeshId = -1

    # -- scene.cpp:15-49
    def createMesh(self, vb, ib):
        vb = np.ascontiguousarray(vb, dtype=VERTEX)
        ib = np.ascontiguousarray(ib, dtype=np.uint32).reshape(-1)
        mesh_id = len(self.mMeshes)
        self.mMeshes.append((self._nidx, len(ib), self._nverts, len(vb)))
        self.mIndices.append(ib)
        self.mVertices.append(vb)
        self._nidx += len(ib)
        self._nverts += len(vb)
        return mesh_id

    # -- scene.cpp:51-87
    def createInstance(self, type_, geomId, materialId, transform, lightId=NO_ID):
        t = np.asarray(transform, np.float64).reshape(4, 4)
        inst_id = len(self.mInstances)
        self.mInstances.append((t[:3, :].astype(np.float32).reshape(12), type_, geomId, materialId & 0xFFFFFFFF, lightId))
        return inst_id

    # -- scene.cpp:89-95.  `material` is a dict of the fixed-layout argument block (skh_material)
    def addMaterial(self, type=MAT_DIFFUSE, base_color=(0.8, 0.8, 0.8), roughness=None, metallic=0.0, specular=0.5, ior=1.5,
                    base_color_texture=0, normal_texture=0, reserved=(0.0,) * 6):
        """MAT_GLASS: `roughness` is OmniGlass' frosting_roughness (default 0 = clear glass: gltfloader.cpp:354-406 sets it from the
        file's roughnessFactor, OmniGlass.mdl's own default is clear).  Other types: default 0.5.  MAT_HAIR: use addHairMaterial."""
        if roughness is None:
            roughness = 0.0 if type == MAT_GLASS else 0.5
        self.mMaterials.append((type, tuple(base_color), roughness, metallic, specular, ior, base_color_texture, normal_texture,
                                tuple(reserved)))
        return len(self.mMaterials) - 1

    def addHairMaterial(self, color=(0.35, 0.2, 0.1), roughness_r=0.3, roughness_n=0.3, roughness_tt=0.0, roughness_trt=0.0,
                        cuticle_angle=math.radians(2.0), ior=1.55, absorption=None, diffuse_weight=0.0, diffuse_tint=(1.0, 1.0, 1.0)):
        """The arguments of df::chiang_hair_bsdf in the fixed-layout block (include/strelka_hip.h, SKH_MAT_HAIR).  `color` is the
        fibre's multiple-scattering albedo; the absorption coefficient follows from it by Chiang et al. 2016, eq. 9 -- what a hair
        material's MDL code does in front of the distribution function -- unless `absorption` gives sigma_a directly."""
        if absorption is None:
            absorption = hair_sigma_a_from_color(color, roughness_n)
        return self.addMaterial(MAT_HAIR, diffuse_tint, roughness=roughness_r, metallic=roughness_tt, specular=roughness_trt, ior=ior,
                                reserved=(absorption[0], absorption[1], absorption[2], roughness_n, cuticle_angle, diffuse_weight))

    def addTexture(self, rgba8):
        """RGBA8 image, rows top to bottom as stbi_load returns them (OptixRender.cpp:1191-1264).  Returns the texture ID
        materials refer to (1-based; 0 = no texture, as MDL's invalid texture)."""
        t = np.ascontiguousarray(rgba8, np.uint8)
        assert t.ndim == 3 and t.shape[2] == 4 and t.shape[0] > 0 and t.shape[1] > 0
        self.mTextures.append(t)
        return len(self.mTextures)

    def addCamera(self, camera):
        self.mCameras.append(camera)
        return len(self.mCameras) - 1

    def getCamera(self, index=0):
        return self.mCameras[index]

    # -- createCurve (scene.h declaration :399-402; HdStrelka/BasisCurves.cpp:189-232 supplies phantom points + radii)
    def createCurve(self, vertexCounts, points, widths):
        vertexCounts = np.ascontiguousarray(vertexCounts, np.uint32)
        points = np.ascontiguousarray(points, np.float32).reshape(-1, 3)
        widths = np.ascontiguousarray(widths, np.float32).reshape(-1)
        cid = len(self.mCurves)
        self.mCurves.append((self._nvc, len(vertexCounts), self._npoints, len(points), self._npoints, len(widths)))
        self.mCurveVertexCounts.append(vertexCounts)
        self.mCurvePoints.append(points)
        self.mCurveWidths.append(widths)
        self._nvc += len(vertexCounts)
        self._npoints += len(points)
        return cid

    # -- light proxy meshes: scene.cpp:119-250
    def _createRectLightMesh(self):
        if self.mRectLightMeshId != -1:
            return self.mRectLightMeshId
        vb = np.zeros(4, VERTEX)
        vb["pos"] = [(0.5, 0.5, 0), (-0.5, 0.5, 0), (-0.5, -0.5, 0), (0.5, -0.5, 0)]
        vb["normal"] = pack_normals([(0, 0, 1)] * 4)
        self.mRectLightMeshId = self.createMesh(vb, [0, 1, 2, 2, 3, 0])
        return self.mRectLightMeshId

    def _createSphereLightMesh(self):
        if self.mSphereLightMeshId != -1:
            return self.mSphereLightMeshId
        segments = rings = 16
        pos = []
        for i in range(rings + 1):
            theta = np.float32(i) * np.float32(math.pi) / np.float32(rings)
            st, ct = np.float32(math.sin(theta)), np.float32(math.cos(theta))
            for j in range(segments + 1):
                phi = np.float32(j) * np.float32(2.0) * np.float32(math.pi) / np.float32(segments)
                sp, cp = np.float32(math.sin(phi)), np.float32(math.cos(phi))
                pos.append((cp * st, ct, sp * st))
        vb = np.zeros(len(pos), VERTEX)
        vb["pos"] = pos
        vb["normal"] = pack_normals(pos)
        ib = []
        for i in range(rings):
            for j in range(segments):
                p0 = i * (segments + 1) + j
                p1, p2 = p0 + 1, (i + 1) * (segments + 1) + j
                p3 = p2 + 1
                ib += [p0, p1, p2, p2, p1, p3]
        self.mSphereLightMeshId = self.createMesh(vb, ib)
        return self.mSphereLightMeshId

    def _createDiscLightMesh(self):
        if self.mDiskLightMeshId != -1:
            return self.mDiskLightMeshId
        pos = [(0, 0, 0), (1, 0, 0)]
        ib = []
        step = np.float32(2.0 * math.pi / 16)
        angle = np.float32(0)
        for _ in range(16):
            ib += [0, len(pos) - 1]
            angle = np.float32(angle + step)
            pos.append((math.cos(angle), math.sin(angle), 0.0))
            ib.append(len(pos) - 1)
        vb = np.zeros(len(pos), VERTEX)
        vb["pos"] = pos
        vb["normal"] = pack_normals([(0, 0, 1)] * len(pos))
        self.mDiskLightMeshId = self.createMesh(vb, ib)
        return self.mDiskLightMeshId

    @staticmethod
    def _light_transform(desc):  # Scene::getTransform(desc) scene.h:331-343
        t = translate(desc.get("position", (0, 0, 0)))
        r = quat_to_mat4(quat_from_euler_deg(desc.get("orientation", (0, 0, 0))))
        s = scale((desc.get("width", 1.0), desc.get("height", 1.0), 1.0))
        return t @ r @ s

    # -- Scene::createLight + updateLight: scene.cpp:306-408.  desc keys follow UniformLightDesc (scene.h:157-180)
    def createLight(self, desc):
        light_id = len(self.mLights)
        typ = int(desc["type"])
        use_xform = bool(desc.get("useXform", "xform" in desc))
        xform = np.asarray(desc.get("xform", np.eye(4)), np.float64).reshape(4, 4)
        radius = float(desc.get("radius", 0.0))
        L = np.zeros((), LIGHT)
        L["color"] = 1.0
        if typ == 0:
            sm = scale((desc["width"], desc["height"], 1.0))
            lt = xform @ sm if use_xform else self._light_transform(desc)
            for k, c in enumerate([(0.5, 0.5, 0, 1), (-0.5, 0.5, 0, 1), (-0.5, -0.5, 0, 1), (0.5, -0.5, 0, 1)]):
                L["points"][k] = (lt @ np.array(c, np.float64)).astype(np.float32)
            L["type"] = 0
            mesh, sm_inst = self._createRectLightMesh(), sm
        elif typ == 1:
            sm = scale((radius, radius, radius))
            lt = xform @ sm if use_xform else self._light_transform(desc)
            L["points"][0] = (radius, 0, 0, 0)
            L["points"][1] = (lt @ np.array([0, 0, 0, 1.0])).astype(np.float32)
            L["points"][2] = (lt @ np.array([1.0, 0, 0, 0])).astype(np.float32)
            L["points"][3] = (lt @ np.array([0, 1.0, 0, 0])).astype(np.float32)
            L["normal"] = (lt @ np.array([0, 0, 1.0, 0])).astype(np.float32)
            L["type"] = 1
            mesh, sm_inst = self._createDiscLightMesh(), sm
        elif typ == 2:
            lt = xform if use_xform else self._light_transform(desc)
            L["points"][0] = (radius, 0, 0, 0)
            L["points"][1] = (lt @ np.array([0, 0, 0, 1.0])).astype(np.float32)
            L["type"] = 2
            mesh, sm_inst = self._createSphereLightMesh(), scale((radius, radius, radius))
        elif typ == 3:
            L["type"] = 3
            L["half_angle"] = desc["halfAngle"]
            lt = xform if use_xform else self._light_transform(desc)
            n = lt @ np.array([0, 0, -1.0, 0])
            L["normal"] = (n / np.linalg.norm(n)).astype(np.float32)
            # scene.cpp:337-345: a light instance of MESH 0 scaled by desc.radius (0 for distant lights):
            # degenerate and unhittable, kept for index parity with the reference
            mesh, sm_inst = 0, scale((radius, radius, radius))
        else:
            raise ValueError("unknown light type")
        col = np.asarray(desc.get("color", (1, 1, 1)), np.float32)
        L["color"] = np.append(col, np.float32(1.0)) * np.float32(desc.get("intensity", 1.0))
        self.mLights.append(L)
        self.mLightDesc.append(dict(desc))
        transform = (xform @ sm_inst) if use_xform else self._light_transform(desc)
        self.createInstance(INSTANCE_LIGHT, mesh, NO_ID, transform, light_id)
        return light_id

    # -- flat getters (scene.h:229-327) in C-ABI layouts
    def arrays(self):
        def cat(lst, dtype, shape=None):
            if not lst:
                return np.zeros((0,) + (shape or ()), dtype)
            return np.ascontiguousarray(np.concatenate(lst), dtype=dtype)

        meshes = np.zeros(len(self.mMeshes), MESH)
        for i, m in enumerate(self.mMeshes):
            meshes[i] = m
        curves = np.zeros(len(self.mCurves), CURVE)
        for i, c in enumerate(self.mCurves):
            curves[i] = c
        inst = np.zeros(len(self.mInstances), INSTANCE)
        for i, (t, ty, g, m, l) in enumerate(self.mInstances):
            inst[i] = (t, ty, g, m, l)
        lights = np.zeros(len(self.mLights), LIGHT)
        for i, l in enumerate(self.mLights):
            lights[i] = l
        mats = np.zeros(max(1, len(self.mMaterials)), MATERIAL)
        if not self.mMaterials:  # material 0 = default.mdl::default_material (OptixRender.cpp:1090-1097)
            mats[0]["base_color"] = 0.8
        for i, (ty, bc, r, me, sp, ior, bt, nt, rsv) in enumerate(self.mMaterials):
            mats[i]["type"], mats[i]["base_color"], mats[i]["roughness"] = ty, bc, r
            mats[i]["metallic"], mats[i]["specular"], mats[i]["ior"] = me, sp, ior
            mats[i]["base_color_texture"], mats[i]["normal_texture"] = bt, nt
            mats[i]["reserved"] = rsv
        return {
            "vertices": cat(self.mVertices, VERTEX),
            "indices": cat(self.mIndices, np.uint32),
            "meshes": meshes,
            "curves": curves,
            "curve_points": cat(self.mCurvePoints, np.float32, (3,)),
            "curve_radii": cat(self.mCurveWidths, np.float32),
            "curve_vertex_counts": cat(self.mCurveVertexCounts, np.uint32),
            "instances": inst,
            "lights": lights,
            "materials": mats,
            "textures": list(self.mTextures),
        }


def hair_sigma_a_from_color(color, roughness_n=0.3):
    """Absorption coefficient that gives a fibre the multiple-scattering albedo `color` at azimuthal roughness beta_n:
    Chiang et al. 2016, eq. 9 (pbrt-v3 HairBSDF::SigmaAFromReflectance)."""
    b = float(roughness_n)
    d = 5.969 - 0.215 * b + 2.532 * b ** 2 - 10.73 * b ** 3 + 5.574 * b ** 4 + 0.245 * b ** 5
    c = np.clip(np.asarray(color, np.float64), 1e-4, 1.0)
    return tuple(float(x) for x in (np.log(c) / d) ** 2)


def pack_textures(textures):
    """list of HxWx4 uint8 images -> (TEXTURE_DESC array, uint32 texel array) in the packed form the renderer keeps on the device"""
    desc = np.zeros(len(textures), TEXTURE_DESC)
    off, parts = 0, []
    for k, t in enumerate(textures):
        t = np.ascontiguousarray(t, np.uint8)
        desc[k] = (off, t.shape[1], t.shape[0], 0)
        parts.append(t.reshape(-1, 4).view(np.uint32).reshape(-1))
        off += t.shape[0] * t.shape[1]
    texels = np.concatenate(parts) if parts else np.zeros(0, np.uint32)
    return desc, np.ascontiguousarray(texels, np.uint32)


def make_vertices(positions, normals=None, uvs=None, tangents=None):
    """Pack float attributes into the 32-byte Vertex the way HdStrelka's _BakeMeshInstance does
    (RenderPass.cpp:69-130): normal/tangent 10-10-10, uv 16-16 with V flipped by the caller."""
    positions = np.asarray(positions, np.float32).reshape(-1, 3)
    vb = np.zeros(len(positions), VERTEX)
    vb["pos"] = positions
    if normals is not None:
        vb["normal"] = pack_normals(normals)
    if tangents is not None:
        vb["tangent"] = pack_normals(tangents)
    if uvs is not None:
        vb["uv"] = pack_uv(uvs)
    return vb


def deindex(positions, tris):
    """HdStrelkaMesh::_UpdateGeometry (Mesh.cpp:123-179): 3 fresh vertices per triangle, flat face normals when no
    primvar is authored, tangent = cross(n, X|Y).  Returns (vertices, indices 0..3T-1)."""
    positions = np.asarray(positions, np.float32)
    tris = np.asarray(tris, np.int64).reshape(-1, 3)
    p = positions[tris.reshape(-1)]
    p3 = p.reshape(-1, 3, 3).astype(np.float64)
    fn = np.cross(p3[:, 1] - p3[:, 0], p3[:, 2] - p3[:, 0])
    ln = np.linalg.norm(fn, axis=1, keepdims=True)
    fn = fn / np.where(ln > 0, ln, 1.0)
    n = np.repeat(fn, 3, axis=0)
    ax = np.where(np.abs(n[:, :1]) < 0.9, np.array([[1.0, 0, 0]]), np.array([[0, 1.0, 0]]))
    t = np.cross(n, ax)
    lt = np.linalg.norm(t, axis=1, keepdims=True)
    t = t / np.where(lt > 0, lt, 1.0)
    vb = make_vertices(p, n, None, t)
    return vb, np.arange(len(p), dtype=np.uint32)


def frame_params(camera, width, height, subframe_index=0, samples_this_launch=1, spp_total=64, max_depth=4,
                 rect_light_sampling_method=0, exposure=None, enable_accumulation=1, debug=0, shadow_ray_tmin=0.0,
                 material_ray_tmin=0.0):
    """The Params fields OptiXRender::render fills per call (OptixRender.cpp:936-1004)."""
    camera.updateAspectRatio(width / float(height))
    camera.updateViewMatrix()
    p = np.zeros((), FRAME_PARAMS)
    p["view_to_world"] = camera.view_to_world_rowmajor()
    p["clip_to_view"] = camera.clip_to_view_rowmajor()
    p["subframe_index"], p["samples_this_launch"], p["spp_total"] = subframe_index, samples_this_launch, spp_total
    p["max_depth"], p["rect_light_sampling_method"] = max_depth, rect_light_sampling_method
    p["exposure"] = default_exposure() if exposure is None else exposure
    p["enable_accumulation"], p["debug"] = enable_accumulation, debug
    p["shadow_ray_tmin"], p["material_ray_tmin"] = shadow_ray_tmin, material_ray_tmin
    return p


def default_exposure(filmIso=100.0, cm2_factor=1.0, fStop=4.0, shutterSpeed=100.0):
    """exposureValue of OptiXRender::render (OptixRender.cpp:961-987), fp32 arithmetic."""
    f = np.float32
    e = np.array([1.0, 1.0, 1.0], f)
    lum = f(f(e[0] * f(0.299) + e[1] * f(0.587)) + e[2] * f(0.114))
    if filmIso > 0.0:
        e = e * f(f(f(f(cm2_factor) * f(filmIso)) / f(f(f(shutterSpeed) * f(fStop)) * f(fStop))) / f(100.0))
    else:
        e = e * f(cm2_factor)
    inv = f(1.0) / lum
    return (e * inv).astype(f)
