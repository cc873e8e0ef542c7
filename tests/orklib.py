"""ctypes loader for the CPU ORACLE (oracle/liboracle.so).  Test infrastructure only: imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the product package."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "liboracle.so")

_lib = None


def build(force=False):
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("oracle.cpp", "ork_math.h", "ork_core.h", "ork_trace.h", "ork_bsdf.h")]
    srcs.append(os.path.join(ROOT, "strelka_amd", "csrc", "skh_libm.h"))  # ork_math.h includes it: a libm edit must rebuild the checker
    if force or not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"], stdout=subprocess.DEVNULL)


def fptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Oracle:
    """Thin object wrapper over the ork_* C API; mirrors strelka_amd.capi.Context method for method."""

    def __init__(self, lib):
        self.lib = lib
        self.ctx = C.c_void_p(lib.ork_create())
        self.width = self.height = 0

    def __del__(self):
        try:
            self.lib.ork_destroy(self.ctx)
        except Exception:
            pass

    def set_scene(self, scene):
        L = self.lib
        v, idx, m = scene["vertices"], scene["indices"], scene["meshes"]
        L.ork_set_geometry(self.ctx, fptr(v), len(v), fptr(idx), len(idx), fptr(m), len(m))
        if len(scene.get("curves", [])):
            L.ork_set_curves(self.ctx, fptr(scene["curve_points"]), len(scene["curve_points"]),
                             fptr(scene["curve_radii"]), len(scene["curve_radii"]),
                             fptr(scene["curve_vertex_counts"]), len(scene["curve_vertex_counts"]),
                             fptr(scene["curves"]), len(scene["curves"]))
        L.ork_set_instances(self.ctx, fptr(scene["instances"]), len(scene["instances"]))
        L.ork_set_lights(self.ctx, fptr(scene["lights"]), len(scene["lights"]))
        L.ork_set_materials(self.ctx, fptr(scene["materials"]), len(scene["materials"]))
        from strelka_amd import scene as S

        desc, texels = S.pack_textures(scene.get("textures") or [])
        if L.ork_set_textures(self.ctx, fptr(desc), len(desc), fptr(texels), C.c_uint64(len(texels))) != 0:
            raise ValueError("ork_set_textures: descriptor outside the texel array")
        L.ork_build_accel(self.ctx)

    def set_bake(self, mode, small_tris=64):
        """the product's bake_world / bake_small_tris options (part of the intersection's definition); call before set_scene"""
        self.lib.ork_set_bake(self.ctx, int(mode), int(small_tris))

    def baked(self, n_instances):
        out = np.zeros(max(1, n_instances), np.uint8)
        self.lib.ork_get_baked(self.ctx, fptr(out), n_instances)
        return out[:n_instances]

    def debug_path(self, params, px, py, sample_index, max_rays=64):
        """(diagnosis) the rays one path traces: rows {kind 0 radiance / 1 shadow, o xyz, tmin, d xyz, tmax, hit t, inst, prim}, and its radiance"""
        out, rad = np.zeros((max_rays, 12), np.float32), np.zeros(3, np.float32)
        n = self.lib.ork_debug_path(self.ctx, fptr(np.ascontiguousarray(params)), px, py, sample_index, fptr(out), max_rays, fptr(rad))
        return out[:n], rad

    def resize(self, w, h):
        self.width, self.height = w, h
        self.lib.ork_resize(self.ctx, w, h)

    def render_subframe(self, params, rows=None):
        p = np.ascontiguousarray(params)
        if rows is None:
            self.lib.ork_render_subframe(self.ctx, fptr(p))
        else:
            self.lib.ork_render_subframe_rows(self.ctx, fptr(p), rows[0], rows[1])

    def read_accum(self):
        out = np.empty((self.height, self.width, 4), np.float32)
        self.lib.ork_read_accum(self.ctx, fptr(out))
        return out

    def read_image(self):
        out = np.empty((self.height, self.width, 4), np.float32)
        self.lib.ork_read_image(self.ctx, fptr(out))
        return out

    def read_aov(self, which):
        out = np.empty((self.height, self.width, 4), np.float32)
        self.lib.ork_read_aov(self.ctx, which, fptr(out))
        return out

    def trace(self, rays, mode=0, brute=False):
        rays = np.ascontiguousarray(rays)
        hits = np.empty(len(rays), dtype=HIT_DTYPE)
        self.lib.ork_trace(self.ctx, fptr(rays), len(rays), mode, 1 if brute else 0, fptr(hits))
        return hits

    def stats(self):
        s = np.zeros(6, np.uint64)
        self.lib.ork_get_stats(self.ctx, fptr(s))
        return dict(zip(["rays_radiance", "rays_shadow", "nodes_visited", "prims_tested", "segs_tested",
                         "instances_entered"], [int(x) for x in s]))

    def reset_stats(self):
        self.lib.ork_reset_stats(self.ctx)

    def set_count_traversal(self, on):
        self.lib.ork_set_count_traversal(self.ctx, 1 if on else 0)


RAY_DTYPE = np.dtype([("origin", np.float32, 3), ("tmin", np.float32), ("dir", np.float32, 3), ("tmax", np.float32)])
HIT_DTYPE = np.dtype([("t", np.float32), ("instance_id", np.uint32), ("prim_id", np.uint32), ("u", np.float32),
                      ("v", np.float32)])


def usable_cpus():
    """CPUs this process may really use: the affinity mask AND the cgroup CPU quota (a GPU box shows 256 hardware threads to a
    container throttled to 16 CPUs: 256 OpenMP threads there run several times slower than 16)."""
    usable = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            usable = max(1, min(usable, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return usable


def _bind(lib):
    """argtypes / restypes of the ork_* C API, for whichever build of oracle.cpp `lib` is"""
    if not os.environ.get("OMP_NUM_THREADS"):
        lib.ork_set_num_threads(usable_cpus())
    lib.ork_create.restype = C.c_void_p
    lib.ork_destroy.argtypes = [C.c_void_p]
    lib.ork_libm.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    lib.ork_libm.restype = None
    lib.ork_mis_weight.restype = C.c_float
    lib.ork_mis_weight.argtypes = [C.c_float, C.c_float]
    for name in ("ork_set_geometry", "ork_set_curves", "ork_set_instances", "ork_set_lights", "ork_set_materials",
                 "ork_build_accel", "ork_resize", "ork_render_subframe", "ork_render_subframe_rows",
                 "ork_read_accum", "ork_read_image", "ork_read_aov", "ork_get_stats", "ork_reset_stats",
                 "ork_trace", "ork_set_count_traversal"):
        getattr(lib, name).restype = C.c_int
    lib.ork_set_geometry.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
    lib.ork_set_curves.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                                   C.c_void_p, C.c_uint32]
    for name in ("ork_set_instances", "ork_set_lights", "ork_set_materials"):
        getattr(lib, name).argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    lib.ork_build_accel.argtypes = [C.c_void_p]
    lib.ork_debug_path.restype = C.c_int
    lib.ork_debug_path.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]
    lib.ork_set_bake.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
    lib.ork_get_baked.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    lib.ork_resize.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
    lib.ork_render_subframe.argtypes = [C.c_void_p, C.c_void_p]
    lib.ork_render_subframe_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
    lib.ork_read_accum.argtypes = [C.c_void_p, C.c_void_p]
    lib.ork_read_image.argtypes = [C.c_void_p, C.c_void_p]
    lib.ork_read_aov.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    lib.ork_get_stats.argtypes = [C.c_void_p, C.c_void_p]
    lib.ork_reset_stats.argtypes = [C.c_void_p]
    lib.ork_set_count_traversal.argtypes = [C.c_void_p, C.c_int]
    lib.ork_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]
    lib.ork_sample_index.restype = C.c_uint32
    lib.ork_sample_index.argtypes = [C.c_uint32] * 4
    lib.ork_sampler_values.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                       C.c_uint32, C.c_uint32, C.c_void_p]
    lib.ork_sobol_matrix.argtypes = [C.c_void_p]
    lib.ork_sample_light.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    lib.ork_light_pdf.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    lib.ork_curve_eval.argtypes = [C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]
    lib.ork_accumulate_seq.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]
    lib.ork_tonemap_pair.argtypes = [C.c_void_p] * 4
    lib.ork_exposure.argtypes = [C.c_float] * 4 + [C.c_void_p]
    lib.ork_pack_normal.restype = C.c_uint32
    lib.ork_pack_normal.argtypes = [C.c_void_p]
    lib.ork_pack_uv.restype = C.c_uint32
    lib.ork_pack_uv.argtypes = [C.c_float, C.c_float]
    lib.ork_unpack_normal.argtypes = [C.c_uint32, C.c_void_p]
    lib.ork_unpack_uv.argtypes = [C.c_uint32, C.c_void_p]
    lib.ork_offset_ray.argtypes = [C.c_void_p] * 3
    lib.ork_camera_ray.argtypes = [C.c_uint32] * 4 + [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
    lib.ork_clip_to_view.argtypes = [C.c_float] * 4 + [C.c_void_p]
    lib.ork_invert_affine.argtypes = [C.c_void_p, C.c_void_p]
    lib.ork_tonemap_image.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_float]
    lib.ork_bsdf_sample.argtypes = [C.c_void_p] * 5 + [C.c_int, C.c_void_p]
    lib.ork_bsdf_evaluate.argtypes = [C.c_void_p] * 6
    lib.ork_bsdf_evaluate_side.argtypes = [C.c_void_p] * 5 + [C.c_int, C.c_void_p]
    lib.ork_bsdf_set_tangent.argtypes = [C.c_void_p]
    lib.ork_intersect_triangle.restype = C.c_int
    lib.ork_intersect_triangle.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
    lib.ork_intersect_curve.restype = C.c_int
    lib.ork_intersect_curve.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
    lib.ork_sizeof.restype = C.c_uint32
    lib.ork_sizeof.argtypes = [C.c_int]
    lib.ork_num_threads.restype = C.c_int
    return lib


_glibc = None


def load_glibc():
    """The SAME checker source built with glibc's sin / cos / acos / asin / atan2 / exp / log / sinh / pow instead of the text it shares with the
    product (oracle/ork_math.h, -DORK_LIBM_GLIBC; oracle/Makefile target liboracle_glibc.so): a second opinion on strelka_amd/csrc/skh_libm.h
    inside whole renders.  Images from it are compared at round 4's tolerances, never bit for bit."""
    global _glibc
    if _glibc is None:
        subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle_glibc.so"], stdout=subprocess.DEVNULL)
        _glibc = _bind(C.CDLL(os.path.join(ORACLE_DIR, "liboracle_glibc.so")))
    return _glibc


def load():
    global _lib
    if _lib is None:
        # ORK_LIB: another build of the same source (tools/oracle_sanitize.sh loads an ASan / UBSan build)
        if os.environ.get("ORK_LIB"):
            lib = C.CDLL(os.environ["ORK_LIB"])
        else:
            build()
            lib = C.CDLL(LIB)
        _lib = _bind(lib)
    return _lib


def new_context():
    return Oracle(load())
