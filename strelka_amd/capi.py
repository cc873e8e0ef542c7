"""ctypes binding of the C ABI in include/strelka_hip.h (libstrelka_hip.so, HIP / gfx950).

This is the only way Python reaches the renderer: there is NO CPU fallback.  Loading fails loudly when the library
has not been built (strelka_amd.build) and skh_create fails loudly when no GPU is visible.
"""
import ctypes as C
import os

import numpy as np

from . import scene as S
from .build import LIB

_lib = None

STATS = np.dtype([("rays_radiance", np.uint64), ("rays_shadow", np.uint64), ("nodes_visited", np.uint64, 2),
                  ("prims_tested", np.uint64, 2), ("segs_tested", np.uint64, 2), ("instances_entered", np.uint64, 2),
                  ("ms_trace_closest", np.float64), ("ms_trace_shadow", np.float64), ("ms_shade", np.float64),
                  ("ms_raygen", np.float64), ("ms_accumulate", np.float64), ("ms_build", np.float64), ("ms_sort", np.float64),
                  ("launches_trace_closest", np.uint32), ("launches_trace_shadow", np.uint32),
                  ("launches_shade", np.uint32), ("launches_other", np.uint32), ("stack_overflows", np.uint32),
                  ("speculated_discarded", np.uint32)])

SYMBOLS = ["skh_create", "skh_destroy", "skh_last_error", "skh_abi_version", "skh_set_geometry", "skh_set_curves",
           "skh_set_instances", "skh_set_lights", "skh_set_textures", "skh_set_materials", "skh_build_accel", "skh_resize", "skh_set_tiles",
           "skh_render_subframe", "skh_render_subframes", "skh_tonemap", "skh_read_accum", "skh_read_aov",
           "skh_buffer_alloc", "skh_buffer_free", "skh_buffer_download", "skh_copy_accum", "skh_copy_accum_tiles", "skh_scatter_tiles", "skh_trace", "skh_trace_device",
           "skh_set_option", "skh_get_stats", "skh_reset_stats", "skh_synchronize", "skh_get_stream", "skh_bsdf_probe", "skh_get_device_info", "skh_comm_unique_id", "skh_comm_init",
           "skh_comm_destroy", "skh_gather_tiles", "skh_host_register", "skh_host_unregister", "skh_get_baked", "skh_comm_info", "skh_probe_memory", "skh_unit_probe", "skh_copy_aov", "skh_get_build_info", "skh_refit_accel"]

BUILD_INFO = np.dtype([("triangles", np.uint32), ("nodes", np.uint32), ("reinsert_rounds", np.uint32), ("reinsert_moves", np.uint32),
                       ("reinsert_min_size", np.uint32), ("refit", np.uint32), ("cost_before", np.float64), ("cost_after", np.float64),
                       ("ms_reinsert", np.float64), ("ms_build", np.float64), ("ms_refit", np.float64)])
DEVICE_INFO = np.dtype([("compute_units", np.uint32), ("simds_per_cu", np.uint32), ("clock_khz", np.uint32), ("memory_clock_khz", np.uint32),
                        ("memory_bus_bits", np.uint32), ("wavefront_size", np.uint32), ("total_memory_bytes", np.uint64), ("name", "S64")])

BSDF_QUERY = np.dtype([("normal", np.float32, 3), ("geom_normal", np.float32, 3), ("tangent_u", np.float32, 3), ("k1", np.float32, 3),
                       ("k2", np.float32, 3), ("xi", np.float32, 4), ("material", np.uint32), ("inside", np.uint32)])
BSDF_RESULT = np.dtype([("k2", np.float32, 3), ("bsdf_over_pdf", np.float32, 3), ("pdf", np.float32), ("event_type", np.int32),
                        ("bsdf_diffuse", np.float32, 3), ("bsdf_glossy", np.float32, 3), ("eval_pdf", np.float32), ("reserved0", np.uint32)])
assert BSDF_QUERY.itemsize == 84 and BSDF_RESULT.itemsize == 64


class SkhError(RuntimeError):
    pass


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB):
        raise SkhError(f"{LIB} is missing: build it with `python -m strelka_amd.build` (hipcc, gfx950). "
                       "There is no CPU fallback.")
    # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64; when torch is used in the same process
    # (device tensors for d_image, torch.distributed/RCCL) it must be the copy that gets loaded first, otherwise
    # torch later reports "No HIP GPUs are available".  torch is plumbing here, not a dependency of the kernels.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    lib = C.CDLL(LIB)
    vp, u32, i32, f32 = C.c_void_p, C.c_uint32, C.c_int, C.c_float
    lib.skh_create.argtypes = [i32, C.POINTER(vp)]
    lib.skh_destroy.argtypes = [vp]
    lib.skh_destroy.restype = None
    lib.skh_last_error.argtypes = [vp]
    lib.skh_last_error.restype = C.c_char_p
    lib.skh_abi_version.restype = u32
    lib.skh_set_geometry.argtypes = [vp, vp, u32, vp, u32, vp, u32]
    lib.skh_set_curves.argtypes = [vp, vp, u32, vp, u32, vp, u32, vp, u32]
    for n in ("skh_set_instances", "skh_set_lights", "skh_set_textures", "skh_set_materials"):
        getattr(lib, n).argtypes = [vp, vp, u32]
    lib.skh_build_accel.argtypes = [vp, u32]
    lib.skh_refit_accel.argtypes = [vp]
    lib.skh_resize.argtypes = [vp, u32, u32]
    lib.skh_set_tiles.argtypes = [vp, u32, vp, u32]
    lib.skh_render_subframe.argtypes = [vp, vp, vp]
    lib.skh_render_subframes.argtypes = [vp, vp, u32, vp]
    lib.skh_tonemap.argtypes = [vp, vp, u32, u32, u32, vp, f32]
    lib.skh_buffer_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    lib.skh_buffer_free.argtypes = [vp, vp]
    lib.skh_buffer_download.argtypes = [vp, vp, vp, C.c_size_t]
    lib.skh_read_accum.argtypes = [vp, vp]
    lib.skh_read_aov.argtypes = [vp, u32, vp]
    lib.skh_copy_accum.argtypes = [vp, vp]
    lib.skh_copy_aov.argtypes = [vp, u32, vp]
    lib.skh_copy_accum_tiles.argtypes = [vp, vp]
    lib.skh_scatter_tiles.argtypes = [vp, vp, vp, u32, u32, vp, u32, u32]
    lib.skh_trace.argtypes = [vp, vp, u32, u32, vp]
    lib.skh_trace_device.argtypes = [vp, vp, u32, u32, vp, u32]
    lib.skh_bsdf_probe.argtypes = [vp, vp, u32, vp]
    lib.skh_unit_probe.argtypes = [vp, u32, u32, vp, vp, u32, vp]
    lib.skh_get_device_info.argtypes = [vp, vp]
    lib.skh_get_build_info.argtypes = [vp, vp]
    lib.skh_host_register.argtypes = [vp, vp, C.c_size_t]
    lib.skh_host_unregister.argtypes = [vp, vp]
    lib.skh_comm_unique_id.argtypes = [vp]
    lib.skh_comm_init.argtypes = [vp, vp, i32, i32]
    lib.skh_comm_destroy.argtypes = [vp]
    lib.skh_comm_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    lib.skh_probe_memory.argtypes = [vp, u32, C.c_uint64, u32, u32, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.skh_gather_tiles.argtypes = [vp, u32, vp, i32]
    lib.skh_set_option.argtypes = [vp, C.c_char_p, C.c_int64]
    lib.skh_get_baked.argtypes = [vp, vp, u32, C.POINTER(u32), C.POINTER(u32)]
    lib.skh_get_stats.argtypes = [vp, vp]
    lib.skh_reset_stats.argtypes = [vp]
    lib.skh_synchronize.argtypes = [vp]
    lib.skh_get_stream.argtypes = [vp]
    lib.skh_get_stream.restype = vp
    for n in SYMBOLS:
        if n not in ("skh_destroy", "skh_last_error", "skh_abi_version", "skh_get_stream"):
            getattr(lib, n).restype = i32
    _lib = lib
    return lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Context:
    """One renderer context per GPU (skh_context).  Method names follow the C ABI."""

    def __init__(self, device=0):
        self.lib = load()
        h = C.c_void_p()
        st = self.lib.skh_create(device, C.byref(h))
        if st != 0 or not h:
            raise SkhError(f"skh_create(device={device}) failed with status {st}: no usable MI355X/HIP device. "
                           "This renderer has no CPU fallback.")
        self.h = h
        self.width = self.height = 0
        self.tile_size = 32
        self.tile_xy = None

    def close(self):
        if getattr(self, "h", None):
            self.lib.skh_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, st, what):
        if st != 0:
            raise SkhError(f"{what} failed ({st}): {self.lib.skh_last_error(self.h).decode()}")

    def set_scene(self, arr, build=True, flags=0):
        L = self.lib
        self._ck(L.skh_set_geometry(self.h, _p(arr["vertices"]), len(arr["vertices"]), _p(arr["indices"]),
                                    len(arr["indices"]), _p(arr["meshes"]), len(arr["meshes"])), "skh_set_geometry")
        if len(arr.get("curves", [])):
            self._ck(L.skh_set_curves(self.h, _p(arr["curve_points"]), len(arr["curve_points"]), _p(arr["curve_radii"]),
                                      len(arr["curve_radii"]), _p(arr["curve_vertex_counts"]),
                                      len(arr["curve_vertex_counts"]), _p(arr["curves"]), len(arr["curves"])),
                     "skh_set_curves")
        self._ck(L.skh_set_instances(self.h, _p(arr["instances"]), len(arr["instances"])), "skh_set_instances")
        self._ck(L.skh_set_lights(self.h, _p(arr["lights"]), len(arr["lights"])), "skh_set_lights")
        self.set_textures(arr.get("textures") or [])
        self._ck(L.skh_set_materials(self.h, _p(arr["materials"]), len(arr["materials"])), "skh_set_materials")
        if build:
            self.build_accel(flags)

    def set_textures(self, textures):
        """textures: list of HxWx4 uint8 images (rows top to bottom).  skh_texture = {const uint8_t* rgba8; u32 width, height}."""

        class SkhTexture(C.Structure):
            _fields_ = [("rgba8", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32)]

        keep = [np.ascontiguousarray(t, np.uint8) for t in textures]
        for t in keep:
            if t.ndim != 3 or t.shape[2] != 4:
                raise ValueError("textures must be HxWx4 uint8")
        recs = (SkhTexture * max(1, len(keep)))()
        for k, t in enumerate(keep):
            recs[k].rgba8, recs[k].width, recs[k].height = t.ctypes.data, t.shape[1], t.shape[0]
        self._ck(self.lib.skh_set_textures(self.h, C.cast(recs, C.c_void_p), len(keep)), "skh_set_textures")

    def build_accel(self, flags=0):
        self._ck(self.lib.skh_build_accel(self.h, flags), "skh_build_accel")

    def resize(self, w, h):
        self.width, self.height = w, h
        self._ck(self.lib.skh_resize(self.h, w, h), "skh_resize")

    def set_tiles(self, tile_size, tile_xy=None):
        self.tile_size = tile_size
        if tile_xy is not None:
            tile_xy = np.ascontiguousarray(tile_xy, np.uint32).reshape(-1, 2)
        self.tile_xy = tile_xy
        self._ck(self.lib.skh_set_tiles(self.h, tile_size, _p(tile_xy), 0 if tile_xy is None else len(tile_xy)),
                 "skh_set_tiles")

    def render_subframe(self, params, d_image=None):
        p = np.ascontiguousarray(params, dtype=S.FRAME_PARAMS)
        self._ck(self.lib.skh_render_subframe(self.h, _p(p), d_image), "skh_render_subframe")

    def render_subframes(self, params, n, d_image=None):
        p = np.ascontiguousarray(params, dtype=S.FRAME_PARAMS)
        self._ck(self.lib.skh_render_subframes(self.h, _p(p), n, d_image), "skh_render_subframes")

    def read_accum(self):
        out = np.zeros((self.height, self.width, 4), np.float32)
        self._ck(self.lib.skh_read_accum(self.h, _p(out)), "skh_read_accum")
        return out

    def read_aov(self, which):
        out = np.zeros((self.height, self.width, 4), np.float32)
        self._ck(self.lib.skh_read_aov(self.h, which, _p(out)), "skh_read_aov")
        return out

    def copy_accum(self, d_dst):
        self._ck(self.lib.skh_copy_accum(self.h, d_dst), "skh_copy_accum")

    def copy_aov(self, which, d_dst):
        """the diffuse (0) / specular (1) AOV accumulator -> a device image (OptixRender.cpp:1029-1042)"""
        self._ck(self.lib.skh_copy_aov(self.h, which, d_dst), "skh_copy_aov")

    def copy_accum_tiles(self, d_dst):
        self._ck(self.lib.skh_copy_accum_tiles(self.h, d_dst), "skh_copy_accum_tiles")

    def scatter_tiles(self, d_src, tile_xy, tile_size, d_dst, width, height):
        t = np.ascontiguousarray(tile_xy, np.uint32).reshape(-1, 2)
        self._ck(self.lib.skh_scatter_tiles(self.h, d_src, _p(t), len(t), tile_size, d_dst, width, height),
                 "skh_scatter_tiles")

    def buffer_download(self, d_src, host):
        """Buffer::map(): device -> host copy of a caller-owned device buffer into the numpy array `host`."""
        self._ck(self.lib.skh_buffer_download(self.h, d_src, _p(host), host.nbytes), "skh_buffer_download")

    def host_register(self, host):
        """page-lock a numpy array that buffer_download will fill (Buffer::map's host mirror)"""
        self._ck(self.lib.skh_host_register(self.h, _p(host), host.nbytes), "skh_host_register")

    def host_unregister(self, host):
        self._ck(self.lib.skh_host_unregister(self.h, _p(host)), "skh_host_unregister")

    def tonemap(self, d_image, width, height, type_, exposure, gamma):
        e = np.ascontiguousarray(exposure, np.float32)
        self._ck(self.lib.skh_tonemap(self.h, d_image, width, height, type_, _p(e), gamma), "skh_tonemap")

    def trace(self, rays, mode=0):
        rays = np.ascontiguousarray(rays, dtype=S.RAY)
        hits = np.zeros(len(rays), S.HIT)
        self._ck(self.lib.skh_trace(self.h, _p(rays), len(rays), mode, _p(hits)), "skh_trace")
        return hits

    def trace_device(self, d_rays, n, mode, d_hits, repeat=1):
        self._ck(self.lib.skh_trace_device(self.h, d_rays, n, mode, d_hits, repeat), "skh_trace_device")

    def set_materials(self, materials):
        m = np.ascontiguousarray(materials, S.MATERIAL).reshape(-1)
        self._ck(self.lib.skh_set_materials(self.h, _p(m), len(m)), "skh_set_materials")

    def bsdf_probe(self, queries):
        """mdlcode_sample + mdlcode_evaluate on the device for each BSDF_QUERY record -> BSDF_RESULT records"""
        q = np.ascontiguousarray(queries, BSDF_QUERY)
        out = np.zeros(len(q), BSDF_RESULT)
        self._ck(self.lib.skh_bsdf_probe(self.h, _p(q), len(q), _p(out)), "skh_bsdf_probe")
        return out

    # skh_unit: (words in, words out) per record
    UNITS = {"sampler": (0, 5, 3), "sobol": (1, 2, 1), "light_sample": (2, 5, 12), "light_pdf": (3, 6, 1), "light_normal": (4, 3, 4),
             "mis": (5, 2, 1), "accumulate": (6, 3, 3), "tonemap": (7, 3, 6), "libm": (8, 2, 10)}

    def unit_probe(self, unit, records, param=0, consts=None):
        """skh_unit_probe: one device call of the named function per record (include/strelka_hip.h lists the record layouts);
        `records` is an (n, words_in) array of 32-bit words, the result an (n, words_out) uint32 array (view it as float32)."""
        uid, win, wout = self.UNITS[unit]
        r = np.ascontiguousarray(records).reshape(-1, win)
        assert r.dtype.itemsize == 4
        out = np.zeros((len(r), wout), np.uint32)
        cst = None if consts is None else np.ascontiguousarray(consts)
        self._ck(self.lib.skh_unit_probe(self.h, uid, int(param), None if cst is None else _p(cst), _p(r), len(r), _p(out)), "skh_unit_probe")
        return out

    @staticmethod
    def comm_unique_id():
        """rank 0: the 128 bytes every rank hands to comm_init (distribute them with whatever the host has)"""
        buf = np.zeros(128, np.uint8)
        st = load().skh_comm_unique_id(_p(buf))
        if st != 0:
            raise SkhError(f"skh_comm_unique_id failed ({st}): RCCL (librccl.so.1) is not available")
        return buf

    def comm_init(self, unique_id, world_size, rank):
        u = np.ascontiguousarray(unique_id, np.uint8)
        assert u.nbytes == 128
        self._ck(self.lib.skh_comm_init(self.h, _p(u), world_size, rank), "skh_comm_init")

    def comm_info(self):
        """(world size, rank, ranks RCCL itself reports for the communicator -- ncclCommCount; 0 = no communicator)"""
        w, r, n = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        self._ck(self.lib.skh_comm_info(self.h, C.byref(w), C.byref(r), C.byref(n)), "skh_comm_info")
        return w.value, r.value, n.value

    def probe_memory(self, kind, nbytes, record_bytes=64, repeat=3):
        """measured memory ceiling: kind 0 stream copy, 1 independent random record fetches, 2 dependent ones -> (GB/s, ms)"""
        g, ms = C.c_double(0), C.c_double(0)
        self._ck(self.lib.skh_probe_memory(self.h, kind, int(nbytes), record_bytes, repeat, C.byref(g), C.byref(ms)), "skh_probe_memory")
        return g.value, ms.value

    def comm_destroy(self):
        self._ck(self.lib.skh_comm_destroy(self.h), "skh_comm_destroy")

    def gather_tiles(self, max_tiles, d_recv=None, root=0):
        """one RCCL gather of this context's tile accumulators into the root's [world][max_tiles][T*T] float4 buffer"""
        self._ck(self.lib.skh_gather_tiles(self.h, max_tiles, d_recv, root), "skh_gather_tiles")

    def device_info(self):
        d = np.zeros((), DEVICE_INFO)
        self._ck(self.lib.skh_get_device_info(self.h, _p(d)), "skh_get_device_info")
        return {k: (d[k].item().decode() if k == "name" else int(d[k])) for k in DEVICE_INFO.names}

    def set_geometry(self, scene):
        """skh_set_geometry alone (a vertex edit: follow it with refit_accel)"""
        v, idx, m = scene["vertices"], scene["indices"], scene["meshes"]
        self._ck(self.lib.skh_set_geometry(self.h, _p(v), len(v), _p(idx), len(idx), _p(m), len(m)), "skh_set_geometry")

    def set_curves(self, scene):
        """skh_set_curves alone (edited control points / radii: follow it with refit_accel)"""
        self._ck(self.lib.skh_set_curves(self.h, _p(scene["curve_points"]), len(scene["curve_points"]), _p(scene["curve_radii"]), len(scene["curve_radii"]),
                                         _p(scene["curve_vertex_counts"]), len(scene["curve_vertex_counts"]), _p(scene["curves"]), len(scene["curves"])), "skh_set_curves")

    def refit_accel(self):
        """skh_refit_accel: keep the hierarchy's topology, recompute leaf records and boxes from the current vertices (falls back to a build)"""
        self._ck(self.lib.skh_refit_accel(self.h), "skh_refit_accel")

    def build_info(self):
        """what the last skh_build_accel did to the triangle hierarchy (reinsertion rounds / moves, cost before and after)"""
        d = np.zeros((), BUILD_INFO)
        self._ck(self.lib.skh_get_build_info(self.h, _p(d)), "skh_get_build_info")
        return {k: (float(d[k]) if d[k].dtype.kind == "f" else int(d[k])) for k in BUILD_INFO.names if k != "reserved"}

    def baked(self, n_instances):
        """(per-instance flags, baked instances, baked triangles) of option bake_world after the build"""
        flags = np.zeros(max(1, n_instances), np.uint8)
        ni, nt = C.c_uint32(0), C.c_uint32(0)
        self._ck(self.lib.skh_get_baked(self.h, _p(flags), n_instances, C.byref(ni), C.byref(nt)), "skh_get_baked")
        return flags[:n_instances], ni.value, nt.value

    def set_option(self, name, value):
        self._ck(self.lib.skh_set_option(self.h, name.encode(), int(value)), f"skh_set_option({name})")

    def stats(self):
        s = np.zeros((), STATS)
        self._ck(self.lib.skh_get_stats(self.h, _p(s)), "skh_get_stats")
        out = {}
        for k in STATS.names:
            v = s[k]
            if v.ndim:
                out[k] = [int(x) for x in v]
            else:
                out[k] = float(v) if v.dtype.kind == "f" else int(v)
        return out

    def reset_stats(self):
        self._ck(self.lib.skh_reset_stats(self.h), "skh_reset_stats")

    def synchronize(self):
        self._ck(self.lib.skh_synchronize(self.h), "skh_synchronize")

    def stream(self):
        return self.lib.skh_get_stream(self.h)
