#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE.json's config.

metric : Mray/s (+ ms/frame) at 1920x1080, 4 bounces, 64 spp, kitchen-class scene, full BSDF set
step   : one FRAME = 64 sub-frame launches of 1 spp (reference default spp=1: src/hdRunner/main.cpp:454,514),
         excluding scene upload, BVH build and D2H read-back; scene + BVH are resident in HBM
         when the timed region starts.
rays   : radiance segments + shadow rays ACTUALLY traced (device queue lengths), not W*H*spp*depth.
N > 1  : one process per GPU (torch.distributed, backend nccl = RCCL); the frame's pixel tiles are dealt round-robin
         to the ranks (no data-path collective), one gather of the tile accumulators to rank 0 per frame.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import re
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured copy)


def algorithmic_bytes(rays, shadow, nodes, prims, segs, insts):
    """bytes/ray = 36 + {20 | 4} + 64 N_node + 48 N_tri + 64 N_seg + 48 N_inst."""
    return rays * (36 + (4 if shadow else 20)) + 64 * nodes + 48 * prims + 64 * segs + 48 * insts


def cpu_baseline(arr, cam, width, height, spp_total, depth, budget_s=12.0):
    """The CPU oracle (kind "port": the reference has no CPU path, include/render/render.h:9-14 + render.cpp:10-35)
    timed on this box's host cores over a bounded sample: sub-frames 0, 1, ... in row bands until ~budget_s of work."""
    from strelka_amd import scene as S
    from tests import orklib

    usable = orklib.usable_cpus()  # affinity mask AND cgroup CPU quota
    orklib.load().ork_set_num_threads(usable)
    o = orklib.new_context()
    t0 = time.time()
    o.set_scene(arr)
    build_s = time.time() - t0
    o.resize(width, height)
    rows, t_used, sub, y = 8, 0.0, 0, 0
    while t_used < budget_s and sub < spp_total:
        p = S.frame_params(cam, width, height, subframe_index=sub, spp_total=spp_total, max_depth=depth)
        y = 0
        while y < height and t_used < budget_s:
            y1 = min(height, y + rows)
            t0 = time.time()
            o.render_subframe(p, rows=(y, y1))
            t_used += time.time() - t0
            y = y1
            rows = min(rows * 2, 128)
        sub += 1
    st = o.stats()
    rays = st["rays_radiance"] + st["rays_shadow"]
    return {"value": round(rays / t_used / 1e6, 4), "unit": "Mray/s", "cores": int(orklib.load().ork_num_threads()),
            "kind": "port",
            "sample": f"sub-frames 0..{sub - 1} of {spp_total} (1 spp each; the last one up to row {y} of {height}) at {width}x{height}, depth {depth}: "
                      f"{rays} rays in {t_used:.2f} s; oracle BVH build {build_s:.2f} s not included"}


PMC_RESULT = None  # filled by live_pmc() before this process touches the GPU
PMC_PASSES = (("FETCH_SIZE",), ("WRITE_SIZE",),  # TCC: the two do not fit one pass (MI355X_MICROARCH.md "rocprofv3 PMC slots")
              ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVE_CYCLES",
               "SQ_WAIT_INST_ANY"),
              ("GRBM_GUI_ACTIVE",))  # busy cycles of the graphics engine per XCD / the dispatch's duration = the shader clock the launch really ran at
# the three hot kernels of a frame (timed builds; the counting pass runs k_trace<.., true, ..>)
KERNELS = {"closest": "void skh::k_trace<false, false", "shadow": "void skh::k_trace<true, false", "shade": "void skh::k_shade<"}
CLOSEST = KERNELS["closest"]


def _pmc_per_launch(outdir, kernel_prefix):
    """{counter: average over the kernel's dispatches of the summed Counter_Value} from rocprofv3's counter_collection.csv"""
    import csv
    import glob
    from collections import defaultdict

    per = defaultdict(lambda: defaultdict(float))
    rows_of, span = defaultdict(int), {}
    for f in glob.glob(os.path.join(outdir, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Kernel_Name"].startswith(kernel_prefix):
                did = row.get("Dispatch_Id", "0")
                per[row["Counter_Name"]][did] += float(row["Counter_Value"])
                if row["Counter_Name"] == "GRBM_GUI_ACTIVE" and row.get("Start_Timestamp") and row.get("End_Timestamp"):
                    rows_of[did] += 1
                    span[did] = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
    out = {c: sum(d.values()) / len(d) for c, d in per.items() if d}
    clocks = []
    for did, total in per.get("GRBM_GUI_ACTIVE", {}).items():
        if span.get(did, 0) > 0:
            ghz = total / span[did]  # busy cycles per ns, summed over the engine instances the rows cover
            inst = rows_of[did] if rows_of[did] > 1 else (8 if ghz > 5.0 else 1)  # one row per XCD, or one row holding the sum of the 8
            clocks.append(ghz / inst)
    if clocks:
        out["__clock_ghz"] = sum(clocks) / len(clocks)
    return out


def pmc_figures(c, units_per_launch):
    """Counter averages of one launch -> the figures a roofline block quotes.  hbm bytes = (2 FETCH_SIZE + WRITE_SIZE) KiB:
    gfx950's FETCH_SIZE tallies 128-byte requests at 64 bytes (MI355X_MICROARCH.md "HBM"); calibrated on coalesced 16-byte
    per lane streams, which is what a node / triangle fetch is per lane -- but lanes scatter, so read the absolute as +-2x."""
    out = {"rays_per_launch": units_per_launch}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        out["hbm_bytes_per_launch"] = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
        out["FETCH_SIZE_KiB"], out["WRITE_SIZE_KiB"] = c["FETCH_SIZE"], c["WRITE_SIZE"]
    if c.get("SQ_INSTS_VALU"):
        out["valu_per_launch"] = c["SQ_INSTS_VALU"]
        out["salu_per_valu"] = round(c.get("SQ_INSTS_SALU", 0.0) / c["SQ_INSTS_VALU"], 3)
        if c.get("SQ_ACTIVE_INST_VALU"):
            out["lanes_per_valu_inst"] = round(c.get("SQ_THREAD_CYCLES_VALU", 0.0) / c["SQ_ACTIVE_INST_VALU"], 2)
        if c.get("SQ_WAVE_CYCLES"):
            out["wait_inst_any_frac"] = round(c.get("SQ_WAIT_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"], 3)
    if c.get("__clock_ghz"):
        out["clock_ghz_measured"] = round(c["__clock_ghz"], 4)
    return out


def under_profiler():
    """True when this process itself runs under rocprofv3 (its tool library is preloaded): nesting profilers is refused."""
    return any(k.startswith(("ROCP_", "ROCPROFILER_", "ROCPROF_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")


def live_pmc(argv, spp, keep_dir=None):
    """The HBM-side and SQ counters of the three hot kernels, measured NOW: this process (which has not touched the GPU yet) runs
    `rocprofv3 --pmc <counters> -- python3 bench.py --pmc-child ...` once per counter group as child processes -- one frame of
    one full batch each, so a launch there is a launch of the timed run -- and averages each kernel's dispatches.  Only --pmc,
    never combined with a trace domain; the children load the scene the parent cached and inherit no profiler variables.
    Returns None (and says why on stderr, keeping the logs) when rocprofv3 is missing, this process is itself profiled, or a
    pass fails: the committed profile is quoted instead and the record says "roofline_replayed"."""
    import shutil
    import subprocess
    import tempfile

    rocprof = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not rocprof:
        sys.stderr.write("[bench] live counters: rocprofv3 not found\n")
        return None
    if under_profiler():
        sys.stderr.write("[bench] live counters: this process already runs under a profiler -- not nesting\n")
        return None
    skip = {"--steps", "--warmup", "--spp", "--cpu-budget", "--gpus", "--pmc-keep", "--pmc-save"}
    child, it = [], iter(argv)
    for a in it:
        if a in skip:
            next(it, None)
        elif a.split("=")[0] in skip or a in ("--no-cpu-baseline", "--no-pmc", "--no-drop-in"):
            pass
        else:
            child.append(a)
    # (the children trace the SAME pass size as the timed run -- all `spp` sub-frames in one pass: a launch there is a launch here)
    child += ["--pmc-child", "--steps", "1", "--warmup", "0", "--spp", str(spp), "--no-cpu-baseline", "--no-pmc", "--no-drop-in"]
    root = os.path.abspath(keep_dir) if keep_dir else tempfile.mkdtemp(prefix="skh_pmc_", dir="/tmp")  # (absolute: the children run in /tmp)
    os.makedirs(root, exist_ok=True)
    env = {k: v for k, v in os.environ.items() if not k.startswith(("ROCP", "LD_PRELOAD"))}
    env["TMPDIR"] = "/tmp"
    counters = {k: {} for k in KERNELS}
    units, seconds = {}, []

    def fail(why):
        sys.stderr.write("[bench] live counters failed (%s); logs kept in %s\n" % (why, root))
        return None

    t_all = time.time()
    for k, group in enumerate(PMC_PASSES):
        outdir = os.path.join(root, "pass%d" % k)
        cmd = [rocprof, "--pmc", *group, "--output-format", "csv", "-d", outdir, "--", sys.executable, os.path.abspath(__file__)] + child
        t0 = time.time()
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=600)
        except (OSError, subprocess.TimeoutExpired) as e:
            return fail("pass %d: %s" % (k, type(e).__name__))
        seconds.append(round(time.time() - t0, 1))
        open(os.path.join(root, "pass%d.log" % k), "w").write(r.stdout[-4000:] + r.stderr[-4000:])
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        optional = group == ("GRBM_GUI_ACTIVE",)  # the clock pass is context: without it the VALU roof is priced at the device's maximum clock and says so
        if r.returncode != 0 or not lines:
            if optional:
                sys.stderr.write("[bench] live counters: the clock pass failed (exit code %d): VALU issue priced at the maximum clock\n" % r.returncode)
                continue
            return fail("pass %d: exit code %d" % (k, r.returncode))
        line = json.loads(lines[-1])
        for name, prefix in KERNELS.items():
            got = _pmc_per_launch(outdir, prefix)
            if not all(g in got for g in group):
                if optional:
                    continue
                return fail("pass %d: no %s counters for %s" % (k, "/".join(group), name))
            counters[name].update(got)
            units[name] = line["roofline"]["kernels"][name]["units_per_launch"]
    source = "live: rocprofv3 --pmc child passes of this run (%s): %s s" % (" | ".join(" ".join(g) for g in PMC_PASSES), " + ".join(str(x) for x in seconds))
    fig = {"source": source, "seconds": round(time.time() - t_all, 1), "kernels": {k: pmc_figures(counters[k], units[k]) for k in KERNELS}}
    if not keep_dir:
        shutil.rmtree(root, ignore_errors=True)
    return fig


def pmc_file(scene, resolution):
    """Where the counter figures of a workload are committed: the default workload in profiles/pmc_kernels.json, any other beside it."""
    base = os.path.join(ROOT, "profiles", "pmc_kernels")
    if scene == "kitchen" and resolution == "1920x1080":
        return base + ".json"
    return "%s_%s_%s.json" % (base, re.sub(r"[^A-Za-z0-9]+", "_", os.path.basename(str(scene))), resolution)


def committed_pmc(workload, resolution, scene="kitchen"):
    """Fallback when the counters cannot be collected in this run: the last committed profile of the same workload."""
    path = pmc_file(scene, resolution)
    try:
        j = json.load(open(path))
    except (OSError, ValueError):
        return None
    if j.get("workload") != workload or j.get("resolution") != resolution:
        return None
    j["source"] = "replayed from profiles/%s (tag %s), not measured in this run" % (os.path.basename(path), j.get("tag"))
    j["replayed"] = True
    return j


def launch_command(gpus, argv, port=None):
    """The command `python bench.py --gpus N ...` turns itself into when it was started WITHOUT a launcher: the driver's own
    multi-rank line (one rank per GPU, rendezvous on 127.0.0.1)."""
    port = port or int(os.environ.get("MASTER_PORT", "29541"))
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def maybe_self_launch(args, argv):
    """--gpus N > 1 without WORLD_SIZE in the environment: start the N ranks as a CHILD process (torch.distributed.run) and
    return its exit code.  Runs before torch is imported or the GPU is touched in this process, and never exec()s.  With a
    launcher present, --gpus must agree with WORLD_SIZE: a silent 1-rank run labelled as N GPUs is the failure this guards."""
    ws = os.environ.get("WORLD_SIZE")
    if ws is None:
        if args.gpus <= 1:
            return None
        import subprocess

        env = dict(os.environ, MASTER_ADDR="127.0.0.1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL's peer buffers need it on this driver
        return subprocess.run(launch_command(args.gpus, argv), env=env).returncode
    if int(ws) != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={ws}: launch with --nproc-per-node {args.gpus} or pass --gpus {ws}\n")
        return 2
    return None


def drop_in_leg(ctx, params, W, H, spp, torch, dev):
    """The call pattern of the reference's caller, timed beside the batched headline: HdStrelkaRenderPass::_Execute calls
    render(output) once per sub-frame and maps the image after every call (src/HdStrelka/RenderPass.cpp:441-447;
    OptixBuffer::map = a D2H copy of the float4 image, OptixBuffer.cpp:37-43).  One frame = spp x (skh_render_subframe +
    skh_buffer_download of W*H float4).  Same scene, same rays; the result is the same image (sub-frame batching is exact)."""
    image = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
    host = np.empty((H, W, 4), np.float32)
    ctx.host_register(host)  # as oka::HipBuffer::map does with its host mirror
    p = np.array(params, copy=True)
    ctx.set_option("timing", 0)
    res = {}
    for with_map in (True, False):
        for rep in range(2):  # first repetition = warm-up
            ctx.reset_stats()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(spp):
                p["subframe_index"] = i
                ctx.render_subframe(p, image.data_ptr())  # synchronous, like OptiXRender::render (OptixRender.cpp:1012)
                if with_map:
                    ctx.buffer_download(image.data_ptr(), host)
            dt = time.perf_counter() - t0
        st = ctx.stats()
        rays = st["rays_radiance"] + st["rays_shadow"]
        res["with_map" if with_map else "without_map"] = {"value": round(rays / dt / 1e6, 1), "ms_per_frame": round(dt * 1e3, 2),
                                                          "ms_per_subframe": round(dt * 1e3 / spp, 3)}
    ctx.host_unregister(host)
    return {"unit": "Mray/s", "pattern": f"{spp} x (skh_render_subframe of 1 spp + map() = D2H of the {W}x{H} float4 image), the reference "
            "caller's loop (RenderPass.cpp:441-447); the library traces up to 8 sub-frames ahead once the caller keeps continuing the "
            "frame (option speculate) and traces the next pass while this one is collected (speculate_async), images bit-identical", **res}


def interactive_leg(ctx, sc, W, H, depth, torch, dev, calls=64):
    """The reference viewer's loop with a MOVING camera (src/hdRunner/main.cpp:663-763: one render() of 1 spp per displayed frame; a camera
    change restarts the frame, OptixRender.cpp:903-934): `calls` x (camera orbit step -> skh_render_subframe with subframe_index 0 + map()).
    Every call is a frame of its own -- 2 M paths at 1080p, nothing to trace ahead (the library's speculation only continues a frame) --, so this is
    the small-pass rate: what a 1 / 8 tile share and the first frame after every edit cost.  ms per call, Mray/s, per-kernel ms and launches."""
    import copy
    import math

    from strelka_amd import scene as S

    cam = copy.deepcopy(sc.getCamera())
    eye0 = np.array(cam.position, np.float64)
    fwd = -np.array(cam.rotation[2, :3], np.float64)
    target = eye0 + fwd * 3.0
    image = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
    host = np.empty((H, W, 4), np.float32)
    ctx.host_register(host)

    def params_at(k):
        a = math.radians(0.25 * k)  # a slow orbit about the point the camera looks at
        r = eye0 - target
        e = target + np.array([r[0] * math.cos(a) + r[2] * math.sin(a), r[1], -r[0] * math.sin(a) + r[2] * math.cos(a)])
        cam.lookAt(tuple(e), tuple(target))
        return np.array(S.frame_params(cam, W, H, subframe_index=0, samples_this_launch=1, spp_total=64, max_depth=depth), copy=True)

    res = {}
    for rep in range(2):  # first repetition = warm-up; the second runs with per-kernel timing off (the loop rate) ...
        ctx.set_option("timing", 0)
        ctx.reset_stats()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(calls):
            ctx.render_subframe(params_at(rep * calls + k), image.data_ptr())
            ctx.buffer_download(image.data_ptr(), host)
        dt = time.perf_counter() - t0
    st = ctx.stats()
    rays = st["rays_radiance"] + st["rays_shadow"]
    res.update({"value": round(rays / dt / 1e6, 1), "ms_per_call": round(dt * 1e3 / calls, 3), "rays_per_call": int(rays / calls)})
    ctx.set_option("timing", 1)  # ... and a third with it on: where a call's time goes
    ctx.reset_stats()
    for k in range(16):
        ctx.render_subframe(params_at(2 * calls + k), image.data_ptr())
    st = ctx.stats()
    res["kernel_ms_per_call"] = {k: round(st[k] / 16, 3) for k in ("ms_trace_closest", "ms_trace_shadow", "ms_shade", "ms_raygen", "ms_accumulate")}
    res["launches_per_call"] = round(sum(st[k] for k in ("launches_trace_closest", "launches_trace_shadow", "launches_shade", "launches_other")) / 16, 1)
    ctx.set_option("timing", 0)
    ctx.host_unregister(host)
    return {"unit": "Mray/s", "pattern": f"{calls} x (camera moved -> skh_render_subframe(subframe_index 0, 1 spp) + map()) at {W}x{H}: every call restarts the "
            "frame (OptixRender.cpp:903-934), nothing can be traced ahead", **res}


SCENE_RECIPES = {
    # name: (generator call, workload line)
    "kitchen": (lambda scenes: scenes.kitchen_standin(),
                "kitchen stand-in (SURVEY 8d C3): %d unique triangles, %d instances of %d meshes, "
                "4 rect + 1 distant light, 60/25/10/5 %% diffuse/glossy/metal/glass"),
    # the shape HdStrelka's bake hands over: one mesh per instance (createMesh per instance, RenderPass.cpp:126-129,252-257):
    # the same room, layout, lights, materials and triangle budget, 2000 unique meshes instead of 150 shared ones
    "kitchen_unshared": (lambda scenes: scenes.kitchen_standin(n_meshes=2000, n_instances=2000, tri_lo=200, tri_hi=5000, target_tris=1.72e6),
                         "kitchen stand-in WITHOUT mesh sharing (HdStrelka's per-instance meshes): %d unique triangles, %d instances of %d meshes, "
                         "4 rect + 1 distant light, 60/25/10/5 %% diffuse/glossy/metal/glass"),
    # a less forgiving C3: big flat quads in two triangles, long thin triangles (rods, slats), nested cabinets -> shelves
    # -> crockery, no mesh sharing -- the shape of a USD kitchen after HdStrelka's bake rather than a room full of round blobs
    "kitchen_arch": (lambda scenes: scenes.kitchen_architectural(),
                     "architectural kitchen stand-in: %d unique triangles, %d instances of %d meshes (no sharing), "
                     "4 rect + 1 distant light, 60/25/10/5 %% diffuse/glossy/metal/glass"),
    "hair": (lambda scenes: scenes.hair_standin(), "hair stand-in (SURVEY 8d C5): %d scalp triangles, %d instances of %d meshes + 100 k strands"),
    # the same strands as 8 / 16 / 17 curve prims (one instance each, identity transforms), and 17 prims under small translations: where the
    # world-only curve kernel's table of per-instance trees ends (skh_kernels.h SKH_WORLD_CURVES) -- `also.hair_multi`
    "hair_8": (lambda scenes: scenes.hair_standin(n_prims=8), "hair stand-in as 8 curve prims: %d scalp triangles, %d instances of %d meshes + 100 k strands"),
    "hair_16": (lambda scenes: scenes.hair_standin(n_prims=16), "hair stand-in as 16 curve prims: %d scalp triangles, %d instances of %d meshes + 100 k strands"),
    "hair_17": (lambda scenes: scenes.hair_standin(n_prims=17), "hair stand-in as 17 curve prims: %d scalp triangles, %d instances of %d meshes + 100 k strands"),
    "hair_2_moved": (lambda scenes: scenes.hair_standin(n_prims=2, prim_offset=1e-3), "hair stand-in as 2 curve prims under translations: %d scalp triangles, %d instances of %d meshes + 100 k strands"),
    "hair_4_moved": (lambda scenes: scenes.hair_standin(n_prims=4, prim_offset=1e-3), "hair stand-in as 4 curve prims under translations: %d scalp triangles, %d instances of %d meshes + 100 k strands"),
    "hair_8_moved": (lambda scenes: scenes.hair_standin(n_prims=8, prim_offset=1e-3), "hair stand-in as 8 curve prims under translations: %d scalp triangles, %d instances of %d meshes + 100 k strands"),
    "hair_17_xform": (lambda scenes: scenes.hair_standin(n_prims=17, prim_offset=0.25, shared_xform=True), "hair stand-in as 17 curve prims under ONE rotation + translation: %d scalp triangles, %d instances of %d meshes + 100 k strands"),
    "hair_17_moved": (lambda scenes: scenes.hair_standin(n_prims=17, prim_offset=1e-3), "hair stand-in as 17 curve prims under translations: %d scalp triangles, %d instances of %d meshes + 100 k strands"),
    "cornell": (lambda scenes: scenes.cornell_box(), "cornell box (C2): %d triangles, %d instances of %d meshes"),
}


def scene_cache_path(name):
    """/tmp/skh_bench_<name>_<hash of the generator's source>.skscene: the procedural scenes take tens of seconds of numpy to
    generate; the counter child passes and the ranks of an N-GPU run load the dump one process wrote (strelka_amd/scene_io.py)."""
    import hashlib

    h = hashlib.sha1()
    for f in ("strelka_amd/scenes.py", "strelka_amd/scene.py", "strelka_amd/scene_io.py"):
        h.update(open(os.path.join(ROOT, f), "rb").read())
    return os.path.join("/tmp", "skh_bench_%s_%s_%d.skscene" % (name, h.hexdigest()[:12], os.getuid()))


def load_workload(name, make=True, writer="local rank 0"):
    """(scene, workload line).  Procedural scenes go through the /tmp cache (make=False: wait for this NODE's writer -- local rank 0 --
    to write it; a writer that fails leaves a `.failed` marker so that the waiters stop at once instead of timing out)."""
    from strelka_amd import scene_io, scenes

    if name in SCENE_RECIPES:
        gen, line = SCENE_RECIPES[name]
        path = scene_cache_path(name)
        failed = path + ".failed"
        if not os.path.exists(path):
            if make:
                try:
                    if os.path.exists(failed):
                        os.remove(failed)
                    sc = gen(scenes)
                    tmp = "%s.tmp.%d" % (path, os.getpid())
                    scene_io.save_scene(tmp, sc.arrays(), sc.getCamera())
                    os.replace(tmp, path)
                except BaseException as e:
                    open(failed, "w").write("%s: %s\n" % (type(e).__name__, e))
                    raise
            else:
                t0 = time.time()
                while not os.path.exists(path):
                    if os.path.exists(failed) and os.path.getmtime(failed) >= t0 - 1.0:
                        raise RuntimeError("scene cache %s: the writer (%s) failed: %s" % (path, writer, open(failed).read().strip()))
                    if time.time() - t0 > 300:
                        raise RuntimeError("scene cache %s did not appear within 300 s (writer: %s of this node)" % (path, writer))
                    time.sleep(0.2)
        sc = scene_io.load_scene(path)
    elif name.endswith((".gltf", ".glb")):
        from strelka_amd import gltf

        sc, line = gltf.load_gltf(name), os.path.basename(name) + ": %d triangles, %d instances of %d meshes"
    else:  # a flat dump of a real bake (strelka_amd/scene_io.py)
        sc, line = scene_io.load_scene(name), os.path.basename(name) + ": %d triangles, %d instances of %d meshes"
    arr = sc.arrays()
    return sc, arr, line % (len(arr["indices"]) // 3, len(arr["instances"]), len(arr["meshes"]))


def gather_fallback_allowed(backend, environ):
    """May a multi-rank run whose RCCL communicator (skh_comm_init) could not be formed time torch.distributed's gather instead?
    Only off the real path: the gloo backend (several ranks on one GPU in the 1-GPU tests) or an explicit override.  With one rank
    per GPU on the nccl backend the run must fail loudly -- a record of a different collective is worse than no record."""
    return backend != "nccl" or bool(environ.get("SKH_ALLOW_GATHER_FALLBACK"))


def other_workload_leg(name, W, H, spp, depth, device_ordinal, steps=2):
    """A second workload timed beside the headline (same resolution, spp, depth; its own context): `kitchen_unshared` is the shape
    HdStrelka's bake really hands over -- one mesh per instance (RenderPass.cpp:126-129,252-257)."""
    from strelka_amd import capi, scene as S

    t0 = time.time()
    sc, arr, workload = load_workload(name)
    load_s = time.time() - t0
    ctx = capi.Context(device_ordinal)
    ctx.set_scene(arr)
    ctx.resize(W, H)
    params = S.frame_params(sc.getCamera(), W, H, subframe_index=0, samples_this_launch=1, spp_total=spp, max_depth=depth)
    # counting pass (= the warm-up): exact traversal counters, so that the leg says what its rays cost and not only how fast they went
    ctx.set_option("count_traversal", 1)
    ctx.reset_stats()
    ctx.render_subframes(params, spp, None)
    cst = ctx.stats()
    ctx.set_option("count_traversal", 0)
    ctx.set_option("timing", 1)
    ctx.reset_stats()
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.render_subframes(params, spp, None)
    dt = time.perf_counter() - t0
    st = ctx.stats()
    baked = ctx.baked(len(arr["instances"]))
    binfo = _bvh_info(ctx)
    ctx.close()
    rays = st["rays_radiance"] + st["rays_shadow"]
    nr, ns = max(1, cst["rays_radiance"]), max(1, cst["rays_shadow"])
    return {"workload": workload, "value": round(rays / dt / 1e6, 1), "unit": "Mray/s", "ms_per_step": round(dt / steps * 1e3, 3),
            "kernel_ms_per_frame": {k: round(st[k] / steps, 3) for k in ("ms_trace_closest", "ms_trace_shadow", "ms_shade")},
            "bounces": depth,
            "per_ray": {"nodes": round(cst["nodes_visited"][0] / nr, 2), "tris": round(cst["prims_tested"][0] / nr, 2),
                        "segs": round(cst["segs_tested"][0] / nr, 2), "instances": round(cst["instances_entered"][0] / nr, 2)},
            "per_shadow_ray": {"nodes": round(cst["nodes_visited"][1] / ns, 2), "tris": round(cst["prims_tested"][1] / ns, 2),
                               "segs": round(cst["segs_tested"][1] / ns, 2), "instances": round(cst["instances_entered"][1] / ns, 2)},
            "rays_per_frame": int(rays / steps),
            "bvh_build_ms": round(st["ms_build"], 2), "bvh": binfo, "bake_world": {"baked_instances": baked[1], "baked_triangles": baked[2]},
            "scene_load_s": round(load_s, 1)}


def _bvh_info(ctx):
    """the triangle hierarchy's build record (skh_get_build_info): reinsertion rounds / moves, SAH-style cost before and after"""
    b = ctx.build_info()
    return {"triangles": b["triangles"], "nodes": b["nodes"], "reinsert_rounds": b["reinsert_rounds"], "reinsert_moves": b["reinsert_moves"],
            "reinsert_min_size": b["reinsert_min_size"], "reinsert_ms": round(b["ms_reinsert"], 2),
            "cost_ratio": round(b["cost_after"] / b["cost_before"], 4) if b["cost_before"] > 0 else None}


def frac_of_copy(blk, ceilings):
    """counter HBM bytes / time against what a plain copy reaches on THIS box (ceilings.stream_copy_GBps) instead of the 8 TB/s data-sheet peak"""
    if not ceilings or not blk.get("achieved") or not ceilings.get("stream_copy_GBps"):
        return None
    return round(blk["achieved"] / ceilings["stream_copy_GBps"], 4)


def derive_limiter(blk, ceilings):
    """Names what the kernel is closest to, from the line's own fractions: HBM bytes (counter upper bound, against the 8 TB/s peak), the
    measured random-line rate (L2-miss lines / s against roofline.ceilings), VALU issue (against SIMDs x clock / 2).  The largest one is
    the limiter when it is above 0.6; below that no roof is reached and the kernel is latency / divergence bound -- the record says so
    instead of naming a roof."""
    cands = {"hbm_bytes": blk.get("frac") or 0.0,
             "random_line_rate": (blk.get("l2_miss_lines") or {}).get("frac_of_measured_random_line_rate") or 0.0,
             "valu_issue": (blk.get("valu") or {}).get("frac_valu_issue") or 0.0}
    name, top = max(cands.items(), key=lambda kv: kv[1])
    fr = {k: round(v, 3) for k, v in cands.items()}
    if top <= 0.0:
        return {"name": "unknown: no counters in this run", "fractions": fr}
    if top < 0.6:
        lanes = (blk.get("valu") or {}).get("lanes_per_valu_inst")
        return {"name": "latency / divergence: none of %s above 0.6 (largest: %s %.2f%s)" % (" / ".join(cands), name, top,
                                                                                             ", %.1f of 64 lanes per VALU instruction" % lanes if lanes else ""),
                "fractions": fr}
    return {"name": name, "fractions": fr}


def roofline_fractions(algorithmic_bytes_per_launch, counter_bytes_per_launch, avg_launch_ms, peak_gbs=HBM_PEAK_GBS):
    """The two fractions a reader may want, side by side:
      algorithmic_frac    SURVEY 8(d)'s no-reuse byte model / launch time / HBM peak.  Above 1 (`model_exceeds_peak`) it has stopped
                          being a roofline: L2 + Infinity Cache serve part of those bytes, the kernel is still doing all the work.
      frac (elsewhere)    the TCC counter bytes / time / peak: `frac_kind` = "counter_upper_bound" (FETCH_SIZE counts Infinity-Cache hits).
      l2_hit_share        1 - counter bytes / algorithmic bytes: the share of the model's bytes that never left the L2s (None without counters)."""
    out = {"algorithmic_frac": None, "model_exceeds_peak": None, "l2_hit_share": None}
    if avg_launch_ms and avg_launch_ms > 0 and algorithmic_bytes_per_launch:
        out["algorithmic_frac"] = round(algorithmic_bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9 / peak_gbs, 4)
        out["model_exceeds_peak"] = bool(out["algorithmic_frac"] > 1.0)
        if counter_bytes_per_launch is not None:
            out["l2_hit_share"] = round(1.0 - counter_bytes_per_launch / algorithmic_bytes_per_launch, 4)
    return out


def shade_bytes(rays, next_rays, shadow_rays):
    """k_shade, algorithmic (what the kernel asks for since round 4: the path's radiance stays in place unless a branch changes it, the constant tmin /
    tmax planes are not rewritten): ray 28 r (origin, direction, path id) + hit record 32 r + path state 20 r + 20 w (throughput, lastBsdfPdf, flags)
    per ray; per surface hit (every ray counted as one: upper bound) instance 64 + 64, shading triangle 96, material 64; 28 w per continuation ray,
    32 + 12 w per shadow ray."""
    return rays * (28 + 32 + 40) + rays * (128 + 96 + 64) + next_rays * 28 + shadow_rays * 44


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--depth", type=int, default=4)
    ap.add_argument("--tile", type=int, default=32)
    ap.add_argument("--scene", default="kitchen", help="kitchen | kitchen_unshared | kitchen_arch | cornell | hair | hair_8 | hair_16 | hair_17 | hair_17_xform | hair_17_moved | path to a .skscene dump or a .gltf file")
    ap.add_argument("--waves-per-cu", type=int, default=0)
    ap.add_argument("--opt", action="append", default=[], help="name=value passed to skh_set_option")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=12.0)
    ap.add_argument("--no-pmc", action="store_true", help="do not run the rocprofv3 --pmc child passes (roofline quotes the committed profile)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--pmc-keep", default=None, help="keep the rocprofv3 output of the child passes in this directory")
    ap.add_argument("--pmc-save", default=None, metavar="TAG", help="write the live counter figures to profiles/pmc_kernels.json under this tag")
    ap.add_argument("--no-drop-in", action="store_true", help="skip the one-render()+map()-per-sub-frame leg")
    ap.add_argument("--no-extra", action="store_true", help="skip the other workloads (kitchen_unshared, kitchen_arch) timed beside the headline")
    ap.add_argument("--strict", action="store_true", help="exit 4 when a LIVE counter fraction (HBM bytes or VALU issue) comes out above 1")
    args = ap.parse_args()

    rc = maybe_self_launch(args, sys.argv[1:])
    if rc is not None:
        sys.exit(rc)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # before anything touches the GPU (RCCL peer buffers)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # the scene: generated (or loaded from the /tmp cache) before the GPU is touched; one process of a job writes the cache
    t_scene = time.time()
    sc, arr, workload = load_workload(args.scene, make=(local_rank == 0))  # (per NODE: /tmp is not shared between nodes)
    t_scene = time.time() - t_scene
    global PMC_RESULT
    if world == 1 and not args.no_pmc and not args.pmc_child:
        PMC_RESULT = live_pmc(sys.argv[1:], args.spp, args.pmc_keep)  # child processes; nothing here has touched the GPU yet
    import torch

    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("SKH_DIST_BACKEND", "nccl")  # "gloo": several ranks on ONE GPU (tests on a 1-GPU box)
        if backend == "gloo":
            local_rank = local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", local_rank)

    from strelka_amd import build, capi, scene as S, tiles

    build.build()
    cam = sc.getCamera()
    W, H = args.width, args.height

    ctx = capi.Context(local_rank)
    if args.waves_per_cu:
        ctx.set_option("waves_per_cu", args.waves_per_cu)
    for kv in args.opt:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    ctx.set_scene(arr)
    my_tiles = tiles.assign_tiles(W, H, args.tile, world, rank)
    ctx.set_tiles(args.tile, my_tiles if world > 1 else None)
    ctx.resize(W, H)
    build_ms = ctx.stats()["ms_build"]
    bvh_info = _bvh_info(ctx)
    baked = ctx.baked(len(arr["instances"]))
    params = S.frame_params(cam, W, H, subframe_index=0, samples_this_launch=1, spp_total=args.spp, max_depth=args.depth)

    max_tiles = tiles.max_tiles_per_rank(W, H, args.tile, world)
    tile_buf = torch.zeros((max_tiles, args.tile * args.tile, 4), dtype=torch.float32, device=dev)
    image = torch.zeros((H, W, 4), dtype=torch.float32, device=dev) if rank == 0 else None
    # root: one receive buffer for all ranks' tiles and ONE de-tiling launch per frame; a rank with fewer tiles than
    # max_tiles sends padding, which gets an origin outside the image and is dropped by the scatter kernel
    all_tiles = torch.zeros((world, max_tiles, args.tile * args.tile, 4), dtype=torch.float32, device=dev) if rank == 0 and world > 1 else None
    all_xy = None
    if rank == 0 and world > 1:
        all_xy = np.full((world, max_tiles, 2), max(W, H), np.uint32)
        for r in range(world):
            t = tiles.assign_tiles(W, H, args.tile, world, r)
            all_xy[r, :len(t)] = t
        all_xy = np.ascontiguousarray(all_xy.reshape(-1, 2))

    # The frame's one collective runs BELOW the C ABI (skh_gather_tiles: grouped RCCL sends into the root on the renderer's own
    # stream); torch.distributed only carries the 128-byte communicator id, the barrier and the timing reduction.  If RCCL cannot
    # build the communicator (several ranks sharing one GPU in the 1-GPU tests) every rank agrees to fall back to
    # torch.distributed's gather and the record says so; with the nccl backend (one rank per GPU: the real thing) that
    # fall-back is an ERROR -- the run ends with a non-zero exit code instead of timing a different collective.
    gather_kind, rccl_nranks, gather_error = ("none" if world == 1 else "torch.distributed gather"), 0, None
    want_rccl = world > 1 and os.environ.get("SKH_GATHER", "rccl" if dist.get_backend() == "nccl" else "torch") == "rccl"
    if want_rccl:
        idt = torch.zeros(128, dtype=torch.uint8)
        ok = 1
        if rank == 0:
            try:
                idt = torch.from_numpy(capi.Context.comm_unique_id().copy())
            except capi.SkhError as e:
                ok, gather_error = 0, str(e)
        cdev = dev if dist.get_backend() == "nccl" else torch.device("cpu")
        idt = idt.to(cdev)
        dist.broadcast(idt, src=0)
        try:
            if not int(idt.any()):
                raise capi.SkhError("no communicator id")
            ctx.comm_init(idt.cpu().numpy(), world, rank)
        except capi.SkhError as e:
            ok, gather_error = 0, str(e)
            sys.stderr.write(f"[bench] rank {rank}: {e}\n")
        okt = torch.tensor([ok], dtype=torch.int32, device=cdev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        if int(okt.item()) == 1:
            gather_kind = "skh_gather_tiles: RCCL send/recv below the C ABI"
            if os.environ.get("SKH_RCCL_LIB"):  # (tests: tests/cpp/rccl_double.cpp stands in for librccl so that N ranks can share one GPU)
                gather_kind += " (SKH_RCCL_LIB test double, not RCCL: %s)" % os.path.basename(os.environ["SKH_RCCL_LIB"])
            rccl_nranks = ctx.comm_info()[2]  # what RCCL itself reports (ncclCommCount)
        else:
            ctx.comm_destroy()
            if not gather_fallback_allowed(dist.get_backend(), os.environ):
                if rank == 0:
                    sys.stderr.write("[bench] --gpus %d on the nccl backend, but the RCCL communicator below the C ABI could not be formed (%s): "
                                     "refusing to time torch.distributed's gather instead (SKH_ALLOW_GATHER_FALLBACK=1 overrides)\n" % (world, gather_error))
                dist.barrier()
                dist.destroy_process_group()
                ctx.close()
                sys.exit(3)
    use_skh_gather = gather_kind.startswith("skh_gather_tiles")

    def frame():
        # all sub-frames of the frame, one device sync at the end; like the reference's render(output) the single-GPU frame writes the
        # output image (a rank of an N-GPU frame hands its tiles to the gather, the root's scatter writes the image)
        ctx.render_subframes(params, args.spp, image.data_ptr() if world == 1 else None)
        if world > 1:
            if use_skh_gather:
                ctx.gather_tiles(max_tiles, all_tiles.data_ptr() if rank == 0 else None, 0)
            else:
                ctx.copy_accum_tiles(tile_buf.data_ptr())
                tiles.gather_tiles(tile_buf, world, rank, dist, out=all_tiles)
            if rank == 0:
                ctx.scatter_tiles(all_tiles.data_ptr(), all_xy, args.tile, image.data_ptr(), W, H)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- untimed counting pass: exact traversal counters for the algorithmic-bytes figures (identical every frame, the
    #      renderer is deterministic); rank 0 only -- they describe rank 0's kernels ----
    cst = None
    if rank == 0:
        ctx.set_option("count_traversal", 1)
        ctx.reset_stats()
        ctx.render_subframes(params, args.spp, None)
        cst = ctx.stats()
        ctx.set_option("count_traversal", 0)
    for _ in range(args.warmup):
        frame()
    ctx.set_option("timing", 1)
    ctx.reset_stats()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        frame()
    torch.cuda.synchronize()
    dt_own = time.perf_counter() - t0  # this rank alone: render + its part of the gather
    barrier()
    dt = time.perf_counter() - t0
    st = ctx.stats()
    rays_local = st["rays_radiance"] + st["rays_shadow"]
    per_rank = None
    if world > 1:
        rdev = dev if dist.get_backend() == "nccl" else torch.device("cpu")
        tt = torch.tensor([dt], dtype=torch.float64, device=rdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        mine = torch.tensor([dt_own, float(rays_local), float(len(my_tiles))], dtype=torch.float64, device=rdev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        allr = np.array([t.cpu().numpy() for t in allr])
        rays_total = float(allr[:, 1].sum())
        K_ = max(1, args.steps)
        ms = allr[:, 0] / K_ * 1e3
        per_rank = {"ms_per_step": {"min": round(float(ms.min()), 3), "mean": round(float(ms.mean()), 3), "max": round(float(ms.max()), 3),
                                    "all": [round(float(x), 3) for x in ms]},
                    "rays_imbalance_max_over_mean": round(float(allr[:, 1].max() / max(1.0, allr[:, 1].mean())), 4),
                    "tiles": [int(x) for x in allr[:, 2]]}
    else:
        rays_total = float(rays_local)

    strict_fail = False
    drop_in = None
    interactive = None
    if rank == 0 and world == 1 and not args.no_drop_in and not args.pmc_child:
        drop_in = drop_in_leg(ctx, params, W, H, args.spp, torch, dev)
        interactive = interactive_leg(ctx, sc, W, H, args.depth, torch, dev)
    extra = None
    if rank == 0 and world == 1 and args.scene == "kitchen" and not args.no_extra and not args.no_drop_in and not args.pmc_child:
        extra = {name: other_workload_leg(name, W, H, args.spp, args.depth, local_rank) for name in ("kitchen_unshared", "kitchen_arch")}
        # C5 (BASELINE.json configs[4]): the basis-curves workload at ITS depth (3), 64 spp per pass like the headline
        extra["hair"] = other_workload_leg("hair", W, H, args.spp, 3, local_rank)
        # the same groom split into several curve prims: Mray/s and what each ray costs
        extra["hair_multi"] = {name: {k: v for k, v in other_workload_leg(name, W, H, args.spp, 3, local_rank).items()
                                      if k in ("workload", "value", "unit", "ms_per_step", "kernel_ms_per_frame", "per_ray", "per_shadow_ray", "bvh_build_ms")}
                               for name in ("hair_8", "hair_16", "hair_17", "hair_17_xform", "hair_17_moved")}
    if rank == 0:
        K = max(1, args.steps)
        # ---- rooflines of the three hot kernels.  Per kernel, from this run's counters and hipEvent times:
        #   frac (= frac_hbm)   HBM-side bytes per launch from the TCC counters / live launch time / 8 TB/s.  FETCH_SIZE counts
        #                       Infinity-Cache hits too (MI355X_MICROARCH.md "HBM"), so this is an UPPER bound on HBM use.
        #   frac_valu_issue     VALU wave-instructions per second / (SIMDs x clock / 2): a wave64 VALU op issues over 2 cycles.
        #   cached_bw           SURVEY 8(d)'s algorithmic bytes / time: what L2 + Infinity Cache + HBM deliver together; it is
        #                       NOT divided by the HBM peak (most of those bytes never reach HBM).
        pmc = PMC_RESULT if PMC_RESULT else committed_pmc(workload, f"{W}x{H}", args.scene)
        info = ctx.device_info()
        clock_ghz = info["clock_khz"] / 1e6  # device maximum engine clock as this box's hipDeviceProp_t reports it
        simds = info["compute_units"] * info["simds_per_cu"]
        peak_issue = simds * clock_ghz / 2.0  # G wave-instructions / s
        flags = []
        # rays entering k_shade = radiance rays; what it emits = the radiance rays of the later bounces + the shadow rays
        first = W * H * args.spp if world == 1 else None
        next_rays = max(0, cst["rays_radiance"] - (first or 0)) if first else 0
        alg = {"closest": algorithmic_bytes(cst["rays_radiance"], False, cst["nodes_visited"][0], cst["prims_tested"][0], cst["segs_tested"][0], cst["instances_entered"][0]),
               "shadow": algorithmic_bytes(cst["rays_shadow"], True, cst["nodes_visited"][1], cst["prims_tested"][1], cst["segs_tested"][1], cst["instances_entered"][1]),
               "shade": shade_bytes(cst["rays_radiance"], next_rays, cst["rays_shadow"])}
        units = {"closest": cst["rays_radiance"], "shadow": cst["rays_shadow"], "shade": cst["rays_radiance"]}
        stat_key = {"closest": "trace_closest", "shadow": "trace_shadow", "shade": "shade"}
        kern = {}
        for name in ("closest", "shadow", "shade"):
            launches = max(1, st["launches_" + stat_key[name]])
            avg_ms = st["ms_" + stat_key[name]] / launches
            upl = units[name] * K / launches  # rays (paths for k_shade) per launch
            blk = {"kernel": {"closest": "k_trace<closest>", "shadow": "k_trace<shadow>", "shade": "k_shade"}[name], "bound": "hbm",
                   "avg_launch_ms": round(avg_ms, 4), "launches_per_frame": launches // K, "units_per_launch": int(upl),
                   "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "frac_kind": "counter_upper_bound", "traffic": None}
            pk = (pmc or {}).get("kernels", {}).get(name)
            if pk and pk.get("hbm_bytes_per_launch") and avg_ms > 0:
                scale = upl / pk["rays_per_launch"] if pk.get("rays_per_launch") else 1.0
                blk["traffic"] = int(pk["hbm_bytes_per_launch"] * scale)
                blk["achieved"] = round(blk["traffic"] / (avg_ms * 1e-3) / 1e9, 2)
                blk["frac"] = round(blk["achieved"] / HBM_PEAK_GBS, 5)
                blk["counter_scale"] = round(scale, 4)  # timed launch size / the counter pass's launch size (1.0: same pass size)
                blk["write_bytes_per_unit"] = round(pk["WRITE_SIZE_KiB"] * 1024.0 / max(1, pk["rays_per_launch"]), 1)
                blk["fetch_bytes_per_unit"] = round(2.0 * pk["FETCH_SIZE_KiB"] * 1024.0 / max(1, pk["rays_per_launch"]), 1)
                if blk["frac"] > 1.0:
                    flags.append("%s: counter bytes / time exceed the HBM peak (FETCH_SIZE counts Infinity-Cache hits)" % name)
            if pk and pk.get("valu_per_launch") and avg_ms > 0:
                scale = upl / pk["rays_per_launch"] if pk.get("rays_per_launch") else 1.0
                rate = pk["valu_per_launch"] * scale / (avg_ms * 1e-3) / 1e9
                lanes = pk.get("lanes_per_valu_inst")
                # The issue roof against the clock the launches REALLY ran at (GRBM_GUI_ACTIVE / duration in this run's counter pass: the boxes run
                # these kernels at 2.25-2.35 GHz, not at the 2.4 GHz hipDeviceProp_t reports) -- `frac_valu_issue` and the limiter use it when it was
                # measured; `frac_valu_issue_at_max_clock` keeps the data-sheet figure beside it.
                mclk = pk.get("clock_ghz_measured")
                peak_meas = simds * mclk / 2.0 if mclk else None
                blk["valu"] = {"valu_wave_insts_per_launch": int(pk["valu_per_launch"] * scale), "rate_G_per_s": round(rate, 1),
                               "peak_G_per_s": round(peak_meas or peak_issue, 1), "clock_ghz": round(mclk or clock_ghz, 3),
                               "clock_is": "measured: GRBM_GUI_ACTIVE / launch duration" if mclk else "device maximum (hipDeviceProp_t): not measured in this run",
                               "frac_valu_issue": round(rate / (peak_meas or peak_issue), 4), "frac_valu_issue_at_max_clock": round(rate / peak_issue, 4),
                               "lanes_per_valu_inst": lanes,
                               "frac_lane_throughput": round(rate / (peak_meas or peak_issue) * lanes / 64.0, 4) if lanes else None,
                               "valu_per_unit": round(pk["valu_per_launch"] / max(1, pk.get("rays_per_launch") or upl), 1),
                               "salu_per_valu": pk.get("salu_per_valu"), "wait_inst_any_frac": pk.get("wait_inst_any_frac")}
                if rate / peak_issue > 1.0:
                    flags.append("%s: VALU issue rate above the roof: the clock or the counters are off" % name)
            bpl = alg[name] * K / launches
            blk["cached_bw"] = {"GB/s": round(bpl / (avg_ms * 1e-3) / 1e9, 2) if avg_ms > 0 else 0.0, "algorithmic_bytes_per_launch": int(bpl),
                                "algorithmic_bytes_per_unit": round(alg[name] / max(1, units[name]), 1)}
            blk.update(roofline_fractions(bpl, blk["traffic"], avg_ms))
            kern[name] = blk
        # ---- measured memory ceilings of this box (skh_probe_memory; SURVEY 8(d) "report a measured STREAM-copy ceiling"): a uint4
        #      copy, and random aligned 64- / 128-byte record fetches from one-wave workgroups on the trace kernels' grid over a buffer
        #      the size of the kitchen hierarchy.  The fabric serves random fetches by the 128-byte LINE (32-, 64- and 128-byte records
        #      all arrive at the same records/s; TCC_EA0_RDREQ = one per record: profiles/r03d_probe_calibration.txt), so a kernel's
        #      L2-miss rate is compared in lines/s: FETCH_SIZE KiB x 1024 / 64 = requests = lines.
        ceilings = None
        try:
            if args.pmc_child:
                raise RuntimeError("counter child pass: not measured under the profiler")
            psize = 1792 << 20
            copy = ctx.probe_memory(0, psize)[0]
            g64, g128, c64 = ctx.probe_memory(1, psize, 64)[0], ctx.probe_memory(1, psize, 128)[0], ctx.probe_memory(2, psize, 64)[0]
            line_rate = max(g64, c64) / 64.0  # G lines / s
            ceilings = {"buffer_MiB": psize >> 20, "stream_copy_GBps": round(copy, 1), "gather64_GBps": round(g64, 1), "gather128_GBps": round(g128, 1),
                        "chase64_GBps": round(c64, 1), "random_line_rate_G_per_s": round(line_rate, 2),
                        "note": "random record fetches cost one 128-byte line each whatever their size; copy counts read + write"}
            for name, blk in kern.items():
                pk = (pmc or {}).get("kernels", {}).get(name)
                if pk and pk.get("FETCH_SIZE_KiB") and blk["avg_launch_ms"] > 0:
                    scale = blk["units_per_launch"] / pk["rays_per_launch"] if pk.get("rays_per_launch") else 1.0
                    lines = pk["FETCH_SIZE_KiB"] * 1024.0 / 64.0 * scale
                    rate = lines / (blk["avg_launch_ms"] * 1e-3) / 1e9
                    blk["l2_miss_lines"] = {"per_unit": round(lines / max(1, blk["units_per_launch"]), 2), "G_per_s": round(rate, 2),
                                            "frac_of_measured_random_line_rate": round(rate / line_rate, 4)}
        except Exception as e:  # a ceiling is context, not the measurement: say so and go on
            sys.stderr.write("[bench] memory ceilings not measured: %s\n" % e)
        for blk in kern.values():
            blk["limiter"] = derive_limiter(blk, ceilings)
            blk["frac_of_measured_copy"] = frac_of_copy(blk, ceilings)
        nrs = max(1, cst["rays_shadow"])
        c0 = kern["closest"]
        roofline = {"kernel": "k_trace<closest>", "bound": "hbm", "achieved": c0["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": c0["frac"],
                    "frac_kind": "counter_upper_bound", "frac_is": "upper bound on HBM use: FETCH_SIZE counts Infinity-Cache hits",
                    "algorithmic_frac": c0["algorithmic_frac"], "model_exceeds_peak": c0["model_exceeds_peak"], "l2_hit_share": c0["l2_hit_share"],
                    "algorithmic_frac_is": "SURVEY 8(d) no-reuse bytes / time / peak; above 1 = the caches serve part of the model's bytes",
                    "traffic": c0["traffic"], "traffic_source": pmc.get("source") if pmc else None,
                    "limiter": derive_limiter(c0, ceilings), "frac_of_measured_copy": frac_of_copy(c0, ceilings), "valu": c0.get("valu"),
                    "cached_bw": dict(c0["cached_bw"], note="SURVEY 8(d) bytes / time; served by L2 + Infinity Cache + HBM together, not an HBM fraction"),
                    "avg_launch_ms": c0["avg_launch_ms"], "rays_per_launch": c0["units_per_launch"],
                    "clock_ghz": round(clock_ghz, 3), "clock_ghz_measured": (c0.get("valu") or {}).get("clock_ghz") if (c0.get("valu") or {}).get("clock_is", "").startswith("measured") else None,
                    "simds": simds,
                    "per_ray": {"nodes": round(cst["nodes_visited"][0] / max(1, cst["rays_radiance"]), 2),
                                "tris": round(cst["prims_tested"][0] / max(1, cst["rays_radiance"]), 2),
                                "instances": round(cst["instances_entered"][0] / max(1, cst["rays_radiance"]), 2)},
                    "per_shadow_ray": {"nodes": round(cst["nodes_visited"][1] / nrs, 2), "tris": round(cst["prims_tested"][1] / nrs, 2),
                                       "instances": round(cst["instances_entered"][1] / nrs, 2)},
                    "kernels": kern, "ceilings": ceilings}
        if flags:
            roofline["flags"] = flags
            sys.stderr.write("[bench] roofline flags: %s\n" % "; ".join(flags))
        out = {
            "metric": "Mray/s", "value": round(rays_total / dt / 1e6, 3), "unit": "Mray/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / K * 1e3, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload, "resolution": f"{W}x{H}", "bounces": args.depth, "spp": args.spp,
                       "step": f"one frame = one skh_render_subframes({args.spp}) pass: {args.spp} sub-frames of 1 spp traced together, accumulated in order (bit-identical to {args.spp} calls; the "
                               "reference caller's one-call-per-sub-frame loop is `value_drop_in` / `drop_in`, its moving-camera loop `interactive`)", "tile": args.tile, "world_size": world,
                       "parallelism": f"pixel tiles round-robin over {world} GPU(s), 1 gather/frame ({gather_kind})" if world > 1
                       else "single GPU", "rays_per_frame": int(rays_total / K), "bvh_build_ms": round(build_ms, 2), "bvh": bvh_info,
                       "bake_world": {"baked_instances": baked[1], "baked_triangles": baked[2]}, "scene_load_s": round(t_scene, 1)},
            "gather": gather_kind, "rccl_nranks": rccl_nranks,
            "roofline_replayed": bool(pmc and pmc.get("replayed")) or pmc is None,
            "kernel_ms_per_frame": {k: round(st[k] / K, 3) for k in ("ms_trace_closest", "ms_trace_shadow", "ms_shade",
                                                                     "ms_raygen", "ms_accumulate", "ms_sort")},
            "roofline": roofline,
        }
        if per_rank is not None:
            out["per_rank"] = per_rank
        if gather_error:
            out["gather_error"] = gather_error
        if drop_in is not None:
            out["drop_in"] = drop_in
            # the number a Strelka user gets through oka::HipRender (one render() + map() per sub-frame, still camera), on the record's first screen beside `value`
            out["value_drop_in"] = drop_in["with_map"]["value"]
        if interactive is not None:
            out["interactive"] = interactive
        if extra is not None:
            out["also"] = extra
        if args.pmc_save and PMC_RESULT:
            json.dump({**PMC_RESULT, "tag": args.pmc_save, "workload": workload, "resolution": f"{W}x{H}"},
                      open(pmc_file(args.scene, f"{W}x{H}"), "w"), indent=1)
        if os.environ.get("SKH_BENCH_CHECKSUM"):
            # CRC of the final accumulation image (tests: a tile-sharded N-rank run must reproduce the 1-rank image exactly)
            import zlib

            img = image.cpu().numpy() if world > 1 else ctx.read_accum()
            out["image_crc32"] = zlib.crc32(np.ascontiguousarray(img[..., :3]).tobytes())
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(arr, cam, W, H, args.spp, args.depth, args.cpu_budget)
        print(json.dumps(out), flush=True)
        if args.strict and flags and PMC_RESULT:
            sys.stderr.write("[bench] --strict: a live counter fraction exceeds 1\n")
            strict_fail = True
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()
    if strict_fail:
        sys.exit(4)


if __name__ == "__main__":
    main()
