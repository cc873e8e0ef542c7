"""Reproduce tools/fuzz_hits.py seeds and print the mismatching rays: GPU (several hierarchies) vs the oracle's BVH vs the oracle's brute-force
loop, which is the definition of the result.  usage (GPU box): python tools/fuzz_debug.py <seed> [<seed> ...]"""
import sys, numpy as np
sys.path.insert(0, ".")
src = open("tools/fuzz_hits.py").read()
for seed in [int(a) for a in sys.argv[1:]]:
    pre, post = src.split("    bk = (seed // 3) % 4")
    pre = pre.replace("for seed in range(int(sys.argv[1]), int(sys.argv[2])):", "for seed in [%d]:" % seed)
    ns = {}
    exec(compile(pre + "    break\n", "fuzz", "exec"), ns)
    arr, rays, S, capi, orklib = ns["arr"], ns["rays"], ns["S"], ns["capi"], ns["orklib"]
    bk = (seed // 3) % 4
    o = orklib.new_context(); o.set_bake(bk); o.set_scene(arr)
    want = o.trace(rays, 0)
    res = {}
    for name, opts in (("default", {"curve_split": 1 + seed % 4, "leaf_max_tris": 1 + seed % 4}), ("karras", {"build_quality": 0}), ("leaf1", {"leaf_max_tris": 1}), ("leaf4", {"leaf_max_tris": 4, "tlas_build": 1})):
        ctx = capi.Context(0); ctx.set_option("bake_world", bk)
        for k, v in opts.items(): ctx.set_option(k, v)
        ctx.set_scene(arr); res[name] = ctx.trace(rays, 0); ctx.close()
    eq = lambda a, b: (a.view(np.uint8).reshape(len(a), -1) == b.view(np.uint8).reshape(len(b), -1)).all(1)
    bad = np.nonzero(~eq(res["default"], want) | ~eq(res["karras"], want) | ~eq(res["leaf1"], want) | ~eq(res["leaf4"], want))[0]
    br = o.trace(rays[bad], 0, brute=True)
    inst = arr["instances"]
    for n, i in enumerate(bad[:6]):
        print("seed", seed, "ray", i, "o", rays["origin"][i], "d", rays["dir"][i])
        print("   brute ", br[n]); print("   orcBVH", want[i])
        for name in res: print("   gpu", name, res[name][i])
        k = br[n]["instance_id"]
        if k != 0xFFFFFFFF:
            M = inst["transform"][k].reshape(3, 4); size = np.abs(M[:, :3]).sum(1).max()
            print("   instance", k, "type", inst["type"][k], "size ~%.3g" % size, "distance / size = %.3g" % (br[n]["t"] / size))
