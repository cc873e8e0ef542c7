"""The closest-hit experiments of round 4 stay in the source behind compile-time flags (docs/LOG.md "five ways to cut its dependent chain":
pop-time culling, postponed leaves, the touch prefetch).  Each changes which boxes a ray looks at and when -- never a primitive test -- so a
build with any of them must reproduce the oracle's hit records bit for bit, through the world-only kernel they live in.  Built with
strelka_amd.build.build_variant (hipcc on the GPU box), run in a child process with SKH_LIB pointing at the variant."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

CODE = r'''
import os, sys
sys.path.insert(0, os.environ["SKH_ROOT"])
import numpy as np
from strelka_amd import capi, scene as S, scenes
from tests import orklib
from tests.test_gpu_parity import small_kitchen, camera_rays, assert_hits_equal
sc = small_kitchen()
arr = sc.arrays()
ctx = capi.Context(0)
o = orklib.new_context(); o.set_scene(arr); ctx.set_scene(arr)
assert ctx.baked(len(arr["instances"]))[1] > 0  # (bake_world 4: the mesh instances are baked and the world-only kernel -- where the experiments live -- runs)
rays = np.concatenate([camera_rays(sc, 96, 96, 40000, 7), scenes.random_rays(40000, 8, -3.5, 3.5)])
assert_hits_equal(ctx.trace(rays, 0), o.trace(rays, 0))
# and a small frame through the wavefront loop (overlapped pass: the tail passes run too), against the same build's own one-stream render
ctx.resize(128, 96)
p = S.frame_params(sc.getCamera(), 128, 96, subframe_index=0, spp_total=4, max_depth=5)
ctx.render_subframes(p, 4)
a = ctx.read_accum()
ctx.set_option("overlap", 0); ctx.resize(128, 96); ctx.render_subframes(p, 4)
assert np.array_equal(a, ctx.read_accum())
o.resize(128, 96)
for i in range(4):
    p["subframe_index"] = i
    o.render_subframe(p)
w = o.read_accum()[..., :3].astype(np.float64)
l2 = np.sqrt(((a[..., :3] - w) ** 2).sum()) / np.sqrt((w ** 2).sum())
assert l2 < 2e-5, l2
print("VARIANT-OK")
'''


# (three builds of ~1 minute each on the GPU box: the culling build as measured, culling with a 3-entry LDS stack so that the global overflow
# area mixes in, and everything at once -- without culling, which excludes the tail passes -- so that they are exercised together)
@pytest.mark.parametrize("defines", [["SKH_POP_CULL=1", "SKH_CULL_LDS=10"], ["SKH_POP_CULL=1", "SKH_CULL_LDS=3", "SKH_PREFETCH2=1"],
                                     ["SKH_POSTPONE=1", "SKH_PREFETCH2=1"]], ids=["cull10", "cull3+prefetch", "postpone+prefetch"])
def test_experiment_builds_give_the_oracles_hit_records(tmp_path, defines):
    from strelka_amd import build

    lib = build.build_variant(str(tmp_path / "libstrelka_hip_variant.so"), defines)
    env = dict(os.environ, SKH_LIB=lib, SKH_ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    out = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "VARIANT-OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
