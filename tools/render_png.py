"""Renders one of the stand-in scenes on the GPU and writes a tonemapped PNG (usage: render_png.py kitchen|cornell|hair W H spp depth [exposure_scale] [out.png])."""
import sys, numpy as np
sys.path.insert(0, ".")
import torch
from strelka_amd import capi, scene as S, scenes, png
name = sys.argv[1]; W, H, spp, depth = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
sc = {"kitchen": scenes.kitchen_standin, "cornell": scenes.cornell_box, "hair": scenes.hair_standin}[name]()
ctx = capi.Context(0); ctx.set_scene(sc.arrays()); ctx.resize(W, H)
p = S.frame_params(sc.getCamera(), W, H, subframe_index=0, spp_total=spp, max_depth=depth)
ctx.render_subframes(p, spp, None)
img = torch.from_numpy(ctx.read_accum()).cuda()
e = S.default_exposure() * np.float32(float(sys.argv[6]) if len(sys.argv) > 6 else 1.0)
ctx.tonemap(img.data_ptr(), W, H, 1, e, 2.2)   # Reinhard + gamma, as the reference's post chain
png.save_png((sys.argv[7] if len(sys.argv) > 7 else f"gpurun_out/{name}.png"), img.cpu().numpy()[..., :3], flipped=True)
print("wrote", name, float(img.mean()))
