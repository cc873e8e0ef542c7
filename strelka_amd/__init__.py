"""strelka_amd -- MI355X-native wavefront path tracer behind Strelka's oka::Render interface.

Package layout (only what the hot path needs):
    csrc/        hand-written HIP kernels (gfx950) + the C-ABI implementation  -> libstrelka_hip.so
    build.py     in-tree hipcc build
    capi.py      ctypes binding of include/strelka_hip.h (no CPU fallback)
    scene.py     host-side mirror of oka::Scene / oka::Camera (flat arrays in the C-ABI layouts)
    scenes.py    seeded procedural stand-ins for the BASELINE scenes
    host/        C++ mirror of oka::Render / Buffer / SettingsManager / SharedContext / Scene: oka::HipRender above the C ABI
    scene_io.py  .skscene flat dump of an oka::Scene bake;  gltf.py  glTF loader with the reference loader's semantics
    tiles.py     multi-GPU pixel-tile assignment (the gather itself is skh_gather_tiles, below the C ABI)
"""
__all__ = ["scene", "scenes", "capi", "build"]
