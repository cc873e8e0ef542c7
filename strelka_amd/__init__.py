"""strelka_amd -- MI355X-native wavefront path tracer behind Strelka's oka::Render interface.

Package layout (only what the hot path needs):
    csrc/        hand-written HIP kernels (gfx950) + the C-ABI implementation  -> libstrelka_hip.so
    build.py     in-tree hipcc build
    capi.py      ctypes binding of include/strelka_hip.h (no CPU fallback)
    scene.py     host-side mirror of oka::Scene / oka::Camera (flat arrays in the C-ABI layouts)
    scenes.py    seeded procedural stand-ins for the BASELINE scenes
    render.py    host-side mirror of oka::Render / Buffer / SettingsManager / SharedContext
    tiles.py     multi-GPU pixel-tile sharding + RCCL gather
"""
__all__ = ["scene", "scenes", "capi", "build"]
