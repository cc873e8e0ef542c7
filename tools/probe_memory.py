"""Measured memory ceilings of the box (skh_probe_memory): copy, random 32/64/128-byte record fetches, over several buffer sizes.
usage (GPU box): python tools/probe_memory.py"""
import sys; sys.path.insert(0, ".")
from strelka_amd import capi
c = capi.Context(0)
for size in (64 << 20, 512 << 20, 1800 << 20, 4096 << 20):
    print("buffer %5d MiB: copy %.0f GB/s |" % (size >> 20, c.probe_memory(0, size)[0]), " | ".join("%s %3d B: %.0f GB/s" % (n, rb, c.probe_memory(k, size, rb)[0]) for rb in (32, 64, 128) for k, n in ((1, "gather"), (2, "chase"))), flush=True)
