"""The probe launches tools/probe_calibrate.sh runs under rocprofv3 --pmc (profiles/r03d_probe_calibration.txt)."""
import sys; sys.path.insert(0, ".")
from strelka_amd import capi
c = capi.Context(0)
size = 1800 << 20
print("copy", c.probe_memory(0, size, 64, 1))
for rb in (32, 64, 128):
    print("gather", rb, c.probe_memory(1, size, rb, 1), "chase", rb, c.probe_memory(2, size, rb, 1), flush=True)
