"""skh_probe_memory: the measured memory ceilings bench.py prints beside the roofline fractions (include/strelka_hip.h)."""
import pytest

pytestmark = pytest.mark.gpu


def test_memory_ceilings_are_sane_and_random_fetches_cost_a_line_each():
    from strelka_amd import capi

    ctx = capi.Context(0)
    size = 1024 << 20
    copy, ms = ctx.probe_memory(0, size)
    assert 1000.0 < copy < 8000.0 and ms > 0.0  # GB/s, read + write; 8 TB/s is the HBM peak
    g32, g64, g128 = (ctx.probe_memory(1, size, rb)[0] for rb in (32, 64, 128))
    c64 = ctx.probe_memory(2, size, 64)[0]
    # the finding the line layout and the bench's lines/s figures rest on: records/s does not depend on the record size up to a line
    assert 0.7 < (g64 / 64.0) / (g32 / 32.0) < 1.4 and 0.7 < (g128 / 128.0) / (g64 / 64.0) < 1.4
    assert 0.5 < c64 / g64 < 2.0
    assert g128 < 8000.0 * 1.05
    with pytest.raises(capi.SkhError):
        ctx.probe_memory(1, size, 48)  # record sizes: 32, 64, 128
    with pytest.raises(capi.SkhError):
        ctx.probe_memory(3, size, 64)
    ctx.close()
