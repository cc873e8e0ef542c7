"""strelka_amd/csrc/skh_libm.h: the sin / cos / acos / asin / atan2 / exp / log / sinh / pow both sides compile (fixed polynomials in correctly
rounded operations).  Here, on the CPU: how far they are from the correctly rounded values (float64 numpy as the reference), at the bars
the header states -- the same <= 4 ulp class at which the reference-generated light fixtures are held.  That the DEVICE returns the same bits
is tests/test_gpu_golden.py::test_libm_is_bit_identical_on_the_device; that the device's OWN outputs meet the same float64 bars -- without the CPU
compile of the shared text in between -- is test_libm_on_the_device_against_float64 there."""
import numpy as np
import pytest

from tests import orklib


def libm_inputs(n=400000, seed=7):
    rs = np.random.RandomState(seed)
    x = np.concatenate([rs.uniform(-6.3, 6.3, n // 4), rs.uniform(-400, 400, n // 4), rs.uniform(-1, 1, n // 4),
                        np.exp(rs.uniform(np.log(1e-6), np.log(1e4), n // 8)), rs.uniform(-87, 88, n // 8)]).astype(np.float32)
    y = np.concatenate([rs.uniform(-50, 50, n // 2), np.full(n // 4, 1.0 / 2.2), np.full(n // 4, 2.2)]).astype(np.float32)
    special = np.array([0.0, -0.0, 1.0, -1.0, 0.5, -0.5, np.inf, -np.inf, np.nan, 1e-45, 1e38, -1e38, 88.8, -104.0, 2.0, 1e-30], np.float32)
    x = np.concatenate([x, special, special[::-1]])
    y = np.concatenate([y[: len(x) - 2 * len(special)], special, special])
    return np.ascontiguousarray(np.stack([x, y], 1))


def cpu_libm(rec):
    lib = orklib.load()
    out = np.zeros((len(rec), 10), np.float32)
    lib.ork_libm(rec.ctypes.data, len(rec), out.ctypes.data)
    return out


def ulp_err(got, ref64):
    ref32 = ref64.astype(np.float32)
    with np.errstate(invalid="ignore", over="ignore"):
        sp = np.spacing(np.abs(ref32)).astype(np.float64)
        return np.abs(got.astype(np.float64) - ref64) / sp


ACCURACY_BARS = [
    # (output column, float64 reference, domain, bar in ulp of the correctly rounded float32 value)
    (0, np.sin, lambda x, y: np.abs(x) <= 400, 1.7), (1, np.cos, lambda x, y: np.abs(x) <= 400, 1.7),
    (2, np.arccos, lambda x, y: np.abs(x) <= 1, 1.3), (3, np.arcsin, lambda x, y: np.abs(x) <= 1, 2.5),
    (5, np.exp, lambda x, y: (x > -87) & (x < 88), 1.2), (6, np.log, lambda x, y: (x > 1e-37) & (x < 1e38), 1.0),
    (7, np.sinh, lambda x, y: np.abs(x) <= 20, 1.8)]


def check_accuracy(rec, out, col, fn, dom, bar):
    """`out` = the ten columns of SKH_UNIT_LIBM / ork_libm for the records `rec`, from WHICHEVER side computed them"""
    x, y = rec[:, 0].astype(np.float64), rec[:, 1].astype(np.float64)
    m = dom(x, y) & np.isfinite(x)
    err = ulp_err(out[m, col], fn(x[m]))
    assert m.sum() > 10000 and err.max() <= bar, (err.max(), x[m][err.argmax()])


def check_atan2_and_pow(rec, out):
    x, y = rec[:, 0].astype(np.float64), rec[:, 1].astype(np.float64)
    m = np.isfinite(x) & np.isfinite(y) & (x != 0) & (y != 0)
    assert ulp_err(out[m, 4], np.arctan2(y[m], x[m])).max() <= 3.5
    assert ulp_err(out[m, 9], np.arctan2(x[m], y[m])).max() <= 3.5
    g = m & (x >= 1e-6) & (x <= 1e4) & ((rec[:, 1] == np.float32(1.0 / 2.2)) | (rec[:, 1] == np.float32(2.2)))
    assert g.sum() > 10000 and ulp_err(out[g, 8], np.power(x[g], y[g])).max() <= 4.0


@pytest.mark.parametrize("col,fn,dom,bar", ACCURACY_BARS)
def test_accuracy_against_the_correctly_rounded_value(col, fn, dom, bar):
    rec = libm_inputs()
    check_accuracy(rec, cpu_libm(rec), col, fn, dom, bar)


def test_atan2_and_pow_accuracy():
    rec = libm_inputs()
    check_atan2_and_pow(rec, cpu_libm(rec))


def test_special_values():
    f = lambda *v: np.array([v], np.float32)
    o = cpu_libm(np.ascontiguousarray(f(0.0, 1.0)))[0]
    assert o[0] == 0 and o[1] == 1 and o[3] == 0 and o[5] == 1 and o[6] == -np.inf and o[7] == 0 and o[8] == 0
    assert abs(o[2] - np.float32(np.pi / 2)) <= np.spacing(np.float32(1.5))
    o = cpu_libm(np.ascontiguousarray(f(np.nan, 1.0)))[0]
    assert np.isnan(o[[0, 1, 2, 3, 4, 5, 6, 7]]).all()
    o = cpu_libm(np.ascontiguousarray(f(np.inf, 2.0)))[0]
    assert np.isnan(o[0]) and np.isnan(o[1]) and o[5] == np.inf and o[6] == np.inf and o[8] == np.inf
    o = cpu_libm(np.ascontiguousarray(f(-1.0, 0.0)))[0]
    assert o[2] == np.float32(np.pi) and np.isnan(o[6]) and o[8] == 1.0  # acos(-1), log(-1), pow(x, 0)
    o = cpu_libm(np.ascontiguousarray(f(-104.5, 1.0)))[0]
    assert o[5] == 0.0
    o = cpu_libm(np.ascontiguousarray(f(-90.0, 1.0)))[0]
    assert 0 < o[5] < 1e-38 and abs(o[5] / np.exp(-90.0) - 1) < 1e-5  # subnormal result
