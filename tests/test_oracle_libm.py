"""The oracle's deterministic libm (oracle/olibm.h): how far is it from the
correctly rounded value and from the host's glibc (what the Rust reference would
call on Linux)?  Parity for transcendental calls is 'unpinned' by the reference;
these bounds document the substitution."""
import numpy as np


def _ulp_diff(a, b):
    ai = a.view(np.int32).astype(np.int64)
    bi = b.view(np.int32).astype(np.int64)
    ai = np.where(ai < 0, -(ai & 0x7FFFFFFF), ai)
    bi = np.where(bi < 0, -(bi & 0x7FFFFFFF), bi)
    return np.abs(ai - bi)


def test_libm_is_correctly_rounded_on_samples_and_within_1ulp_of_glibc(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(5)
    x = rng.uniform(-7, 7, 20000).astype(np.float32)
    for name, ref64 in (("orc_sinf", np.sin), ("orc_cosf", np.cos), ("orc_tanf", np.tan)):
        got = np.array([getattr(L, name)(float(v)) for v in x], dtype=np.float32)
        cr = ref64(x.astype(np.float64)).astype(np.float32)  # round(f64 libm) = correctly rounded except ~2^-29 of cases
        assert _ulp_diff(got, cr).max() == 0, name
    g32 = {"orc_sinf": np.sin, "orc_cosf": np.cos}
    for name, f in g32.items():
        got = np.array([getattr(L, name)(float(v)) for v in x], dtype=np.float32)
        assert _ulp_diff(got, f(x)).max() <= 1, name  # numpy float32 sin/cos
    p = (np.abs(x) + 1e-6).astype(np.float32)
    got = np.array([L.orc_logf(float(v)) for v in p], dtype=np.float32)
    assert _ulp_diff(got, np.log(p.astype(np.float64)).astype(np.float32)).max() == 0
    c = rng.uniform(-1, 1, 20000).astype(np.float32)
    got = np.array([L.orc_acosf(float(v)) for v in c], dtype=np.float32)
    assert _ulp_diff(got, np.arccos(c.astype(np.float64)).astype(np.float32)).max() == 0
    y = rng.uniform(-3, 3, 20000).astype(np.float32)
    got = np.array([L.orc_atan2f(float(a), float(b)) for a, b in zip(y, x)], dtype=np.float32)
    assert _ulp_diff(got, np.arctan2(y.astype(np.float64), x.astype(np.float64)).astype(np.float32)).max() == 0


def test_libm_special_values(oracle):
    L = oracle.lib()
    assert L.orc_sinf(0.0) == 0.0 and L.orc_cosf(0.0) == 1.0 and L.orc_logf(1.0) == 0.0
    assert np.isnan(L.orc_sinf(float("inf"))) and np.isnan(L.orc_logf(-1.0))
    assert L.orc_logf(0.0) == float("-inf")
    assert L.orc_acosf(1.0) == 0.0 and abs(L.orc_acosf(-1.0) - np.float32(np.pi)) == 0.0
    assert L.orc_atan2f(0.0, -1.0) == np.float32(np.pi) and L.orc_atan2f(1.0, 0.0) == np.float32(np.pi / 2)
