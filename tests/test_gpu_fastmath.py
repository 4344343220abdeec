"""csrc/flux_math.h on the device (through flux_debug_fastmath) against correctly rounded references
(mpmath at 40 digits / numpy), in units in the last place.  These functions carry MATH_FAST's accuracy
claim: full double precision to a couple of ulp on the render loop's operand ranges."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RSQRT, SQRT, DIV, LOG2, EXP2, POW, SIN2PI, COS2PI, RAW_RSQ, RAW_RCP = range(10)


def ulp_err(got, want):
    want = np.asarray(want, dtype=np.float64)
    return np.abs(got - want) / np.spacing(np.abs(want))


def mp_map(fn, *arrs):
    import mpmath as mp
    mp.mp.dps = 40
    return np.array([float(fn(*[mp.mpf(float(v)) for v in vals])) for vals in zip(*arrs)])


@pytest.fixture(scope="module")
def rng():
    return np.random.default_rng(20261003)


def test_rsqrt_sqrt_div(flux, rng):
    x = np.concatenate([rng.uniform(1e-6, 4.0, 20000), 10.0 ** rng.uniform(-30, 30, 20000)])
    import mpmath as mp
    want = mp_map(lambda v: 1 / mp.sqrt(v), x)
    assert ulp_err(flux.debug_fastmath(RSQRT, x), want).max() <= 2.0
    assert ulp_err(flux.debug_fastmath(SQRT, x), np.sqrt(x)).max() <= 1.0
    assert np.array_equal(flux.debug_fastmath(SQRT, np.array([0.0, 1.0, 4.0, 2.25])), [0.0, 1.0, 2.0, 1.5])
    # tiny negatives (1 - c*c or a discriminant rounded below zero) are 0, never NaN
    neg = -np.concatenate([10.0 ** rng.uniform(-320, -10, 2000), [1e-16, 2.2e-16, 5e-324, 0.0]])
    assert np.array_equal(flux.debug_fastmath(SQRT, neg), np.zeros_like(neg))
    a = rng.uniform(-10, 10, 40000)
    b = np.concatenate([rng.uniform(0.5, 4.0, 20000), 10.0 ** rng.uniform(-20, 20, 20000)])
    assert ulp_err(flux.debug_fastmath(DIV, a, b), a / b).max() <= 1.0
    # the hardware seeds the refinements start from (recorded, loosely bounded)
    assert (np.abs(flux.debug_fastmath(RAW_RSQ, x) * np.sqrt(x) - 1.0)).max() < 1e-6
    assert (np.abs(flux.debug_fastmath(RAW_RCP, b) * b - 1.0)).max() < 1e-6


def test_log2_exp2(flux, rng):
    import mpmath as mp
    x = np.concatenate([rng.uniform(0.0, 1.0, 20000) + 1e-300, 1.0 - 10.0 ** rng.uniform(-16, -1, 10000),
                        10.0 ** rng.uniform(-300, 3, 10000)])
    want = mp_map(lambda v: mp.log(v, 2), x)
    got = flux.debug_fastmath(LOG2, x)
    nz = want != 0
    assert ulp_err(got[nz], want[nz]).max() <= 4.0
    assert flux.debug_fastmath(LOG2, np.array([1.0, 2.0, 0.5, 8.0])).tolist() == [0.0, 1.0, -1.0, 3.0]
    t = np.concatenate([rng.uniform(-1000, 10, 30000), rng.uniform(-1.0, 1.0, 10000)])
    want = mp_map(lambda v: mp.power(2, v), t)
    assert ulp_err(flux.debug_fastmath(EXP2, t), want).max() <= 2.0
    assert flux.debug_fastmath(EXP2, np.array([0.0, -1.0, 3.0, -2000.0, -np.inf])).tolist() == [1.0, 0.5, 8.0, 0.0, 0.0]


def test_pow_render_domain(flux, rng):
    """pow as the render loop uses it: (1 - y)^(1/(e+1)) and cos^e for the demo exponents."""
    import mpmath as mp
    for e in (10.0, 100.0, 1e4, 1e5):
        y = rng.uniform(0.0, 1.0, 8000)
        base = 1.0 - y
        inv = np.full_like(base, 1.0 / (e + 1.0))
        want = mp_map(lambda b, p: mp.power(b, p) if b > 0 else mp.mpf(0), base, inv)
        got = flux.debug_fastmath(POW, base, inv)
        assert ulp_err(got, want).max() <= 4.0
        # lobe = cos^e: relative error grows with |e * ln cos| (the argument of exp2 carries it);
        # 1e-12 relative is what the estimator needs (lobe cancels in f*s), 1e-13 is what we get
        ee = np.full_like(got, e)
        want_l = mp_map(lambda b, p: mp.power(b, p), got, ee)
        got_l = flux.debug_fastmath(POW, got, ee)
        ok = want_l > 1e-300
        assert (np.abs(got_l[ok] - want_l[ok]) / want_l[ok]).max() < 2e-13
    assert flux.debug_fastmath(POW, np.array([0.0, 1.0, 0.25]), np.array([0.5, 123.0, 0.5])).tolist() == [0.0, 1.0, 0.5]


def test_sincos_2pi(flux, rng):
    import mpmath as mp
    x = np.concatenate([rng.uniform(0.0, 1.0, 30000), np.array([0.0, 0.125, 0.25, 0.375, 0.5, 0.625, 0.75, 0.875, 1.0]),
                        np.nextafter(np.arange(1, 8) / 8.0, 0.0), np.nextafter(np.arange(1, 8) / 8.0, 1.0)])
    ws = mp_map(lambda v: mp.sin(2 * mp.pi * v), x)
    wc = mp_map(lambda v: mp.cos(2 * mp.pi * v), x)
    gs, gc = flux.debug_fastmath(SIN2PI, x), flux.debug_fastmath(COS2PI, x)
    assert np.abs(gs - ws).max() < 4e-16 and np.abs(gc - wc).max() < 4e-16
    exact = flux.debug_fastmath(SIN2PI, np.array([0.0, 0.25, 0.5, 0.75, 1.0]))
    assert exact.tolist() == [0.0, 1.0, 0.0, -1.0, 0.0] or np.abs(exact - [0, 1, 0, -1, 0]).max() == 0.0
