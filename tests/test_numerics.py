"""The numeric specification (csrc/ymc_numerics.h) on the CPU build.

Philox4x32-10 is pinned by the Random123 known-answer vectors; log / exp /
sincospi are pinned against libm (numpy) and, where mpmath is importable,
against 120-bit references.  The GPU build of the same header is compared
bit-for-bit with this one in tests/test_gpu_parity.py.
"""
import numpy as np
import pytest

from oracle import oracle as o


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32 10 rounds
    kat = [
        ([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
        ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
        ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
         [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
    ]
    for ctr, key, want in kat:
        assert [int(v) for v in o.philox(ctr, key)] == want


def test_uniforms_are_the_53_bit_mantissas_of_their_philox_words():
    """The accept uniform of transition t is m 2^-53 with m = (hi:lo) >> 11 of words {0,1} (even t) / {2,3} (odd t) of the
    block with counter (chain, 0, t >> 1) -- DESIGN section 2.  The header forms it as hi 2^-32 + (lo >> 11) 2^-53 without
    the 64-bit shift; here the definition is evaluated with Python integers (exact) from the pinned Philox function."""
    seed, chain0, K, t0, nt = 0x9E3779B97F4A7C15, 2**31 - 3, 5, 2**33 + 1, 9
    _, u = o.rng_fill(seed, chain0, K, t0, nt, 1, normals=False)
    key = [seed & 0xFFFFFFFF, seed >> 32]
    for i in range(nt):
        t = t0 + i
        p = t >> 1
        for k in range(K):
            chain = (chain0 + k) & 0xFFFFFFFF
            w = [int(v) for v in o.philox([chain, 0, p & 0xFFFFFFFF, p >> 32], key)]
            lo, hi = (w[2], w[3]) if t & 1 else (w[0], w[1])
            m = ((hi << 32) | lo) >> 11
            assert u[i, k] == m / 2.0**53 and 0 <= m < 2**53     # an exactly representable quotient


def _ulps(got, want):
    want = np.asarray(want, dtype=np.float64)
    return np.abs(got - want) / np.spacing(np.abs(want))


def test_log_against_libm():
    rng = np.random.default_rng(11)
    x = np.concatenate([rng.integers(1, 2**53, size=200000) * 2.0**-53,
                        np.exp(rng.uniform(-700, 700, size=100000)),
                        [2.0**-53, 1.0, 0.5, 1 - 2.0**-53]])
    assert _ulps(o.log(x), np.log(x)).max() <= 2.0
    assert o.log([1.0])[0] == 0.0


def test_exp_against_libm():
    rng = np.random.default_rng(12)
    x = np.concatenate([rng.uniform(-708, 709.7, size=200000), rng.uniform(-40, 1, size=200000)])
    assert _ulps(o.exp(x), np.exp(x)).max() <= 2.0
    got = o.exp([0.0, -745.3, -1e-300, np.inf, -np.inf, 709.79])
    assert got[0] == 1.0 and got[1] == 0.0 and got[2] == 1.0 and np.isinf(got[3]) and got[4] == 0.0
    assert np.isinf(got[5])
    assert np.isnan(o.exp([np.nan])[0])
    # subnormal results: within one subnormal spacing
    xs = rng.uniform(-745.1, -708.4, size=20000)
    assert (np.abs(o.exp(xs) - np.exp(xs)) <= 2 * 4.94065645841246544e-324 + 2 * np.spacing(np.exp(xs))).all()


def test_accept_probability_is_clipped_exp():
    """ymc_accept_prob(d) == min(1, ymc_exp(d)) for d >= -64 with NaN -> 1 (mrw.py:54-57); below -64 it
    is exp(-64) < 2^-53, which decides every u of the 2^-53 grid like the true value does"""
    rng = np.random.default_rng(21)
    d = np.concatenate([rng.uniform(-760, 5, size=300000), rng.normal(size=100000) * 1e-8,
                        [0.0, -0.0, np.inf, -np.inf, np.nan, 800.0, -745.19, -745.21, -708.39, -708.4, 1e-320, -1e-320]])
    e = o.exp(d)
    want = np.where(e < 1.0, e, 1.0)
    got = o.accept_prob(d)
    far = d < -64.0
    assert np.array_equal(got[~far], want[~far])
    assert np.all(got[far] == o.exp(np.array([-64.0]))[0]) and got[far].max() < 2.0**-53
    # identical decisions for every u of the grid, including u = 0
    u = np.concatenate([rng.integers(0, 2**53, size=d.size - 3) * 2.0**-53, [0.0, 2.0**-53, 1 - 2.0**-53]])
    assert np.array_equal(u <= got, u <= want)


def test_division_by_reciprocal_is_the_ieee_quotient():
    """Markstein's correction with y = RN(1/b): random and near-midpoint quotients"""
    rng = np.random.default_rng(22)
    for rep in range(8):
        n = 500000
        b = rng.integers(1, 2**31 if rep % 2 else 5000, size=n).astype(np.float64)
        a = rng.normal(size=n) * np.exp(rng.uniform(-20, 20, size=n))
        if rep >= 4:
            m = rng.integers(2**52, 2**53, size=n).astype(np.float64)
            a = b * (m + 0.5) * 2.0**rng.integers(-60, -40, size=n)
            a = np.nextafter(a, a * rng.choice([0.5, 2.0], size=n))
        assert np.array_equal(o.div_by_recip(a, b), a / b)


def test_sincospi_against_libm_and_exact_points():
    rng = np.random.default_rng(13)
    t = rng.integers(0, 2**53, size=300000) * 2.0**-52  # the Box-Muller argument grid, [0,2)
    s, c = o.sincospi(t)
    # libm on pi*t loses accuracy near zeros of sin/cos; compare absolutely there
    assert np.abs(s - np.sin(np.pi * t)).max() < 1e-15
    assert np.abs(c - np.cos(np.pi * t)).max() < 1e-15
    assert np.abs(s * s + c * c - 1.0).max() < 5e-16
    s, c = o.sincospi([0.0, 0.5, 1.0, 1.5, 0.25])
    assert list(s[:4]) == [0.0, 1.0, 0.0, -1.0] and list(np.abs(c[:4])) == [1.0, 0.0, 1.0, 0.0]
    assert abs(s[4] - np.sqrt(0.5)) < 2e-16 and abs(c[4] - np.sqrt(0.5)) < 2e-16


def test_high_precision_when_mpmath_present():
    mp = pytest.importorskip("mpmath")
    mp.mp.prec = 120
    rng = np.random.default_rng(14)

    def max_ulp(got, ref):
        worst = 0.0
        for a, b in zip(got, ref):
            e = max(mp.floor(mp.log(abs(b), 2)), -1022)
            worst = max(worst, float(abs(mp.mpf(float(a)) - b) / mp.mpf(2) ** (e - 52)))
        return worst

    # the table-driven log: < 1.7 ulp on the draws' domain (0, 1], next to 1 and over the whole range
    x = np.concatenate([rng.integers(1, 2**53, size=3000) * 2.0**-53, 1.0 - rng.integers(1, 2**40, size=1500) * 2.0**-53,
                        np.exp(rng.uniform(-700, 700, size=1500))])
    assert max_ulp(o.log(x), [mp.log(mp.mpf(float(v))) for v in x]) < 1.7
    x = rng.uniform(-745, 709, size=3000)
    assert max_ulp(o.exp(x), [mp.exp(mp.mpf(float(v))) for v in x]) < 1.0
    t = rng.integers(0, 2**53, size=3000) * 2.0**-52
    s, c = o.sincospi(t)
    assert max_ulp(s, [mp.sin(mp.pi * mp.mpf(float(v))) for v in t]) < 2.0
    assert max_ulp(c, [mp.cos(mp.pi * mp.mpf(float(v))) for v in t]) < 2.0


def test_draw_stream_shape_and_moments():
    xi, u = o.rng_fill(123, 5, 64, 0, 2000, 3)
    assert xi.shape == (2000, 64, 3) and u.shape == (2000, 64)
    assert (u >= 0).all() and (u < 1).all()
    assert abs(u.mean() - 0.5) < 3e-3
    assert abs(xi.mean()) < 6e-3 and abs(xi.var() - 1.0) < 1e-2
    # odd d drops the last Box-Muller output, it does not shift anything
    xi2, _ = o.rng_fill(123, 5, 64, 0, 2000, 2, uniforms=False)
    assert np.array_equal(xi[:, :, :2], xi2)
    # addressing: a window of a longer stream / another chain block is the same numbers
    xi3, u3 = o.rng_fill(123, 37, 8, 100, 50, 3)
    assert np.array_equal(xi3, xi[100:150, 32:40]) and np.array_equal(u3, u[100:150, 32:40])
    # streams of different seeds / chains differ
    xi4, _ = o.rng_fill(124, 5, 64, 0, 10, 3, uniforms=False)
    assert not np.array_equal(xi4, xi[:10])
