"""CPU model of the rasteriser's power tables (psfmc_amd/csrc/psfmc_device.h: build_pow_table / fast_pow_tab).

The fused rasteriser evaluates (rho^2)^p per Sersic pixel as PE[e] * PB[j] * (1 + r)^p from two per-walker tables
and a degree-5 binomial polynomial instead of a log2 and an exp2 (Sersic.py:121-127 evaluates
exp(-kappa * expm1(log(sq_radii) * radius_pow))).  The device code is checked on the GPU
(tests/test_gpu_parity.py::test_device_math); this file pins the ALGORITHM on the CPU tier with the same
constants the kernels use (the generated mantissa table), in numpy float64 with the device's operation order:
its truncation and rounding budget hold for every Sersic index the reference's priors reach.
"""
import os
import re

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
TABLE = os.path.join(HERE, '..', 'psfmc_amd', 'csrc', 'psfmc_log_table.h')
E_BIAS, N_E = 128, 256


def load_table():
    txt = open(TABLE).read()
    body = txt[txt.index('kLog2Tab[256][2]'):txt.index('};')]
    pairs = re.findall(r'\{(-?0x[0-9a-fp.+-]+), (-?0x[0-9a-fp.+-]+)\}', body)
    assert len(pairs) == 256
    a = np.array([float.fromhex(x) for x, _ in pairs])
    b = np.array([float.fromhex(y) for _, y in pairs])
    return a, b


def build_tables(p, b):
    """k_pow_tables: PB[j] = 2^(p b_j), PE[i] = 2^(p (i - 128)), the product carried exactly (hi + lo)."""
    ld = np.longdouble
    pb = np.exp2(ld(p) * b.astype(ld)).astype(np.float64)
    e = np.arange(N_E) - E_BIAS
    with np.errstate(over='ignore', under='ignore'):
        pe = np.exp2(ld(p) * e.astype(ld)).astype(np.float64)
    return pb, pe


def pow_tab(x, p, a, pb, pe):
    """fast_pow_tab in float64, the device's operation order (an fma is modelled by a longdouble product
    rounded once)."""
    ld = np.longdouble
    m, e = np.frexp(x)                                  # m in [0.5, 1)
    j = ((m * 512.0).astype(np.int64) - 256).clip(0, 255)   # the top eight fraction bits of m
    ei = np.clip(e, -E_BIAS, N_E - E_BIAS - 1) + E_BIAS
    r = (m.astype(ld) * a[j].astype(ld) - 1).astype(np.float64)
    q1 = p
    q2 = q1 * (p - 1.0) * 0.5
    q3 = q2 * (p - 2.0) * (1.0 / 3.0)
    q4 = q3 * (p - 3.0) * 0.25
    q5 = q4 * (p - 4.0) * 0.2
    d = q5
    for q in (q4, q3, q2, q1):
        d = (ld(1) * d * r + q).astype(np.float64)
    d = d * r
    eb = pe[ei] * pb[j]
    return (eb.astype(ld) * d + eb).astype(np.float64)


@pytest.fixture(scope='module')
def table():
    return load_table()


def test_mantissa_table_contract(table):
    """|m a_j - 1| <= 2^-9 over every mantissa cell, b_j = -log2(a_j) (tools/gen_log_table.py)."""
    a, b = table
    lo = 0.5 + np.arange(256) / 512.0
    hi = 0.5 + (np.arange(256) + 1) / 512.0
    assert np.max(np.abs(lo * a - 1)) <= 2.0 ** -9 * (1 + 1e-9)
    assert np.max(np.abs(hi * a - 1)) <= 2.0 ** -9 * (1 + 1e-9)
    assert np.max(np.abs(b + np.log2(a.astype(np.longdouble)).astype(np.float64))) <= 1.2e-16


@pytest.mark.parametrize('n_index,bound', [(0.05, 4e-14), (0.3, 1e-15), (0.5, 1e-15), (1.0, 1e-15), (2.5, 1e-15),
                                           (4.0, 1e-15), (8.0, 1e-15), (60.0, 1e-15)])
def test_power_table_accuracy(table, n_index, bound):
    a, b = table
    p = 0.5 / n_index
    pb, pe = build_tables(p, b)
    rng = np.random.RandomState(11)
    edges = 2.0 ** rng.randint(-30, 30, 512) * (1.0 + rng.randint(0, 256, 512) / 256.0)      # cell borders
    x = np.concatenate([10.0 ** rng.uniform(-30, 12, 40000), edges, np.nextafter(edges, 0),
                        np.nextafter(edges, np.inf), [1.0, 2.0 ** -128, 2.0 ** 126]])
    got = pow_tab(x, p, a, pb, pe)
    ref = (x.astype(np.longdouble) ** np.longdouble(p)).astype(np.float64)
    ok = np.isfinite(ref) & (ref > 1e-300) & np.isfinite(got)
    assert ok.sum() > 0.9 * x.size
    assert np.max(np.abs(got[ok] - ref[ok]) / ref[ok]) <= bound


def test_truncation_budget():
    """binomial(p, 6) 2^-54: the degree-5 polynomial's truncation for the exponents p = 1 / (2n)."""
    from math import comb
    for n_index, want in ((0.25, 1e-17), (0.5, 1e-30), (1.0, 2e-18), (4.0, 2e-18), (0.05, 1.3e-14)):
        p = 0.5 / n_index
        c6 = abs(p * (p - 1) * (p - 2) * (p - 3) * (p - 4) * (p - 5) / 720.0)
        assert c6 * 2.0 ** -54 <= want, (n_index, c6 * 2.0 ** -54)
    assert comb(10, 6) == 210                              # p = 10 (n = 0.05): the worst case quoted in the kernel


def test_exponent_clamp_is_out_of_any_image():
    """The exponent table covers rho^2 in [2^-129, 2^127): a pixel 1e-13 px from the centre of a component
    with r_eff = 1e5 px, or 2e3 px away from one with r_eff = 1e-15 px, is still inside."""
    assert (1e-13 / 1e5) ** 2 > 2.0 ** -(E_BIAS + 1)
    assert (2e3 / 1e-15) ** 2 < 2.0 ** (N_E - E_BIAS - 1)
