"""The oracle against the committed reference outputs (CPU)."""
import numpy as np
import pytest

import helpers
import psfmc_oracle as orc

CASES = ['example', 'synth256', 'synth128x2', 'edge']


@pytest.mark.parametrize('name', CASES)
def test_oracle_matches_reference_lnprob(name):
    case = helpers.load_case(name)
    field = helpers.oracle_field(case)
    layout = helpers.LAYOUT[name]
    has_idx = name in helpers.HAS_PSF_INDEX
    step = 1 if name != 'synth256' else 3          # keep the CPU suite short
    idx = np.arange(0, len(case['params']), step)
    got32, got64 = [], []
    for i in idx:
        prior = case['lnprior'][i]
        if not np.isfinite(prior):
            got32.append(-np.inf)
            got64.append(-np.inf)
            continue
        theta = case['params'][i]
        got32.append(helpers.oracle_loglike(field, layout, theta, has_idx, None) + prior)
        got64.append(helpers.oracle_loglike(field, layout, theta, has_idx) + prior)
    ref = case['lnprob'][idx]
    got32, got64 = np.array(got32), np.array(got64)
    ok = helpers.well_conditioned(name, len(case['params']))[idx]
    # reference semantics (float32 raw accumulator): pinned at rounding level
    assert helpers.rel_err(got32[ok], ref[ok]) <= 1e-12
    assert helpers.rel_err(got32[~ok], ref[~ok]) <= 1e-6
    # all-fp64 pipeline (what the GPU implements): SURVEY note D, <= ~1e-7
    assert helpers.rel_err(got64[ok], ref[ok]) <= 2e-7
    assert helpers.rel_err(got64[~ok], ref[~ok]) <= 1e-6


@pytest.mark.parametrize('name,step', [('synth512x2', 3), ('synth1024x4', 4)])
def test_oracle_matches_reference_at_the_large_configurations(name, step):
    """BASELINE configs 3 and 4 (512^2 / 2 Sersic, 1024^2 / 4 Sersic) pinned to the reference itself: the
    `light` fixtures hold the reference's log-posteriors for vectors on the field tools/synth_field.py
    regenerates from its seed (its checksums are in the fixture).  A subset here (CPU suite time); every vector
    goes through the GPU in tests/test_gpu_headline.py."""
    case, _ = helpers.load_light_case(name)
    field = helpers.oracle_field(case)
    layout = helpers.LAYOUT[name]
    n = len(case['params'])
    idx = sorted(set(list(range(0, n, step)) + [n - 2, n - 1]))       # ... and the two -inf vectors at the end
    got32, got64 = [], []
    for i in idx:
        prior = case['lnprior'][i]
        if not np.isfinite(prior):
            got32.append(-np.inf)
            got64.append(-np.inf)
            continue
        ll32 = helpers.oracle_loglike(field, layout, case['params'][i], False, None)
        ll64 = helpers.oracle_loglike(field, layout, case['params'][i], False)
        got32.append(ll32 + prior if np.isfinite(ll32) else -np.inf)
        got64.append(ll64 + prior if np.isfinite(ll64) else -np.inf)
        # the fixture's fp64 log-likelihood is this oracle's, evaluated next to the reference
        if np.isfinite(ll64):
            assert abs(ll64 - case['loglike_f64'][i]) <= 1e-11 * abs(ll64)
    ref = case['lnprob'][idx]
    assert (ref == -np.inf).sum() >= 2
    assert helpers.rel_err(np.array(got32), ref) <= 1e-12
    assert helpers.rel_err(np.array(got64), ref) <= 1e-6


def test_oracle_images_match_reference():
    case = helpers.load_case('example')
    field = helpers.oracle_field(case)
    comps, psf = helpers.comps_from_theta(helpers.LAYOUT['example'], case['params'][1])
    _, imgs = orc.evaluate(field, comps, psf, raw_dtype=None, want_ps_sub=True)
    for kind in ('raw_model', 'convolved_model', 'residual', 'composite_ivm',
                 'point_source_subtracted'):
        ref = case['img1_' + kind].astype(np.float64)
        assert np.abs(imgs[kind] - ref).max() <= 1e-12 * np.abs(ref).max(), kind


def test_derived_rows_match_golden():
    for name in CASES:
        case = helpers.load_case(name)
        field = helpers.oracle_field(case)
        for i in range(0, len(case['params']), 5):
            if not np.isfinite(case['lnprior'][i]):
                continue
            comps, psf = helpers.comps_from_theta(helpers.LAYOUT[name], case['params'][i],
                                                  name in helpers.HAS_PSF_INDEX)
            row = orc.derived_row(field, comps, psf)
            assert np.allclose(row, case['derived'][i], rtol=1e-13, atol=0)


@pytest.mark.parametrize('tag', ['0p5', '1p0', '3p1', '4p0', '6p5'])
def test_sersic_vs_galfit_fixture(tag):
    """The reference's own (assertion-free) Sersic check, tests/test_components.py:49-118:
    agreement with GALFIT is ~1 % outside the core."""
    gal = helpers.load_case('galfit')
    xc, yc, mag, re, n, ar, pa, zp = gal['pars_' + tag]
    img = np.zeros((128, 128))
    orc.add_sersic(img, (xc - 1, yc - 1), mag, re, re * ar, n, pa, True, zp,
                   orc.array_coords(img.shape))
    assert np.abs(img - gal['psfmc_' + tag]).max() <= 1e-12 * img.max()
    ref = gal['galfit_' + tag].astype(np.float64)
    yy, xx = np.mgrid[0:128, 0:128]
    radius = np.hypot(xx - (xc - 1), yy - (yc - 1))
    sel = (radius > re) & (ref > 1e-2 * ref.max())
    assert (np.abs(img - ref) / ref)[sel].max() < 4e-2
    assert np.abs(img - ref).sum() / ref.sum() < 2e-2


def test_bilinear_point_source_matches_ndimage_shift():
    """The reference's only hard assertion (tests/test_components.py:121-144)."""
    from scipy.ndimage import shift
    ref = np.zeros((5, 5))
    ref[1, 1] = 1.0
    ref = shift(ref, np.array((2.2, 2.7))[::-1] - 1, order=1)
    img = np.zeros((5, 5))
    orc.add_point_source(img, (2.2, 2.7), 0.0, 0.0, orc.array_coords(img.shape), 'bilinear')
    assert np.allclose(ref, img)
