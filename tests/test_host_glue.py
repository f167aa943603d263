"""Host-side glue on CPU: FITS I/O, model-file DSL, packing contract, priors,
derived rows, and the C-ABI library's exported surface (no compute calls)."""
import ctypes
import os
import re

import numpy as np
import pytest

import helpers
import psfmc_oracle as orc
from psfmc_amd import MultiComponentModel, fits_io, engine
from psfmc_amd.distributions import Normal, Uniform, DiscreteUniform, WeibullMinimum
from psfmc_amd.ModelComponents import Sersic, PointSource, Sky

CASES = ['example', 'synth256', 'synth128x2', 'edge']


def test_fits_roundtrip(tmp_path):
    rng = np.random.RandomState(0)
    for dtype in (np.float32, np.float64, np.int16, np.uint8):
        arr = (rng.normal(size=(6, 10)) * 50).astype(dtype)
        path = str(tmp_path / 'a.fits')
        fits_io.write_image(path, arr, header={'MAGZPT': 25.5, 'OBJECT': 'x y'})
        back, hdr = fits_io.read_image(path, with_header=True)
        assert back.dtype == arr.dtype and np.array_equal(back, arr)
        assert hdr['MAGZPT'] == 25.5 and hdr['OBJECT'] == 'x y'
        assert os.path.getsize(path) % 2880 == 0


def test_fits_reads_reference_example_files():
    ex = os.path.join(helpers.GOLDEN, 'example')
    case = helpers.load_case('example')
    assert np.array_equal(fits_io.read_image(os.path.join(ex, 'sci_J0005-0006.fits')), case['sci'])
    assert np.array_equal(fits_io.read_image(os.path.join(ex, 'ivm_psf.fits')), case['psf_ivms'][0])


@pytest.mark.parametrize('name', CASES)
def test_packing_priors_and_rows_match_reference(name, tmp_path):
    case = helpers.load_case(name)
    model = helpers.build_model(name, case, tmp_path)
    assert model.param_names == [str(s) for s in case['param_names']]
    assert model.num_params == case['params'].shape[1]
    lnprior = model.log_priors_batch(case['params'])
    assert helpers.rel_err(lnprior, case['lnprior']) <= 1e-13
    fin = np.isfinite(case['lnprior'])
    rows = model.derived_rows(case['params'][fin])
    assert np.allclose(rows, case['derived'][fin], rtol=1e-13, atol=0)
    # setup arrays equal the oracle's restatement of preprocess_obs / preprocess_psf
    field = helpers.oracle_field(case)
    assert np.array_equal(model.config.bad_px, field.bad_px)
    assert np.array_equal(model.config.obs_var, field.obs_var)
    for a, b in zip(model.config.psf_selector.psf_data, field.psf_list):
        assert np.array_equal(a, b)
    for a, b in zip(model.config.psf_selector.psf_var, field.var_list):
        assert np.array_equal(a, b)


def test_scalar_and_batch_priors_agree(tmp_path):
    case = helpers.load_case('edge')
    model = helpers.build_model('edge', case, tmp_path)
    batch = model.log_priors_batch(case['params'])
    for i in range(0, len(case['params']), 4):
        model.param_values = case['params'][i]
        one = model.log_priors()
        assert (one == batch[i]) or abs(one - batch[i]) <= 1e-13 * abs(one)


def test_init_params_from_priors_are_valid(tmp_path):
    case = helpers.load_case('example')
    model = helpers.build_model('example', case, tmp_path)
    np.random.seed(3)
    p0 = model.init_params_from_priors(12)
    assert p0.shape == (12, 18)
    assert np.isfinite(model.log_priors_batch(p0)).all()


def test_constants_are_not_packed():
    ser = Sersic(xy=(3.0, 4.0), mag=Uniform(loc=20, scale=2), reff=5.0, reff_b=Uniform(loc=1, scale=3),
                 index=WeibullMinimum(c=1.5, scale=4), angle=10.0, angle_degrees=True)
    assert ser.free_names() == ['index', 'mag', 'reff_b']
    vals = ser.values_batch(np.array([[2.0, 21.0, 3.0], [1.0, 20.5, 6.0]]))
    assert vals['xy'].shape == (2, 2) and vals['reff'].tolist() == [5.0, 5.0]
    lp = ser.log_priors_batch(np.array([[2.0, 21.0, 3.0], [1.0, 20.5, 6.0]]))
    assert np.isfinite(lp[0]) and lp[1] == -np.inf      # reff_b=6 > reff=5


def test_discrete_prior_rounds():
    d = DiscreteUniform(low=0, high=3)
    d.value = np.array([1.6])
    assert d.value == 2 and isinstance(d.value, int)
    assert np.allclose(d.logp_batch(np.array([[0.2], [2.4], [2.6]])),
                       [np.log(1 / 3.), np.log(1 / 3.), -np.inf])


def test_unknown_component_is_rejected(tmp_path):
    from psfmc_amd.ModelComponents.ComponentBase import ComponentBase
    from psfmc_amd.ModelComponents import Configuration
    case = helpers.load_case('synth128x2')

    class Blob(ComponentBase):
        pass
    cfg = Configuration(case['sci'], case['ivm'], case['psfs'][0], case['psf_ivms'][0])
    with pytest.raises(NotImplementedError):
        MultiComponentModel([cfg, Blob()])
    with pytest.raises(ValueError):
        MultiComponentModel([Sky(adu=0.0)])


def test_model_file_redirects_reference_imports(tmp_path):
    case = helpers.load_case('synth128x2')
    helpers.write_case_files('synth128x2', case, tmp_path)
    text = ('import psfMC.distributions\n'
            'from psfMC.ModelComponents import Configuration, Sky\n'
            'from psfMC.distributions import Normal\n'
            "Configuration(obs_file='sci.fits', obsivm_file='ivm.fits', psf_files='psf.fits',\n"
            "              psfivm_files='psf_ivm.fits', mag_zeropoint=25.0)\n"
            'for level in (0.0,):\n'
            '    Sky(adu=Normal(loc=level, scale=0.1))\n')
    path = tmp_path / 'm.py'
    path.write_text(text)
    model = MultiComponentModel(str(path))
    assert model.param_names == ['0_Sky_adu']
    assert os.getcwd() != str(tmp_path)


def test_library_exports_every_declared_symbol():
    """include/psfmc_hip.h <-> libpsfmc_hip.so (load only; no GPU calls)."""
    root = os.path.dirname(helpers.GOLDEN.rstrip('/').rsplit('/tests', 1)[0] + '/x')
    header = open(os.path.join(root, 'include', 'psfmc_hip.h')).read()
    declared = set(re.findall(r'\b(psfmc_[a-z_]+)\s*\(', header))
    assert {'psfmc_ctx_create', 'psfmc_eval_batch', 'psfmc_eval_batch_device',
            'psfmc_eval_images', 'psfmc_ctx_destroy', 'psfmc_last_error'} <= declared
    lib = engine.load_library()
    for sym in declared:
        assert hasattr(lib, sym), sym
    assert lib.psfmc_abi_version() == 1
    assert isinstance(lib, ctypes.CDLL)


def test_native_argument_errors_without_gpu():
    """Argument validation happens before any device is touched."""
    lib = engine.load_library()
    handle = ctypes.c_void_p()
    a = np.zeros((4, 4))
    bad = np.zeros((4, 4), dtype=np.uint8)
    dp = ctypes.POINTER(ctypes.c_double)
    rc = lib.psfmc_ctx_create(ctypes.byref(handle), 0, 5, 4, a.ctypes.data_as(dp),
                              a.ctypes.data_as(dp), bad.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)),
                              1, 2, 2, a.ctypes.data_as(dp), a.ctypes.data_as(dp), 1, 1, 8, 1)
    assert rc == -1 and b'even' in lib.psfmc_last_error()
    assert handle.value is None


def test_no_gpu_fails_loudly(tmp_path):
    """The product has no CPU fallback: without a gfx950 device the context
    cannot be created and the error says so."""
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is present')
    case = helpers.load_case('synth128x2')
    model = helpers.build_model('synth128x2', case, tmp_path)
    with pytest.raises(engine.NativeError) as err:
        model.log_posterior_batch(case['params'][:2])
    assert err.value.code == -4 and 'HIP device' in str(err.value)


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(engine, '_lib', None)
    monkeypatch.setenv('PSFMC_LIB', '/nonexistent/libpsfmc_hip.so')
    with pytest.raises(ImportError) as err:
        engine.load_library()
    assert 'no' in str(err.value).lower() and 'CPU fallback' in str(err.value)


def test_ds9_region_mask(tmp_path):
    """ds9 region masks (circle / -circle / box in image coordinates).  Semantics as
    documented in utils.mask_from_file; parity with pyregion is unpinned (not installed)."""
    from psfmc_amd.utils import mask_from_file, region_filter
    reg = tmp_path / 'm.reg'
    reg.write_text('# Region file format: DS9 version 4.1\n'
                   'global color=green dashlist=8 3 width=1\nimage\n'
                   'circle(33,33,10)\n-circle(36,33,3)\nbox(10,50,6,4,0)\n')
    inside = region_filter(str(reg), (64, 64))
    assert inside[32, 30] and not inside[32, 35] and not inside[32, 44] and inside[32, 42]
    assert inside[49, 9] and inside[49 + 2, 9 + 3] and not inside[49 + 3, 9]
    bad = mask_from_file(str(reg), {}, (64, 64))
    assert np.array_equal(bad, ~inside)
    # the reference's example region file parses too
    ex = tmp_path / 'ex.reg'
    ex.write_text('image\ncircle(64.540771,64.079391,55.620614)\n-circle(111.37667,58.936343,11.905084)\n')
    keep = region_filter(str(ex), (128, 128))
    assert keep[63, 63] and not keep[0, 0] and not keep[58, 110]
    # rotated ellipse / box, annulus, polygon (hand-computed on pixel centres; ds9 is 1-based)
    more = tmp_path / 'more.reg'
    more.write_text('image\nellipse(33,33,20,5,90)\n-annulus(33,33,0,2)\npolygon(5,5,15,5,15,15,5,15)\n'
                    'box(50,12,10,2,45)\n')
    m = region_filter(str(more), (64, 64))
    assert m[32 + 15, 32] and not m[32, 32 + 15]          # the long axis points along y after 90 degrees
    assert not m[32, 32] and not m[33, 32] and m[36, 32]  # the centre is cut out by the annulus
    assert m[9, 9] and m[4, 4] and m[13, 13] and not m[15, 15] and not m[9, 20]   # square polygon
    assert m[11 + 3, 49 + 3] and not m[11 + 3, 49 - 3]    # the box lies along the +x +y diagonal
    # unsupported content -> ignored with a warning, like the reference without pyregion
    other = tmp_path / 'o.reg'
    other.write_text('fk5\ncircle(10:00:00,+02:00:00,5")\n')
    with pytest.warns(UserWarning):
        assert mask_from_file(str(other), {}, (8, 8)) is None


def test_nearest_fused_sides():
    from psfmc_amd import engine
    assert engine.nearest_fused_sides(256) == (256, 256)
    assert engine.nearest_fused_sides(134) == (132, 140)
    assert engine.nearest_fused_sides(50) == (None, 64)
    assert engine.nearest_fused_sides(2000) == (1536, 2048) and engine.nearest_fused_sides(3000) == (2048, None)
    assert engine.fused_supports(140, 256) and not engine.fused_supports(134, 256)
    assert engine.fused_supports(134, 256, (64, 64)) and engine.embedding_side(134, 64) == 200
    assert engine.embedding_side(1000, 64) == 1152 and engine.embedding_side(170, 33) == 208
    assert engine.embedding_side(1985, 64) == 2048 and engine.embedding_side(1986, 64) is None


def test_side_tables_agree_between_python_and_the_kernels():
    """The per-side tables exist in several places: the list of built sides (engine.FUSED_SIDES, two copies in
    psfmc_hip.hip), the sides whose columns run on the general three-stage engine with their (R2, R3) split
    (psfmc_fft.h fft3g_pick, mirrored by engine.column_engine), and the measured cost table the embedding ranks
    candidate sides with (psfmc_side_costs.h).  They are parsed from the sources here and compared."""
    import re
    from psfmc_amd import engine
    csrc = os.path.join(os.path.dirname(os.path.abspath(engine.__file__)), 'csrc')
    hip = open(os.path.join(csrc, 'psfmc_hip.hip')).read()
    lists = re.findall(r'static const int (?:sides|kFusedSides)\[\] = \{([0-9, ]+)\}', hip)
    assert len(lists) == 2
    for text in lists:
        assert tuple(int(v) for v in text.split(',')) == tuple(engine.FUSED_SIDES)
    fft = open(os.path.join(csrc, 'psfmc_fft.h')).read()
    body = fft[fft.index('constexpr Fft3gPick fft3g_pick(int n)'):]
    body = body[:body.index('default:')]
    picks = {}
    for cases, r2, r3 in re.findall(r'((?:case \d+:\s*)+)return \{(\d+), (\d+)\};', body):
        for n in re.findall(r'case (\d+):', cases):
            picks[int(n)] = (int(r2), int(r3))
    assert picks and set(picks) <= set(engine.FUSED_SIDES)
    for n in engine.FUSED_SIDES:
        name, shape = engine.column_engine(n)
        if n in (512, 1024, 1536, 2048):
            assert name == 'k_cols3f' and picks[n] == (8, 8) and shape == (n // 64, 8, 8)
        elif n in picks:
            r2, r3 = picks[n]
            assert name == 'k_cols3g' and shape == (n // (r2 * r3), r2, r3), n
            assert n % (r2 * r3) == 0 and r2 * r3 <= 64 and 4 <= n // (r2 * r3) <= (16 if n <= 1024 else 32)
        else:
            assert name == 'k_cols' and shape is None, n
    costs = open(os.path.join(csrc, 'psfmc_side_costs.h')).read()
    rows = re.findall(r'\{(\d+), ([0-9.]+)f, ([0-9.]+)f\}', costs)
    assert tuple(int(r[0]) for r in rows) == tuple(engine.FUSED_SIDES)
    assert all(5.0 < float(r[1]) < 40.0 and 3.0 < float(r[2]) < 30.0 for r in rows)      # picoseconds per pixel per walker
