"""
ctypes binding of libpsfmc_hip.so (include/psfmc_hip.h) -- the only way the
Python host reaches the GPU.  There is deliberately no CPU fallback: if the
library is missing or no gfx950 device is present, construction raises.
"""
import ctypes
import os

import numpy as np

BACKEND_FUSED = 0
BACKEND_HIPFFT = 1
BACKENDS = {'fused': BACKEND_FUSED, 'hipfft': BACKEND_HIPFFT}

ROW_SKY, ROW_PS, ROW_SERSIC = 1, 4, 9

# sides the fused kernels are built for (psfmc_amd/csrc/psfmc_fft.h FftShape)
FUSED_SIDES = (64, 84, 88, 96, 98, 100, 104, 110, 112, 120, 126, 128, 130, 132, 140, 144, 150, 156,
               160, 168, 176, 180, 192, 196, 200, 208, 210, 220, 224, 240, 250, 252, 256, 260, 264,
               280, 286, 288, 294, 300, 308, 312, 320, 330, 336, 350, 352, 360, 364, 384, 390, 392,
               400, 416, 420, 440, 448, 480, 484, 500, 504, 512, 520, 528, 560, 572, 576, 600, 616,
               624, 630, 640, 650, 660, 672, 676, 700, 704, 720, 728, 768, 780, 784, 800, 832, 840,
               896, 900, 960, 1024,
               # round 4: sides above 1024 (three-stage row AND column kernels only: R1 x 8 x 8, R1 = 18 ... 32)
               1152, 1280, 1536, 2048)


def nearest_fused_sides(n):
    """The supported sides around `n`: (largest side <= n or None, smallest side >= n or None) --
    for choosing a cut-out size that stays on the fused kernels."""
    below = [v for v in FUSED_SIDES if v <= n]
    above = [v for v in FUSED_SIDES if v >= n]
    return (below[-1] if below else None), (above[0] if above else None)


# psfmc_fft.h fft3g_pick: the sides whose columns run on the general wave-wide three-stage engine, with
# their (R2, R3) split of the wave's lanes
_COLS3G_SHAPES = {}
for _sides, _shape in (((264, 308, 352, 484), (4, 11)), ((528, 576), (4, 12)),
                       ((312, 364, 416, 520, 572, 624, 676), (4, 13)),
                       ((560, 616, 672, 840, 896), (4, 14)), ((900,), (4, 15)),
                       ((640, 704), (4, 16)), ((330, 440), (5, 11)), ((250, 500), (5, 10)), ((294,), (7, 7)),
                       ((660, 720), (5, 12)),
                       # round 4: the re-surveyed shapes and the second survey's new sides (csrc/psfmc_fft.h fft3g_pick)
                       ((480, 600, 780), (6, 10)), ((392, 504, 728, 784), (7, 8)), ((448,), (8, 7)),
                       ((280, 336), (7, 8)), ((288,), (6, 8)), ((300,), (5, 12)), ((350,), (5, 10)), ((360,), (6, 10)),
                       ((630,), (7, 9)),
                       ((384, 768, 832, 960, 1152, 1280, 1536, 2048), (8, 8))):
    for _n in _sides:
        _COLS3G_SHAPES[_n] = _shape


def column_engine(ny):
    """Which column kernel the fused back end launches for transform side `ny` (psfmc_hip.hip
    launch_cols): 'k_cols3f' (512, 1024, 1536, 2048: the general three-stage engine run forward both ways; round 3's
    power-of-two kernel 'k_cols3' remains behind option cols3 = 3 and for storage='f32'), 'k_cols3g' (the
    sides psfmc_fft.h fft3g_pick lists: ny = R1 * R2 * R3 on R2 * R3 <= 64 lanes) or 'k_cols' (the
    two-stage engine).  Returns (kernel name, (R1, R2, R3) or None)."""
    if ny in (512, 1024, 1536, 2048):
        return 'k_cols3f', (ny // 64, 8, 8)
    if ny in _COLS3G_SHAPES:
        r2, r3 = _COLS3G_SHAPES[ny]
        return 'k_cols3g', (ny // (r2 * r3), r2, r3)
    return 'k_cols', None


def embedding_side(side, psf_side):
    """The smallest transform side an image side the kernels are not built for can be EMBEDDED in: the
    smallest built side >= side + psf_side - 1 (the image, a wrap-around margin of psf_side - 1 pixels,
    zeros: the circular convolution of that length equals the image's own on the image's pixels --
    csrc/psfmc_device.h WrapDesc), or None when there is none (side + psf_side - 1 > 2048).  The library
    may pick a LARGER built side whose kernels are cheaper per walker (psfmc_hip.hip choose_embedding,
    measured table csrc/psfmc_side_costs.h); `Context.get_option('transform_ny' / 'transform_nx')` tells."""
    need = side + psf_side - 1
    fits = [v for v in FUSED_SIDES if v >= need]
    return fits[0] if fits else None


def fused_supports(ny, nx, psf_shape=None):
    """Whether psfmc_ctx_create accepts this image shape for the fused back end: both sides from
    FUSED_SIDES in any combination -- or, with the PSF's (py, px) given, any even side that can be
    embedded in a built one (`embedding_side`)."""
    if ny in FUSED_SIDES and nx in FUSED_SIDES:
        return True
    if psf_shape is None or ny % 2 or nx % 2:
        return False
    return all(side in FUSED_SIDES or embedding_side(side, pk) is not None
               for side, pk in ((ny, psf_shape[0]), (nx, psf_shape[1])))


_LIB_NAME = 'libpsfmc_hip.so'
_lib = None

_c_double_p = ctypes.POINTER(ctypes.c_double)
_c_u8_p = ctypes.POINTER(ctypes.c_uint8)


class NativeError(RuntimeError):
    """A libpsfmc_hip call failed (code + message of psfmc_last_error)."""

    def __init__(self, code, message):
        super(NativeError, self).__init__('libpsfmc_hip error {}: {}'.format(code, message))
        self.code = code


def library_path():
    """The in-tree extension; PSFMC_LIB may name another build of it (A/B runs)."""
    override = os.environ.get('PSFMC_LIB')
    if override:
        return override
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), _LIB_NAME)


def load_library():
    """Load (once) and type the C ABI.  Raises ImportError when the extension
    has not been built -- the product never runs without it."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise ImportError(
            '{} not found: build it with `make -C psfmc_amd/csrc` (or '
            '`python -c "import __graft_entry__ as g; g.build()"`). psfmc_amd has no '
            'CPU fallback.'.format(path))
    # torch (device memory / streams / torch.distributed plumbing) ships its own
    # libamdhip64 with the same SONAME; importing it first makes this process use
    # ONE HIP runtime for both.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
    vp, ci, cd = ctypes.c_void_p, ctypes.c_int, ctypes.c_double
    lib.psfmc_abi_version.restype = ci
    lib.psfmc_abi_version.argtypes = []
    lib.psfmc_last_error.restype = ctypes.c_char_p
    lib.psfmc_last_error.argtypes = []
    lib.psfmc_ctx_create.restype = ci
    lib.psfmc_ctx_create.argtypes = [ctypes.POINTER(vp), ci, ci, ci, _c_double_p, _c_double_p,
                                     _c_u8_p, ci, ci, ci, _c_double_p, _c_double_p, ci, ci, ci, ci]
    lib.psfmc_ctx_destroy.restype = ci
    lib.psfmc_ctx_destroy.argtypes = [vp]
    lib.psfmc_row_len.restype = ci
    lib.psfmc_row_len.argtypes = [vp]
    lib.psfmc_pass_size.restype = ci
    lib.psfmc_pass_size.argtypes = [vp, ci]
    lib.psfmc_eval_batch.restype = ci
    lib.psfmc_eval_batch.argtypes = [vp, ci, _c_double_p, _c_u8_p, _c_double_p]
    lib.psfmc_eval_batch_field.restype = ci
    lib.psfmc_eval_batch_field.argtypes = [vp, ci, ci, _c_double_p, _c_u8_p, _c_double_p]
    lib.psfmc_eval_batch_device.restype = ci
    lib.psfmc_eval_batch_device.argtypes = [vp, ci, vp, vp, vp, vp]
    lib.psfmc_eval_images.restype = ci
    lib.psfmc_eval_images.argtypes = [vp, ci, _c_double_p] + [_c_double_p] * 5
    lib.psfmc_get_spectra.restype = ci
    lib.psfmc_get_spectra.argtypes = [vp, _c_double_p, _c_double_p]
    lib.psfmc_set_option.restype = ci
    lib.psfmc_set_option.argtypes = [vp, ctypes.c_char_p, cd]
    lib.psfmc_get_option.restype = cd
    lib.psfmc_get_option.argtypes = [vp, ctypes.c_char_p]
    ip = ctypes.POINTER(ctypes.c_int)
    lib.psfmc_set_layout.restype = ci
    lib.psfmc_set_layout.argtypes = [vp, ci, ci, ip, _c_double_p, ip, ip, cd, ip, _c_double_p,
                                     _c_double_p, _c_double_p]
    lib.psfmc_eval_theta.restype = ci
    lib.psfmc_eval_theta.argtypes = [vp, ci, _c_double_p, _c_double_p, _c_double_p]
    lib.psfmc_eval_theta_device.restype = ci
    lib.psfmc_eval_theta_device.argtypes = [vp, ci, vp, vp, vp, vp]
    ip = ctypes.POINTER(ctypes.c_int)
    lib.psfmc_ctx_create_fields.restype = ci
    lib.psfmc_ctx_create_fields.argtypes = [ctypes.POINTER(vp), ci, ci, ci, ci, _c_double_p, _c_double_p, _c_u8_p,
                                            ci, ci, ci, _c_double_p, _c_double_p, ci, ci, ci]
    lib.psfmc_set_layout_field.restype = ci
    lib.psfmc_set_layout_field.argtypes = [vp, ci, ci, ci, ip, _c_double_p, ip, ip, ctypes.c_double, ip,
                                           _c_double_p, _c_double_p, _c_double_p]
    lib.psfmc_eval_theta_fields.restype = ci
    lib.psfmc_eval_theta_fields.argtypes = [vp, ci, ip, ip, _c_double_p, _c_double_p, _c_double_p]
    lib.psfmc_eval_theta_device_fields.restype = ci
    lib.psfmc_eval_theta_device_fields.argtypes = [vp, ci, ip, ip, vp, vp, vp, vp]
    lib.psfmc_accumulate_theta.restype = ci
    lib.psfmc_accumulate_theta.argtypes = [vp, ci, _c_double_p]
    lib.psfmc_debug_theta_rows.restype = ci
    lib.psfmc_debug_theta_rows.argtypes = [vp, ci, _c_double_p, _c_double_p, _c_double_p, _c_u8_p]
    lib.psfmc_stretch_run.restype = ci
    lib.psfmc_stretch_run.argtypes = [vp, ci, ci, _c_double_p, _c_double_p, ci, _c_double_p, _c_double_p,
                                      ip, _c_double_p, _c_double_p, _c_double_p,
                                      ctypes.POINTER(ctypes.c_longlong), ci]
    llp = ctypes.POINTER(ctypes.c_longlong)
    lib.psfmc_stretch_run_fields.restype = ci
    lib.psfmc_stretch_run_fields.argtypes = lib.psfmc_stretch_run.argtypes
    lib.psfmc_accumulate_theta_field.restype = ci
    lib.psfmc_accumulate_theta_field.argtypes = [vp, ci, ci, _c_double_p]
    lib.psfmc_reset_accumulated_field.restype = ci
    lib.psfmc_reset_accumulated_field.argtypes = [vp, ci]
    lib.psfmc_get_accumulated_field.restype = ci
    lib.psfmc_get_accumulated_field.argtypes = [vp, ci] + [_c_double_p] * 5 + [llp]
    lib.psfmc_eval_images_field.restype = ci
    lib.psfmc_eval_images_field.argtypes = [vp, ci, ci, _c_double_p] + [_c_double_p] * 5
    lib.psfmc_stretch_open.restype = ci
    lib.psfmc_stretch_open.argtypes = [vp, ci, ci, _c_double_p, _c_double_p, _c_double_p, _c_double_p, ip,
                                       _c_double_p, llp, ci]
    lib.psfmc_stretch_half_eval.restype = ci
    lib.psfmc_stretch_half_eval.argtypes = [vp, ci, ci, ci, ci, vp, vp]
    lib.psfmc_stretch_half_accept.restype = ci
    lib.psfmc_stretch_half_accept.argtypes = [vp, ci, ci, vp, vp]
    lib.psfmc_stretch_accumulate.restype = ci
    lib.psfmc_stretch_accumulate.argtypes = [vp, ci, ci, vp]
    lib.psfmc_stretch_close.restype = ci
    lib.psfmc_stretch_close.argtypes = [vp, _c_double_p, _c_double_p, _c_double_p, _c_double_p, llp, vp]
    lib.psfmc_get_accumulated_sums.restype = ci
    lib.psfmc_get_accumulated_sums.argtypes = [vp, _c_double_p, llp]
    lib.psfmc_set_accumulated_sums.restype = ci
    lib.psfmc_set_accumulated_sums.argtypes = [vp, _c_double_p, ctypes.c_longlong]
    lib.psfmc_accumulate_images.restype = ci
    lib.psfmc_accumulate_images.argtypes = [vp, ci, _c_double_p]
    lib.psfmc_get_accumulated.restype = ci
    lib.psfmc_get_accumulated.argtypes = [vp] + [_c_double_p] * 5 + [ctypes.POINTER(ctypes.c_longlong)]
    lib.psfmc_reset_accumulated.restype = ci
    lib.psfmc_reset_accumulated.argtypes = [vp]
    lib.psfmc_group_create.restype = ci
    lib.psfmc_group_create.argtypes = [ctypes.POINTER(vp), ci, ip, ci, ci, _c_double_p, _c_double_p,
                                       _c_u8_p, ci, ci, ci, _c_double_p, _c_double_p, ci, ci, ci, ci]
    lib.psfmc_group_destroy.restype = ci
    lib.psfmc_group_destroy.argtypes = [vp]
    lib.psfmc_group_size.restype = ci
    lib.psfmc_group_size.argtypes = [vp]
    lib.psfmc_group_set_layout.restype = ci
    lib.psfmc_group_set_layout.argtypes = [vp, ci, ci, ip, _c_double_p, ip, ip, cd, ip, _c_double_p,
                                           _c_double_p, _c_double_p]
    lib.psfmc_group_eval_batch.restype = ci
    lib.psfmc_group_eval_batch.argtypes = [vp, ci, _c_double_p, _c_u8_p, _c_double_p]
    lib.psfmc_group_eval_theta.restype = ci
    lib.psfmc_group_eval_theta.argtypes = [vp, ci, _c_double_p, _c_double_p, _c_double_p]
    lib.psfmc_debug_math.restype = ci
    lib.psfmc_debug_math.argtypes = [ci, ci, ci, _c_double_p, _c_double_p]
    lib.psfmc_debug_sweep.restype = ci
    lib.psfmc_debug_sweep.argtypes = [ci, ci, ctypes.c_size_t, ci, _c_double_p]
    lib.psfmc_debug_valu_rate.restype = ci
    lib.psfmc_debug_valu_rate.argtypes = [ci, ci, ci, _c_double_p]
    if lib.psfmc_abi_version() != 1:
        raise ImportError('libpsfmc_hip ABI version mismatch')
    _lib = lib
    return lib


def _dp(arr):
    return arr.ctypes.data_as(_c_double_p)


def _f64(arr):
    return np.ascontiguousarray(arr, dtype=np.float64)


def debug_math(op, values, device=0):
    """Evaluate a device elementary function ('log2', 'exp2', 'rcp', 'rcp1',
    'exp2_noclamp', 'log2_tab', 'exp2_floor') on an array (test hook for the rasteriser's hand-written fp64 math)."""
    lib = load_library()
    code = {'log2': 0, 'exp2': 1, 'rcp': 2, 'rcp1': 3, 'exp2_noclamp': 4, 'log2_tab': 5, 'exp2_floor': 6}[op]
    x = _f64(np.ravel(values))
    out = np.empty_like(x)
    rc = lib.psfmc_debug_math(int(device), code, x.size, _dp(x), _dp(out))
    if rc != 0:
        raise NativeError(rc, lib.psfmc_last_error().decode('utf-8', 'replace'))
    return out.reshape(np.shape(values))


def debug_pow_tab(values, p, device=0):
    """values ** p through the rasteriser's power tables (k_pow_tables' builder + `fast_pow_tab`, the
    function that replaced log2 + exp2 per Sersic pixel): test hook."""
    lib = load_library()
    x = _f64(np.ravel(values))
    buf = np.concatenate([x, [float(p)]])
    out = np.empty_like(x)
    rc = lib.psfmc_debug_math(int(device), 100, x.size, _dp(buf), _dp(out))
    if rc != 0:
        raise NativeError(rc, lib.psfmc_last_error().decode('utf-8', 'replace'))
    return out.reshape(np.shape(values))


def debug_sweep(mode, nbytes, reps=20, device=0):
    """Average microseconds of a plain memory sweep over `nbytes` of scratch ('write', 'read_write',
    'read'): the traffic of the three kernels of a pass without their arithmetic (measurement hook)."""
    lib = load_library()
    us = ctypes.c_double(0.0)
    rc = lib.psfmc_debug_sweep(int(device), {'write': 0, 'read_write': 1, 'read': 2}[mode], int(nbytes), int(reps),
                               ctypes.byref(us))
    if rc != 0:
        raise NativeError(rc, lib.psfmc_last_error().decode('utf-8', 'replace'))
    return us.value


def debug_valu_rate(waves_per_simd=2, iters=20000, device=0):
    """Nanoseconds per fp64 vector wave-instruction per SIMD with every SIMD of the chip busy
    (psfmc_debug_valu_rate): the VALU ceiling of the path on this GPU, measured."""
    lib = load_library()
    out = ctypes.c_double(0.0)
    rc = lib.psfmc_debug_valu_rate(int(device), int(waves_per_simd), int(iters), ctypes.byref(out))
    if rc != 0:
        raise NativeError(rc, lib.psfmc_last_error().decode('utf-8', 'replace'))
    return float(out.value)


class Context(object):
    """One observed field resident on one GPU (wraps `psfmc_ctx`)."""

    IMAGE_KINDS = ('raw_model', 'convolved_model', 'residual', 'composite_ivm',
                   'point_source_subtracted')

    def __init__(self, sci, obs_var, bad_px, psfs, psf_vars, n_ps, n_sersic,
                 max_walkers=4096, device=0, backend='fused'):
        self._lib = load_library()
        self._ctx = None
        sci = _f64(sci)
        obs_var = _f64(obs_var)
        if sci.ndim != 2 or obs_var.shape != sci.shape:
            raise ValueError('sci / obs_var must be 2-D arrays of one shape')
        bad = np.ascontiguousarray(np.asarray(bad_px).astype(bool), dtype=np.uint8)
        psfs = _f64(psfs)
        psf_vars = _f64(psf_vars)
        if psfs.ndim != 3 or psf_vars.shape != psfs.shape:
            raise ValueError('psfs / psf_vars must be [n_psf, py, px]')
        self.shape = sci.shape
        self.n_psf = psfs.shape[0]
        self.n_ps, self.n_sersic = int(n_ps), int(n_sersic)
        self.max_walkers = int(max_walkers)
        self.device = int(device)
        self.backend = backend
        handle = ctypes.c_void_p()
        rc = self._lib.psfmc_ctx_create(
            ctypes.byref(handle), self.device, sci.shape[0], sci.shape[1], _dp(sci),
            _dp(obs_var), bad.ctypes.data_as(_c_u8_p), self.n_psf, psfs.shape[1],
            psfs.shape[2], _dp(psfs), _dp(psf_vars), self.n_ps, self.n_sersic,
            self.max_walkers, BACKENDS[backend] if isinstance(backend, str) else int(backend))
        self._check(rc)
        self._ctx = handle
        self.row_len = self._lib.psfmc_row_len(self._ctx)

    def _check(self, rc):
        if rc != 0:
            raise NativeError(rc, self._lib.psfmc_last_error().decode('utf-8', 'replace'))

    def close(self):
        if self._ctx is not None:
            self._lib.psfmc_ctx_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ------------------------------------------------------------------
    def _rows(self, rows):
        rows = _f64(rows)
        if rows.ndim != 2 or rows.shape[1] != self.row_len:
            raise ValueError('rows must be [W, {}], got {}'.format(self.row_len, rows.shape))
        if rows.shape[0] > self.max_walkers:
            raise ValueError('W={} exceeds max_walkers={}'.format(rows.shape[0], self.max_walkers))
        return rows

    def pass_size(self, n_w):
        """Walkers per internal pass for a batch of n_w (psfmc_pass_size)."""
        rc = self._lib.psfmc_pass_size(self._ctx, int(n_w))
        if rc < 0:
            self._check(rc)
        return rc

    def loglike(self, rows, skip=None):
        """[W, row_len] derived rows -> [W] log-likelihoods (NaN/inf preserved;
        skipped walkers -inf)."""
        rows = self._rows(rows)
        n_w = rows.shape[0]
        out = np.empty(n_w, dtype=np.float64)
        if n_w == 0:
            return out
        skip_p = None
        if skip is not None:
            skip = np.ascontiguousarray(np.asarray(skip).astype(bool), dtype=np.uint8)
            if skip.shape != (n_w,):
                raise ValueError('skip must be [W]')
            skip_p = skip.ctypes.data_as(_c_u8_p)
        self._check(self._lib.psfmc_eval_batch(self._ctx, n_w, _dp(rows), skip_p, _dp(out)))
        return out

    def loglike_device(self, n_w, d_rows, d_skip, d_out, stream=None):
        """Enqueue on a HIP stream with raw device pointers (ints); no sync."""
        self._check(self._lib.psfmc_eval_batch_device(
            self._ctx, int(n_w), ctypes.c_void_p(d_rows),
            ctypes.c_void_p(d_skip) if d_skip else None, ctypes.c_void_p(d_out),
            ctypes.c_void_p(stream) if stream else None))

    def images(self, rows, kinds=None):
        """dict kind -> [W, ny, nx] for the requested image kinds."""
        rows = self._rows(rows)
        kinds = self.IMAGE_KINDS if kinds is None else tuple(kinds)
        n_w = rows.shape[0]
        bufs, args = {}, []
        for k in self.IMAGE_KINDS:
            if k in kinds:
                bufs[k] = np.empty((n_w,) + self.shape, dtype=np.float64)
                args.append(_dp(bufs[k]))
            else:
                args.append(None)
        unknown = set(kinds) - set(self.IMAGE_KINDS)
        if unknown:
            raise ValueError('unknown image kinds: {}'.format(sorted(unknown)))
        if n_w:
            self._check(self._lib.psfmc_eval_images(self._ctx, n_w, _dp(rows), *args))
        return bufs

    # -- raw emcee vectors (device-side priors and derivation) ---------------
    def set_layout(self, n_sky, n_params, slot_col, slot_const, ps_method, sersic_degrees,
                   mag_zeropoint, family, p0, p1, p2):
        i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
        ipt = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_int))
        slot_col, ps_method, sersic_degrees, family = map(i32, (slot_col, ps_method, sersic_degrees,
                                                                family))
        slot_const, p0, p1, p2 = map(_f64, (slot_const, p0, p1, p2))
        self._check(self._lib.psfmc_set_layout(
            self._ctx, int(n_sky), int(n_params), ipt(slot_col), _dp(slot_const), ipt(ps_method),
            ipt(sersic_degrees), float(mag_zeropoint), ipt(family), _dp(p0), _dp(p1), _dp(p2)))
        self.n_params = int(n_params)

    def _theta(self, theta):
        theta = _f64(theta)
        if theta.ndim != 2 or theta.shape[1] != self.n_params:
            raise ValueError('theta must be [W, {}], got {}'.format(self.n_params, theta.shape))
        if theta.shape[0] > self.max_walkers:
            raise ValueError('W={} exceeds max_walkers={}'.format(theta.shape[0], self.max_walkers))
        return theta

    def logpost_theta(self, theta, extra_lnprior=None):
        """[W, P] raw parameter vectors -> [W] log-posteriors, all on the device
        (never NaN; outside the priors or non-finite likelihood -> -inf)."""
        theta = self._theta(theta)
        n_w = theta.shape[0]
        out = np.empty(n_w)
        if n_w == 0:
            return out
        extra = None
        if extra_lnprior is not None:
            extra = _f64(extra_lnprior)
            if extra.shape != (n_w,):
                raise ValueError('extra_lnprior must be [W]')
        self._check(self._lib.psfmc_eval_theta(self._ctx, n_w, _dp(theta),
                                               _dp(extra) if extra is not None else None, _dp(out)))
        return out

    def logpost_theta_device(self, n_w, d_theta, d_extra, d_out, stream=None):
        self._check(self._lib.psfmc_eval_theta_device(
            self._ctx, int(n_w), ctypes.c_void_p(d_theta), ctypes.c_void_p(d_extra) if d_extra else None,
            ctypes.c_void_p(d_out), ctypes.c_void_p(stream) if stream else None))

    def debug_theta_rows(self, theta):
        """(rows [W, row_len], lnprior [W], skip [W]) as derived on the device."""
        theta = self._theta(theta)
        n_w = theta.shape[0]
        rows = np.zeros((n_w, self.row_len))
        lnprior = np.zeros(n_w)
        skip = np.zeros(n_w, dtype=np.uint8)
        if n_w:
            self._check(self._lib.psfmc_debug_theta_rows(self._ctx, n_w, _dp(theta), _dp(rows),
                                                         _dp(lnprior), skip.ctypes.data_as(_c_u8_p)))
        return rows, lnprior, skip.astype(bool)

    def stretch_run(self, pos, lnprob, z, lz, partner, log_u, naccepted, store=True,
                    accumulate=False):
        """Run z.shape[0] stretch-move iterations on the device (include/psfmc_hip.h
        psfmc_stretch_run).  pos [W,P], lnprob [W] or None, z/lz/log_u/partner
        [n_iter, 2, W/2], naccepted int64 [W].  Returns (pos, lnprob, chain
        [W,n_iter,P] | None, lnprob_chain [W,n_iter] | None); naccepted is updated."""
        pos = np.array(pos, dtype=np.float64, order='C')
        n_w, n_p = pos.shape
        n_iter = int(np.shape(z)[0])
        have = lnprob is not None
        lnp = np.array(lnprob, dtype=np.float64) if have else np.empty(n_w)
        z, lz, log_u = (_f64(a).reshape(n_iter, 2, n_w // 2) for a in (z, lz, log_u))
        partner = np.ascontiguousarray(partner, dtype=np.int32).reshape(n_iter, 2, n_w // 2)
        if naccepted.dtype != np.int64 or naccepted.shape != (n_w,):
            raise ValueError('naccepted must be int64 [W]')
        chain = np.empty((n_w, n_iter, n_p)) if store and n_iter else None
        lnchain = np.empty((n_w, n_iter)) if store and n_iter else None
        self._check(self._lib.psfmc_stretch_run(
            self._ctx, n_w, n_iter, _dp(pos), _dp(lnp), int(have), _dp(z), _dp(lz),
            partner.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), _dp(log_u),
            _dp(chain) if chain is not None else None, _dp(lnchain) if lnchain is not None else None,
            naccepted.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong)), int(bool(accumulate))))
        return pos, lnp, chain, lnchain

    # -- the sampler one half-step at a time (walkers sharded over ranks) -----
    def stretch_open(self, pos, lnprob, z, lz, partner, log_u, naccepted, store=True):
        pos = np.ascontiguousarray(pos, dtype=np.float64)
        n_w, n_iter = pos.shape[0], int(np.shape(z)[0])
        lnp = np.ascontiguousarray(lnprob, dtype=np.float64)
        z, lz, log_u = (_f64(a).reshape(n_iter, 2, n_w // 2) for a in (z, lz, log_u))
        partner = np.ascontiguousarray(partner, dtype=np.int32).reshape(n_iter, 2, n_w // 2)
        if naccepted.dtype != np.int64 or naccepted.shape != (n_w,):
            raise ValueError('naccepted must be int64 [W]')
        self._check(self._lib.psfmc_stretch_open(
            self._ctx, n_w, n_iter, _dp(pos), _dp(lnp), _dp(z), _dp(lz),
            partner.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), _dp(log_u),
            naccepted.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong)), int(bool(store))))
        self._stretch_shape = (n_w, n_iter, pos.shape[1], bool(store))

    def stretch_half_eval(self, it, h, lo, n, d_out, stream=None):
        self._check(self._lib.psfmc_stretch_half_eval(
            self._ctx, int(it), int(h), int(lo), int(n), ctypes.c_void_p(d_out) if d_out else None,
            ctypes.c_void_p(stream) if stream else None))

    def stretch_half_accept(self, it, h, d_newlnp, stream=None):
        self._check(self._lib.psfmc_stretch_half_accept(
            self._ctx, int(it), int(h), ctypes.c_void_p(d_newlnp), ctypes.c_void_p(stream) if stream else None))

    def stretch_accumulate(self, lo, n, stream=None):
        self._check(self._lib.psfmc_stretch_accumulate(self._ctx, int(lo), int(n),
                                                       ctypes.c_void_p(stream) if stream else None))

    def stretch_close(self, naccepted, stream=None):
        n_w, n_iter, n_p, store = self._stretch_shape
        pos, lnp = np.empty((n_w, n_p)), np.empty(n_w)
        chain = np.empty((n_w, n_iter, n_p)) if store else None
        lnchain = np.empty((n_w, n_iter)) if store else None
        self._check(self._lib.psfmc_stretch_close(
            self._ctx, _dp(pos), _dp(lnp), _dp(chain) if store else None, _dp(lnchain) if store else None,
            naccepted.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong)),
            ctypes.c_void_p(stream) if stream else None))
        return pos, lnp, chain, lnchain

    def accumulated_sums(self):
        """(sums [4, ny, nx], count): the raw posterior-image sums of this context."""
        sums = np.empty((4,) + self.shape)
        count = ctypes.c_longlong(0)
        self._check(self._lib.psfmc_get_accumulated_sums(self._ctx, _dp(sums), ctypes.byref(count)))
        return sums, int(count.value)

    def set_accumulated_sums(self, sums, count):
        sums = _f64(sums)
        if sums.shape != (4,) + self.shape:
            raise ValueError('sums must be [4, ny, nx]')
        self._check(self._lib.psfmc_set_accumulated_sums(self._ctx, _dp(sums), int(count)))

    def accumulate(self, rows):
        """Add the five images of every row's walker to the device-resident
        posterior sums."""
        rows = self._rows(rows)
        if len(rows):
            self._check(self._lib.psfmc_accumulate_images(self._ctx, len(rows), _dp(rows)))

    def accumulate_theta(self, theta):
        """Add the images of [W, P] raw parameter vectors to the device-resident sums (records derived
        on the device)."""
        theta = self._theta(theta)
        if len(theta):
            self._check(self._lib.psfmc_accumulate_theta(self._ctx, len(theta), _dp(theta)))

    def accumulated(self):
        """(dict kind -> mean image, sample count) of the device sums."""
        bufs = {k: np.empty(self.shape, dtype=np.float64) for k in self.IMAGE_KINDS}
        count = ctypes.c_longlong(0)
        self._check(self._lib.psfmc_get_accumulated(
            self._ctx, *[_dp(bufs[k]) for k in self.IMAGE_KINDS], ctypes.byref(count)))
        return bufs, int(count.value)

    def reset_accumulated(self):
        self._check(self._lib.psfmc_reset_accumulated(self._ctx))

    def spectra(self):
        """(psf_spec, var_spec) complex128 [n_psf, ny, nx//2+1] as computed on
        the device (== numpy.fft.rfft2 of the centre-padded images)."""
        shp = (self.n_psf, self.shape[0], self.shape[1] // 2 + 1, 2)
        a = np.empty(shp)
        b = np.empty(shp)
        self._check(self._lib.psfmc_get_spectra(self._ctx, _dp(a), _dp(b)))
        return a[..., 0] + 1j * a[..., 1], b[..., 0] + 1j * b[..., 1]

    def set_option(self, key, value):
        self._check(self._lib.psfmc_set_option(self._ctx, key.encode(), float(value)))

    def get_option(self, key):
        return self._lib.psfmc_get_option(self._ctx, key.encode())


class FieldSetContext(object):
    """Several observed fields of ONE shape resident on one GPU in one context (wraps
    `psfmc_ctx_create_fields`): their walkers share the batches, so many small ensembles run at the
    rate of one large one -- log-posteriors, the device-resident sampler (every field's ensemble stepped
    together), posterior-image sums and per-sample images.

    fields: sequence of (sci, obs_var, bad_px, psfs [n_psf, py, px], psf_vars) with the same shapes."""

    def __init__(self, fields, n_ps, n_sersic, max_walkers=4096, device=0):
        self._lib = load_library()
        self._ctx = None
        fields = list(fields)
        if not fields:
            raise ValueError('no field')
        sci = _f64(np.stack([_f64(f[0]) for f in fields]))
        var = _f64(np.stack([_f64(f[1]) for f in fields]))
        bad = np.ascontiguousarray(np.stack([np.asarray(f[2]).astype(bool) for f in fields]), dtype=np.uint8)
        psfs = _f64(np.stack([_f64(f[3]) for f in fields]))
        pvar = _f64(np.stack([_f64(f[4]) for f in fields]))
        if sci.ndim != 3 or var.shape != sci.shape or bad.shape != sci.shape:
            raise ValueError('every field needs sci / obs_var / bad_px of one 2-D shape')
        if psfs.ndim != 4 or pvar.shape != psfs.shape:
            raise ValueError('every field needs psfs / psf_vars of one [n_psf, py, px] shape')
        self.n_fields, self.shape, self.n_psf = len(fields), sci.shape[1:], psfs.shape[1]
        self.n_ps, self.n_sersic = int(n_ps), int(n_sersic)
        self.max_walkers, self.device = int(max_walkers), int(device)
        self.n_params = None
        handle = ctypes.c_void_p()
        rc = self._lib.psfmc_ctx_create_fields(
            ctypes.byref(handle), self.device, sci.shape[1], sci.shape[2], self.n_fields, _dp(sci), _dp(var),
            bad.ctypes.data_as(_c_u8_p), self.n_psf, psfs.shape[2], psfs.shape[3], _dp(psfs), _dp(pvar),
            self.n_ps, self.n_sersic, self.max_walkers)
        self._check(rc)
        self._ctx = handle

    def _check(self, rc):
        if rc != 0:
            raise NativeError(rc, self._lib.psfmc_last_error().decode('utf-8', 'replace'))

    def close(self):
        if self._ctx is not None:
            self._lib.psfmc_ctx_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, key, value):
        self._check(self._lib.psfmc_set_option(self._ctx, key.encode(), float(value)))

    def get_option(self, key):
        return self._lib.psfmc_get_option(self._ctx, key.encode())

    def layout_of(self, field):
        """An object with the `set_layout` of a Context that registers field `field`'s layout."""
        owner = self

        class _Proxy(object):
            def set_layout(self, n_sky, n_params, slot_col, slot_const, ps_method, sersic_degrees,
                           mag_zeropoint, family, p0, p1, p2):
                i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
                ipt = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_int))
                sc, pm, sd, fam = map(i32, (slot_col, ps_method, sersic_degrees, family))
                cst, a0, a1, a2 = map(_f64, (slot_const, p0, p1, p2))
                owner._check(owner._lib.psfmc_set_layout_field(
                    owner._ctx, int(field), int(n_sky), int(n_params), ipt(sc), _dp(cst), ipt(pm), ipt(sd),
                    float(mag_zeropoint), ipt(fam), _dp(a0), _dp(a1), _dp(a2)))
                if owner.n_params not in (None, int(n_params)):
                    raise ValueError('every field must have the same number of free parameters')
                owner.n_params = int(n_params)
        return _Proxy()

    @staticmethod
    def _segments(seg_field, seg_count):
        f = np.ascontiguousarray(seg_field, dtype=np.int32)
        n = np.ascontiguousarray(seg_count, dtype=np.int32)
        if f.shape != n.shape or f.ndim != 1:
            raise ValueError('seg_field / seg_count must be 1-D arrays of one length')
        ipt = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_int))
        return f, n, ipt(f), ipt(n)

    def logpost_theta(self, thetas):
        """thetas: one [W_f, P] array per field (None or empty to leave a field out) -> list of [W_f]
        log-posteriors (empty arrays for the fields left out)."""
        if len(thetas) != self.n_fields:
            raise ValueError('one parameter array per field')
        parts = [(f, _f64(t)) for f, t in enumerate(thetas) if t is not None and len(t)]
        outs = [np.empty(0) for _ in range(self.n_fields)]
        if not parts:
            return outs
        for _, t in parts:
            if t.ndim != 2 or t.shape[1] != self.n_params:
                raise ValueError('theta must be [W, {}]'.format(self.n_params))
        theta = _f64(np.concatenate([t for _, t in parts]))
        if len(theta) > self.max_walkers:
            raise ValueError('W={} exceeds max_walkers={}'.format(len(theta), self.max_walkers))
        f, n, fp, np_ = self._segments([f for f, _ in parts], [len(t) for _, t in parts])
        out = np.empty(len(theta))
        self._check(self._lib.psfmc_eval_theta_fields(self._ctx, len(f), fp, np_, _dp(theta), None, _dp(out)))
        off = 0
        for fld, t in parts:
            outs[fld] = out[off:off + len(t)].copy()
            off += len(t)
        return outs

    def logpost_theta_device(self, seg_field, seg_count, d_theta, d_out, stream=None):
        f, n, fp, np_ = self._segments(seg_field, seg_count)
        self._check(self._lib.psfmc_eval_theta_device_fields(
            self._ctx, len(f), fp, np_, ctypes.c_void_p(d_theta), None, ctypes.c_void_p(d_out),
            ctypes.c_void_p(stream) if stream else None))

    IMAGE_KINDS = Context.IMAGE_KINDS

    def stretch_run(self, pos, lnprob, z, lz, partner, log_u, naccepted, store=True, accumulate=False):
        """Every field's ensemble stepped together on the device (psfmc_stretch_run_fields).
        pos [F, W, P], lnprob [F, W] or None, z / lz / log_u / partner [F, n_iter, 2, W/2], naccepted
        int64 [F, W] (updated).  Returns (pos, lnprob, chain [F, W, n_iter, P] | None, lnprob_chain
        [F, W, n_iter] | None)."""
        pos = np.array(pos, dtype=np.float64, order='C')
        if pos.ndim != 3 or pos.shape[0] != self.n_fields:
            raise ValueError('pos must be [n_fields, W, P]')
        n_f, n_w, n_p = pos.shape
        n_iter = int(np.shape(z)[1])
        have = lnprob is not None
        lnp = np.array(lnprob, dtype=np.float64).reshape(n_f, n_w) if have else np.empty((n_f, n_w))
        z, lz, log_u = (_f64(a).reshape(n_f, n_iter, 2, n_w // 2) for a in (z, lz, log_u))
        partner = np.ascontiguousarray(partner, dtype=np.int32).reshape(n_f, n_iter, 2, n_w // 2)
        if naccepted.dtype != np.int64 or naccepted.shape != (n_f, n_w) or not naccepted.flags.c_contiguous:
            raise ValueError('naccepted must be a contiguous int64 [n_fields, W]')
        chain = np.empty((n_f, n_w, n_iter, n_p)) if store and n_iter else None
        lnchain = np.empty((n_f, n_w, n_iter)) if store and n_iter else None
        self._check(self._lib.psfmc_stretch_run_fields(
            self._ctx, n_w, n_iter, _dp(pos), _dp(lnp), int(have), _dp(z), _dp(lz),
            partner.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), _dp(log_u),
            _dp(chain) if chain is not None else None, _dp(lnchain) if lnchain is not None else None,
            naccepted.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong)), int(bool(accumulate))))
        return pos, lnp, chain, lnchain

    def accumulate_theta(self, field, theta):
        """Add the images of [W, P] raw parameter vectors of one field to its posterior sums."""
        theta = _f64(theta)
        if theta.ndim != 2 or theta.shape[1] != self.n_params:
            raise ValueError('theta must be [W, {}]'.format(self.n_params))
        for lo in range(0, len(theta), self.max_walkers):
            part = _f64(theta[lo:lo + self.max_walkers])
            self._check(self._lib.psfmc_accumulate_theta_field(self._ctx, int(field), len(part), _dp(part)))

    def accumulated(self, field):
        """(dict kind -> mean image, sample count) of one field's posterior sums."""
        bufs = {k: np.empty(self.shape, dtype=np.float64) for k in self.IMAGE_KINDS}
        count = ctypes.c_longlong(0)
        self._check(self._lib.psfmc_get_accumulated_field(
            self._ctx, int(field), *[_dp(bufs[k]) for k in self.IMAGE_KINDS], ctypes.byref(count)))
        return bufs, int(count.value)

    def reset_accumulated(self, field=None):
        """Clear the posterior sums of one field, or (None) of every field."""
        if field is None:
            self._check(self._lib.psfmc_reset_accumulated(self._ctx))
        else:
            self._check(self._lib.psfmc_reset_accumulated_field(self._ctx, int(field)))

    def view(self, field):
        """The part of `Context`'s interface a `MultiComponentModel` uses, for ONE field of this context."""
        return FieldView(self, field)

    def loglike(self, field, rows, skip=None):
        """[W] log-likelihoods of derived rows of one field (`Context.loglike` for a field of this context)."""
        rows = _f64(rows)
        n_w = rows.shape[0]
        if n_w > self.max_walkers:
            raise ValueError('W={} exceeds max_walkers={}'.format(n_w, self.max_walkers))
        out = np.empty(n_w, dtype=np.float64)
        skip_p = None
        if skip is not None:
            skip = np.ascontiguousarray(np.asarray(skip).astype(bool), dtype=np.uint8)
            skip_p = skip.ctypes.data_as(_c_u8_p)
        if n_w:
            self._check(self._lib.psfmc_eval_batch_field(self._ctx, int(field), n_w, _dp(rows), skip_p, _dp(out)))
        return out

    def images(self, field, rows, kinds=None):
        """dict kind -> [W, ny, nx] of the requested per-sample images for derived rows of one field."""
        rows = _f64(rows)
        kinds = self.IMAGE_KINDS if kinds is None else tuple(kinds)
        n_w = rows.shape[0]
        bufs, args = {}, []
        for k in self.IMAGE_KINDS:
            if k in kinds:
                bufs[k] = np.empty((n_w,) + tuple(self.shape), dtype=np.float64)
                args.append(_dp(bufs[k]))
            else:
                args.append(None)
        if n_w:
            self._check(self._lib.psfmc_eval_images_field(self._ctx, int(field), n_w, _dp(rows), *args))
        return bufs


class FieldView(object):
    """One field of a `FieldSetContext` behind the interface of a one-field `Context`, so that a
    `MultiComponentModel` of a `FieldSet` (images, posterior sums, log-posteriors) works on the shared
    context instead of creating its own."""

    IMAGE_KINDS = Context.IMAGE_KINDS

    def __init__(self, owner, field):
        self.owner, self.field = owner, int(field)
        self.shape, self.n_psf = tuple(owner.shape), owner.n_psf
        self.max_walkers, self.device = owner.max_walkers, owner.device

    def close(self):                       # the FieldSet owns the context
        pass

    def set_option(self, key, value):
        self.owner.set_option(key, value)

    def get_option(self, key):
        return self.owner.get_option(key)

    def logpost_theta(self, theta, extra_lnprior=None):
        if extra_lnprior is not None:
            raise ValueError('a FieldSet evaluates priors on the GPU only')
        thetas = [None] * self.owner.n_fields
        thetas[self.field] = theta
        return self.owner.logpost_theta(thetas)[self.field]

    def images(self, rows, kinds=None):
        return self.owner.images(self.field, rows, kinds)

    def loglike(self, rows, skip=None):
        return self.owner.loglike(self.field, rows, skip)

    def __getattr__(self, name):
        # the rest of Context's interface (device-resident sampler, raw-sum exchange, device pointers, ...) has
        # no per-field form on a shared context: say so instead of an AttributeError deep inside a caller
        if callable(getattr(Context, name, None)):
            field = self.__dict__.get('field')

            def _unsupported(*args, **kwargs):
                raise NotImplementedError(
                    "'{}' is not available for field {} of a FieldSet (its context is shared by the set's "
                    "fields): use FieldSet / FieldSetSampler / fitting.model_fields_mcmc, or a "
                    "MultiComponentModel with a context of its own".format(name, field))
            return _unsupported
        raise AttributeError(name)

    def accumulate_theta(self, theta):
        self.owner.accumulate_theta(self.field, theta)

    def accumulated(self):
        return self.owner.accumulated(self.field)

    def reset_accumulated(self):
        self.owner.reset_accumulated(self.field)


class ContextGroup(object):
    """One observed field replicated on several GPUs driven by THIS process (wraps
    `psfmc_group`): walkers are split into contiguous blocks, one per listed device.
    The one-process-per-GPU form (torch.distributed / RCCL) is `psfmc_amd.parallel`."""

    def __init__(self, devices, sci, obs_var, bad_px, psfs, psf_vars, n_ps, n_sersic,
                 max_walkers=4096, backend='fused'):
        self._lib = load_library()
        self._grp = None
        sci, obs_var = _f64(sci), _f64(obs_var)
        bad = np.ascontiguousarray(np.asarray(bad_px).astype(bool), dtype=np.uint8)
        psfs, psf_vars = _f64(psfs), _f64(psf_vars)
        devs = np.ascontiguousarray(devices, dtype=np.int32)
        handle = ctypes.c_void_p()
        rc = self._lib.psfmc_group_create(
            ctypes.byref(handle), len(devs), devs.ctypes.data_as(ctypes.POINTER(ctypes.c_int)),
            sci.shape[0], sci.shape[1], _dp(sci), _dp(obs_var), bad.ctypes.data_as(_c_u8_p),
            psfs.shape[0], psfs.shape[1], psfs.shape[2], _dp(psfs), _dp(psf_vars), int(n_ps),
            int(n_sersic), int(max_walkers), BACKENDS[backend] if isinstance(backend, str) else int(backend))
        self._check(rc)
        self._grp = handle
        self.devices = [int(d) for d in devs]
        self.max_walkers = int(max_walkers)
        self.row_len = ROW_SKY + ROW_PS * int(n_ps) + ROW_SERSIC * int(n_sersic) + 1
        self.n_params = None

    def _check(self, rc):
        if rc != 0:
            raise NativeError(rc, self._lib.psfmc_last_error().decode('utf-8', 'replace'))

    def close(self):
        if self._grp is not None:
            self._lib.psfmc_group_destroy(self._grp)
            self._grp = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_layout(self, n_sky, n_params, slot_col, slot_const, ps_method, sersic_degrees,
                   mag_zeropoint, family, p0, p1, p2):
        i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
        ipt = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_int))
        slot_col, ps_method, sersic_degrees, family = map(i32, (slot_col, ps_method, sersic_degrees, family))
        slot_const, p0, p1, p2 = map(_f64, (slot_const, p0, p1, p2))
        self._check(self._lib.psfmc_group_set_layout(
            self._grp, int(n_sky), int(n_params), ipt(slot_col), _dp(slot_const), ipt(ps_method),
            ipt(sersic_degrees), float(mag_zeropoint), ipt(family), _dp(p0), _dp(p1), _dp(p2)))
        self.n_params = int(n_params)

    def loglike(self, rows, skip=None):
        rows = _f64(rows)
        if rows.ndim != 2 or rows.shape[1] != self.row_len:
            raise ValueError('rows must be [W, {}]'.format(self.row_len))
        out = np.empty(rows.shape[0])
        skip_p = None
        if skip is not None:
            skip = np.ascontiguousarray(np.asarray(skip).astype(bool), dtype=np.uint8)
            skip_p = skip.ctypes.data_as(_c_u8_p)
        if len(rows):
            self._check(self._lib.psfmc_group_eval_batch(self._grp, len(rows), _dp(rows), skip_p, _dp(out)))
        return out

    def logpost_theta(self, theta, extra_lnprior=None):
        theta = _f64(theta)
        if theta.ndim != 2 or theta.shape[1] != self.n_params:
            raise ValueError('theta must be [W, {}]'.format(self.n_params))
        out = np.empty(theta.shape[0])
        extra = _f64(extra_lnprior) if extra_lnprior is not None else None
        if len(theta):
            self._check(self._lib.psfmc_group_eval_theta(self._grp, len(theta), _dp(theta),
                                                         _dp(extra) if extra is not None else None, _dp(out)))
        return out
