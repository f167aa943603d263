"""
Composite model + the batched log-posterior -- the drop-in for
`psfMC.models.MultiComponentModel` (psfMC/models.py).

Host side (this file): model-file parsing, the parameter packing contract,
vectorised priors with the non-finite early-out, derivation of the per-walker
scalars the GPU needs (flux, Sersic b_n, surface brightness at r_e, inverse
ellipse matrix), NaN -> -inf mapping.  Device side (libpsfmc_hip): everything
from models.py:213 to :236 for all walkers of a batch at once.
"""
import copy

import numpy as np

from .ModelComponents import Configuration, PointSource, Sersic, Sky
from .ModelComponents.ComponentBase import ComponentBase
from .ModelComponents.PointSource import SHIFT_METHODS
from .ModelComponents.PSFSelector import PSFSelector
from .model_parser import component_list_from_file
from .utils import mag_to_flux
from . import engine

IMAGE_KINDS = engine.Context.IMAGE_KINDS


def _device_prior(prior, width):
    """(family, p0, p1, p2) if the library evaluates this prior itself
    (include/psfmc_hip.h PSFMC_PRIOR_*), else None."""
    rv = getattr(prior, 'rv_frozen', None)
    if rv is None:
        return None
    try:
        shapes, loc, scale = rv.dist._parse_args(*rv.args, **rv.kwds)
    except Exception:
        return None
    name = rv.dist.name
    ok_size = all(np.size(v) in (1, width) for v in (loc, scale) + tuple(shapes))
    if not ok_size:
        return None
    if name == 'uniform':
        return 1, loc, scale, 0.0
    if name == 'norm':
        return 2, loc, scale, 0.0
    if name == 'weibull_min':
        return 3, shapes[0], loc, scale
    if name == 'randint' and np.all(np.asarray(loc) == 0):
        return 4, shapes[0], shapes[1], 0.0
    return None


class MultiComponentModel(object):
    """A 2-D surface-brightness model made of components (Sky, PointSource,
    Sersic) plus one Configuration, given as a model file or a list
    (reference: models.py:9-61).

    device / backend / max_walkers configure the GPU context, which is created
    on first use.  backend: 'fused' (hand-written FFT kernels; sides from
    `engine.FUSED_SIDES`: the powers of two 64...1024 and the even 5-smooth sides in
    between), 'hipfft' (any even size) or 'auto' (fused whenever the shape allows).
    storage: 'f64' (default) or 'f32' -- keep the intermediate half-spectra of the fused path as
    complex64 while all arithmetic stays fp64: half the memory traffic, log-posteriors good to
    ~1e-7 relative (the class of the reference's own float32 raw model, models.py:249) instead
    of ~1e-15; power-of-two sides only.
    """

    def __init__(self, components, device=0, backend='auto', max_walkers=4096, storage='f64'):
        np.seterr(divide='ignore')
        if isinstance(components, str):
            try:
                components = component_list_from_file(components)
            except IOError as err:
                raise IOError('Unable to open model file {}. Does it exist? ({})'
                              .format(components, err))
        components = list(components)
        configs = [c for c in components if isinstance(c, Configuration)]
        if not configs:
            raise ValueError('Unable to find the Configuration component, '
                             'required for setting up input images.')
        config = configs[-1]
        components.remove(config)
        components.append(config.psf_selector)      # always last (models.py:37-38)
        for count, comp in enumerate(components):
            comp.update_stochastic_names(count=count)
            if not isinstance(comp, PSFSelector) and comp.device_kind is None:
                raise NotImplementedError(
                    'component {} has no GPU rasteriser; supported: Sky, '
                    'PointSource, Sersic'.format(type(comp).__name__))

        self.config = config
        self.components = components
        self.raw_model_components = [c for c in components
                                     if c.device_kind is not None]
        self.psf_comps = [c for c in components if isinstance(c, PointSource)]
        self.obs_header = config.obs_header

        # column ranges of each component in the emcee vector
        self._spans, pos = [], 0
        for comp in components:
            n = comp.num_stochastics()
            self._spans.append(slice(pos, pos + n))
            pos += n
        self._num_params = pos
        self._param_vector = np.zeros(pos)

        self._sky = [c for c in components if isinstance(c, Sky)]
        self._ps = self.psf_comps
        self._sersic = [c for c in components if isinstance(c, Sersic)]
        if backend == 'auto':
            ny, nx = config.obs_data.shape
            psf_shape = np.shape(config.psf_selector.psf_data[0])
            backend = 'fused' if engine.fused_supports(ny, nx, psf_shape) else 'hipfft'
        if storage not in ('f64', 'f32'):
            raise ValueError("storage must be 'f64' or 'f32'")
        self._device, self._backend, self._storage = device, backend, storage
        self._max_walkers = int(max_walkers)
        self._engine = None

        self.blob_images = False      # log_posterior() returns image blobs
        self.posterior_images = {}
        self.accumulated_samples = 0
        self._device_samples = 0
        self.reset_images()

    # -- engine --------------------------------------------------------------
    @property
    def engine(self):
        if self._engine is None:
            sel = self.config.psf_selector
            self._engine = engine.Context(
                self.config.obs_data, self.config.obs_var, self.config.bad_px,
                np.stack(sel.psf_data), np.stack(sel.psf_var),
                n_ps=len(self._ps), n_sersic=len(self._sersic),
                # small ensembles: room for the device sampler's whole-iteration launches (3 proposal sets of
                # half an ensemble each, psfmc_hip.hip stretch_run_impl)
                max_walkers=self._max_walkers + (self._max_walkers // 2 + 1 if self._max_walkers <= 512 else 0),
                device=self._device, backend=self._backend)
            if self._storage == 'f32':
                self._engine.set_option('storage_f32', 1)
            self._register_layout(self._engine)
        return self._engine

    def _register_layout(self, eng):
        """Hand the parameter layout and the priors to the library so that raw
        emcee vectors can be evaluated without host arithmetic.  Priors of
        families the library does not know stay on the host (`_host_priors`)."""
        col_of = {}                                   # (component id, attr, element) -> column
        family = np.zeros(self.num_params, dtype=np.int32)
        p0, p1, p2 = (np.zeros(self.num_params) for _ in range(3))
        self._host_priors = []                        # (prior, column slice)
        for comp, span in zip(self.components, self._spans):
            pos = span.start
            for name, width in zip(comp.free_names(), comp.stochastic_lens()):
                prior = comp._priors[name]
                desc = _device_prior(prior, width)
                if desc is None:
                    self._host_priors.append((prior, slice(pos, pos + width)))
                for j in range(width):
                    col_of[(id(comp), name, j)] = pos + j
                    if desc is not None:
                        family[pos + j] = desc[0]
                        p0[pos + j], p1[pos + j], p2[pos + j] = (np.ravel(v)[j if np.size(v) > 1 else 0]
                                                                 for v in desc[1:])
                pos += width
        slot_col, slot_const = [], []

        def add(comp, name, elem=0):
            key = (id(comp), name, elem)
            if key in col_of:
                slot_col.append(col_of[key])
                slot_const.append(0.0)
            else:
                slot_col.append(-1)
                slot_const.append(float(np.ravel(comp._constants[name])[elem]))
        for c in self._sky:
            add(c, 'adu')
        for c in self._ps:
            add(c, 'mag'); add(c, 'xy', 0); add(c, 'xy', 1)
        for c in self._sersic:
            for name in ('angle', 'index', 'mag', 'reff', 'reff_b'):
                add(c, name)
            add(c, 'xy', 0); add(c, 'xy', 1)
        add(self.config.psf_selector, 'psf_index')
        eng.set_layout(len(self._sky), self.num_params, slot_col, slot_const,
                       [SHIFT_METHODS[c.shift_method] for c in self._ps],
                       [int(bool(c.angle_degrees)) for c in self._sersic],
                       self.config.mag_zeropoint, family, p0, p1, p2)

    def device_group(self, devices, max_walkers=None):
        """A `engine.ContextGroup`: this model's field on several GPUs driven by this one
        process, layout and priors registered (psfmc_group_*).  Its `logpost_theta(theta)`
        splits the walkers over the devices.  The caller closes it."""
        sel = self.config.psf_selector
        grp = engine.ContextGroup(devices, self.config.obs_data, self.config.obs_var, self.config.bad_px,
                                  np.stack(sel.psf_data), np.stack(sel.psf_var), n_ps=len(self._ps),
                                  n_sersic=len(self._sersic), max_walkers=max_walkers or self._max_walkers,
                                  backend=self._backend)
        self._register_layout(grp)
        return grp

    def _host_prior_sum(self, theta):
        """log-prior of the priors the library leaves to the host, or None."""
        if not self._host_priors:
            return None
        total = np.zeros(theta.shape[0])
        with np.errstate(all='ignore'):
            for prior, cols in self._host_priors:
                total = total + prior.logp_batch(theta[:, cols])
        return total

    def close(self):
        if self._engine is not None:
            self._engine.close()
            self._engine = None

    # -- parameter vector -----------------------------------------------------
    @property
    def num_params(self):
        return self._num_params

    @property
    def param_names(self):
        return [n for c in self.components for n in c.stochastic_names()]

    @property
    def param_fits_abbrs(self):
        return [n for c in self.components
                for n in c.stochastic_names(name_attr='fitsname')]

    @property
    def param_lens(self):
        return [n for c in self.components for n in c.stochastic_lens()]

    @property
    def param_values(self):
        parts = np.split(self._param_vector, np.cumsum(self.param_lens)[:-1])
        return dict(zip(self.param_names, parts))

    @param_values.setter
    def param_values(self, vector):
        vector = np.asarray(vector, dtype=np.float64)
        self._param_vector = vector
        for comp, span in zip(self.components, self._spans):
            comp.set_stochastic_values(vector[span])

    def get_distribution(self, param_name):
        found = None
        for comp in self.components:
            try:
                found = comp.get_distribution(param_name)
            except KeyError:
                pass
        return found

    def init_params_from_priors(self, nwalkers):
        """Walker start positions drawn from the priors, re-drawing a component
        until its joint prior is finite (models.py:108-130)."""
        out = np.zeros((nwalkers, self.num_params))
        for w in range(nwalkers):
            for comp, span in zip(self.components, self._spans):
                while True:
                    vals = comp.set_stochastic_values('random')
                    if np.isfinite(comp.log_priors()):
                        break
                out[w, span] = vals
        return out

    # -- priors ---------------------------------------------------------------
    def log_priors(self):
        return np.sum([c.log_priors() for c in self.components])

    def log_priors_batch(self, theta):
        theta = self._theta(theta)
        total = np.zeros(theta.shape[0])
        with np.errstate(all='ignore'):
            for comp, span in zip(self.components, self._spans):
                total = total + comp.log_priors_batch(theta[:, span])
        return total

    # -- host -> device rows ---------------------------------------------------
    def _theta(self, theta):
        theta = np.asarray(theta, dtype=np.float64)
        if theta.ndim == 1:
            theta = theta[None, :]
        if theta.ndim != 2 or theta.shape[1] != self.num_params:
            raise ValueError('expected [W, {}] parameter vectors, got {}'
                             .format(self.num_params, theta.shape))
        return theta

    def derived_rows(self, theta):
        """[W, P] emcee vectors -> [W, row_len] rows of include/psfmc_hip.h."""
        theta = self._theta(theta)
        n_w = theta.shape[0]
        zp = self.config.mag_zeropoint
        vals = {id(c): c.values_batch(theta[:, s])
                for c, s in zip(self.components, self._spans)}
        cols = [sum((vals[id(c)]['adu'] for c in self._sky), np.zeros(n_w))]
        with np.errstate(all='ignore'):
            for c in self._ps:
                v = vals[id(c)]
                cols += [mag_to_flux(v['mag'], zp), v['xy'][:, 0], v['xy'][:, 1],
                         np.full(n_w, float(SHIFT_METHODS[c.shift_method]))]
            for c in self._sersic:
                v = vals[id(c)]
                theta_rot = (np.deg2rad(v['angle']) if c.angle_degrees
                             else v['angle']) + 0.5 * np.pi
                sin_t, cos_t = np.sin(theta_rot), np.cos(theta_rot)
                kappa = Sersic.kappa(v['index'])
                cols += [v['xy'][:, 0], v['xy'][:, 1],
                         cos_t / v['reff'], sin_t / v['reff'],
                         -sin_t / v['reff_b'], cos_t / v['reff_b'],
                         kappa, 0.5 / v['index'],
                         Sersic.sb_eff(mag_to_flux(v['mag'], zp), v['index'],
                                       v['reff'], v['reff_b'], kappa)]
        sel = vals[id(self.config.psf_selector)]
        psf = sel.get('psf_index', np.zeros(n_w))
        cols.append(np.clip(psf, 0, len(self.config.psf_selector.psf_data) - 1))
        return np.ascontiguousarray(np.stack([np.asarray(c, dtype=np.float64)
                                              for c in cols], axis=1))

    # -- the hot path -----------------------------------------------------------
    def log_likelihood_batch(self, theta, skip=None):
        """[W] Gaussian log-likelihoods from the GPU; non-finite -> -inf."""
        rows = self.derived_rows(theta)
        cap = self._max_walkers                  # larger batches go through in slices
        parts = [self.engine.loglike(rows[lo:lo + cap], None if skip is None else skip[lo:lo + cap])
                 for lo in range(0, len(rows), cap)]
        ll = np.concatenate(parts) if parts else np.zeros(0)
        return np.where(np.isfinite(ll), ll, -np.inf)

    def log_posterior_batch(self, theta):
        """log-posterior of W parameter vectors in one GPU batch: priors with the
        non-finite early-out (models.py:208-211), Sersic constants and likelihood
        all on the device; only priors of families the library does not know are
        evaluated here with scipy and passed along per walker."""
        theta = self._theta(theta)
        eng = self.engine
        extra = self._host_prior_sum(theta)
        cap = self._max_walkers
        parts = [eng.logpost_theta(theta[lo:lo + cap], None if extra is None else extra[lo:lo + cap])
                 for lo in range(0, theta.shape[0], cap)]
        return np.concatenate(parts) if parts else np.zeros(0)

    def log_posterior_batch_host(self, theta):
        """The same through host-side priors and derived rows (scipy.stats /
        scipy.special exactly as the reference calls them); the device only runs the
        likelihood.  Kept as the cross-check of the raw-vector path."""
        theta = self._theta(theta)
        lnprior = self.log_priors_batch(theta)
        skip = ~np.isfinite(lnprior)
        out = np.full(theta.shape[0], -np.inf)
        if not skip.all():
            safe = np.where(skip[:, None], theta[np.argmin(skip)], theta)
            ll = self.log_likelihood_batch(safe, skip)
            ok = ~skip
            out[ok] = ll[ok] + lnprior[ok]
        return out

    @staticmethod
    def log_posterior(param_values, **kwargs):
        """Single-vector form with the reference's signature (models.py:193-243):
        `kwargs` must hold `model`.  Returns (lnprob, blobs)."""
        model = kwargs.pop('model')
        model.param_values = param_values
        lnp = model.log_posterior_batch(param_values)[0]
        blobs = {}
        if model.blob_images and np.isfinite(model.log_priors()):
            blobs = {k: v[0] for k, v in model.sample_images(param_values).items()}
        return float(lnp), blobs

    # -- images ---------------------------------------------------------------
    def sample_images(self, theta, kinds=None):
        """The per-sample images of models.py:222-226 for W vectors:
        dict kind -> [W, ny, nx]."""
        rows = self.derived_rows(theta)
        cap = self._max_walkers
        parts = [self.engine.images(rows[lo:lo + cap], kinds) for lo in range(0, max(len(rows), 1), cap)]
        return {k: np.concatenate([p[k] for p in parts]) for k in parts[0]}

    def _one_image(self, kind):
        return self.sample_images(self._param_vector, (kind,))[kind][0]

    def raw_model(self):
        return self._one_image('raw_model')

    def convolved_model(self, raw_px=None):
        return self._one_image('convolved_model')

    def composite_ivm(self, raw_px=None):
        return self._one_image('composite_ivm')

    def residual(self, convolved_px=None, raw_px=None):
        return self._one_image('residual')

    def point_source_subtracted(self):
        return self._one_image('point_source_subtracted')

    def reset_images(self):
        shape = self.config.obs_data.shape
        self.accumulated_samples = 0
        self._device_samples = 0
        if self._engine is not None:
            self._engine.reset_accumulated()
        for kind in IMAGE_KINDS:
            self.posterior_images[kind] = np.ones(shape, dtype=np.float64)

    def accumulate_samples(self, theta):
        """Add the images of W parameter vectors to the posterior means ON THE
        DEVICE (no image leaves the GPU until `collect_posterior_images`)."""
        theta = self._theta(theta)
        eng = self.engine
        for lo in range(0, len(theta), self._max_walkers):
            # flux, kappa, Sigma_e, ellipse matrix on the device (host scipy cost 0.2 ms per sample)
            eng.accumulate_theta(theta[lo:lo + self._max_walkers])
        self._device_samples += len(theta)
        self.accumulated_samples += len(theta)

    def reduce_accumulated(self, ranks):
        """Walkers sharded over GPUs (`parallel.RankGroup`): add up the ranks' device-resident
        posterior sums, so that every rank holds the sums over ALL walkers.  One all-reduce of
        4 images at the end of sampling."""
        if ranks is None or ranks.single or self._engine is None:
            return
        sums, count = self._engine.accumulated_sums()
        total = ranks.all_reduce_sum_host(np.concatenate([sums.ravel(), [float(count)]]))
        n_all = int(round(total[-1]))
        self._engine.set_accumulated_sums(total[:-1].reshape(sums.shape), n_all)
        self._device_samples += n_all - count
        self.accumulated_samples += n_all - count

    def collect_posterior_images(self):
        """Merge the device-resident sums into `posterior_images` (sample-count
        weighted; the weight map in the variance domain) and return that dict."""
        if self._device_samples:
            dev, n_dev = self.engine.accumulated()
            n_host = self.accumulated_samples - self._device_samples
            post = self.posterior_images
            with np.errstate(all='ignore'):
                for kind in IMAGE_KINDS:
                    if n_host == 0:
                        post[kind] = dev[kind]
                    elif kind == 'composite_ivm':
                        post[kind] = (n_host + n_dev) / (n_host / post[kind] + n_dev / dev[kind])
                    else:
                        post[kind] = (n_host * post[kind] + n_dev * dev[kind]) / (n_host + n_dev)
            self.engine.reset_accumulated()
            self._device_samples = 0
            self._merged_device = True
        return self.posterior_images

    def accumulate_images(self, sample_images):
        """Running mean of the per-sample images; the weight map is averaged
        as a variance (models.py:74-97).  `sample_images`: list of dicts
        {kind: [ny, nx]} (emcee blobs) or one dict {kind: [W, ny, nx]}."""
        self.collect_posterior_images()
        if isinstance(sample_images, dict):
            n_w = len(next(iter(sample_images.values())))
            sample_images = [{k: v[i] for k, v in sample_images.items()}
                             for i in range(n_w)]
        post = self.posterior_images
        with np.errstate(all='ignore'):
            post['composite_ivm'] = 1 / post['composite_ivm']
            for imgs in sample_images:
                if not imgs:
                    continue
                self.accumulated_samples += 1
                n = self.accumulated_samples
                for kind, img in imgs.items():
                    step = 1 / img if kind == 'composite_ivm' else img
                    post[kind] = (post[kind] * (n - 1) + step) / n
            post['composite_ivm'] = 1 / post['composite_ivm']


class FieldSet(object):
    """Several fields of one image shape and one model structure (the same component lists, their own
    data, constants and priors) evaluated in shared GPU batches: `engine.FieldSetContext`.  The
    reference has no counterpart (one `MultiComponentModel` per model file and process,
    psfMC/fitting.py:13-113); this is for surveys of many small fields, where a batch per field would
    be dominated by its fixed cost (BASELINE config 5).

    models: MultiComponentModel objects (or model files) whose priors all have a device form."""

    def __init__(self, models, max_walkers=4096, device=0):
        # a model object handed in stays what it was (its own context, if it has one, included): the set works
        # on shallow copies that share the components and the data but route through the shared context
        self.models = [copy.copy(m) if isinstance(m, MultiComponentModel) else
                       MultiComponentModel(m, device=device, backend='fused', max_walkers=1) for m in models]
        first = self.models[0]
        for m in self.models:
            if (len(m._ps), len(m._sersic), len(m._sky), m.num_params) != \
                    (len(first._ps), len(first._sersic), len(first._sky), first.num_params):
                raise ValueError('the fields of a FieldSet need the same component lists and free parameters')
        fields = []
        for m in self.models:
            sel = m.config.psf_selector
            fields.append((m.config.obs_data, m.config.obs_var, m.config.bad_px, np.stack(sel.psf_data),
                           np.stack(sel.psf_var)))
        self.context = engine.FieldSetContext(fields, n_ps=len(first._ps), n_sersic=len(first._sersic),
                                              max_walkers=max_walkers, device=device)
        for f, m in enumerate(self.models):
            m._register_layout(self.context.layout_of(f))
            if m._host_priors:
                raise ValueError('field {}: a prior has no device form; a FieldSet evaluates priors on the '
                                 'GPU only'.format(f))
            # the model's images, posterior sums and log-posteriors go through ITS field of the shared
            # context (it never creates a context of its own)
            m._engine = self.context.view(f)
            m.posterior_images = dict(m.posterior_images)      # (a copy's own running means and vector)
            m._param_vector = m._param_vector.copy()
            m._max_walkers = int(max_walkers)
        self.num_params = first.num_params
        self.max_walkers = int(max_walkers)

    def log_posterior_batch(self, thetas):
        """thetas: one [W_f, num_params] array per field -> list of [W_f] log-posteriors."""
        return self.context.logpost_theta(thetas)

    def close(self):
        self.context.close()
        for m in self.models:
            m._engine = None               # (views of the context just closed)

