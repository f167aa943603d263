"""
Component plugin base -- host glue (parameter containers; rasterisation itself
happens on the GPU).

Keeps the reference's contract (psfMC/ModelComponents/ComponentBase.py):
  * a constructor argument is either a constant or a prior (anything with a
    `.value`; :26-34);
  * free parameters are packed in alphabetical attribute order, each taking
    `size(value)` slots (`xy` -> 2; :45-74, :82-89) -- the emcee vector is the
    concatenation over components in model-file order (models.py:174-185);
  * trace names `<idx>_<Class>_<attr>` and abbreviated FITS names (:99-119);
  * the component log-prior is the sum of its priors' log-probabilities (:121-129).
On top of that every operation has a batch form working on walker columns
`[W, n]`, which is what `BatchLogPosterior` uses.
"""
import numpy as np


class StochasticProperty(object):
    """Descriptor declaring an attribute that may hold a prior or a constant
    (reference: ComponentBase.py:132-153).  `xy = StochasticProperty()` or
    `StochasticProperty('xy')`."""

    def __init__(self, key=None):
        self.key = key

    def __set_name__(self, owner, name):
        if self.key is None:
            self.key = name
        declared = list(getattr(owner, '_declared', ()))
        if self.key not in declared:
            declared.append(self.key)
        owner._declared = tuple(declared)

    def __get__(self, obj, owner=None):
        if obj is None:
            return self
        return obj.get_stochastic_val(self.key)

    def __set__(self, obj, value):
        obj.assign_stochastic(self.key, value)

    def __delete__(self, obj):
        raise NotImplementedError('Cannot delete stochastics')


class ComponentBase(object):
    _fits_abbrs = []
    _declared = ()
    #: rasteriser kind understood by the GPU path ('sky', 'ps', 'sersic') or None
    device_kind = None

    def __init__(self):
        self._priors = {}
        self._constants = {}

    # -- storage ----------------------------------------------------------
    def assign_stochastic(self, name, value):
        if hasattr(value, 'value'):
            self._constants.pop(name, None)
            self._priors[name] = value
        else:
            self._priors.pop(name, None)
            self._constants[name] = value

    def get_stochastic_val(self, name):
        if name in self._priors:
            return self._priors[name].value
        return self._constants[name]

    def get_distribution(self, stoch_name):
        hits = [p for p in self._priors.values() if p.name == stoch_name]
        if len(hits) != 1:
            raise KeyError('Could not find unique prior with name: {}'
                           .format(stoch_name))
        return hits[0]

    # -- packing contract --------------------------------------------------
    def free_names(self):
        return sorted(self._priors)

    def stochastic_lens(self):
        return [self._priors[k].size for k in self.free_names()]

    def num_stochastics(self):
        return int(sum(self.stochastic_lens()))

    def stochastic_names(self, name_attr='name'):
        return [getattr(self._priors[k], name_attr) for k in self.free_names()]

    def update_stochastic_names(self, count=None):
        kind = type(self).__name__
        for attr, prior in self._priors.items():
            long_name = '{}_{}'.format(kind, attr)
            short = long_name
            for word, abbr in type(self)._fits_abbrs:
                short = short.replace(word, abbr)
            if count is not None:
                long_name = '{:d}_{}'.format(count, long_name)
                short = '{:d}{}'.format(count, short)
            prior.name = long_name
            prior.fitsname = short

    def set_stochastic_values(self, param_values='random'):
        """Vector of this component's free values, or 'random' / 'median' to
        take them from the priors.  Returns the vector that was set."""
        names = self.free_names()
        if isinstance(param_values, str):
            how = param_values
            parts = [np.ravel(getattr(self._priors[k], how)()) for k in names]
            param_values = (np.concatenate(parts) if parts
                            else np.zeros(0))
        pos = 0
        for k, width in zip(names, self.stochastic_lens()):
            self._priors[k].value = np.array(param_values[pos:pos + width])
            pos += width
        return param_values

    def log_priors(self):
        total = 0
        for prior in self._priors.values():
            total += np.sum(prior.logp(prior.value))
        return total

    # -- batch forms --------------------------------------------------------
    def _columns(self):
        pos, cols = 0, {}
        for k, width in zip(self.free_names(), self.stochastic_lens()):
            cols[k] = slice(pos, pos + width)
            pos += width
        return cols

    def values_batch(self, block):
        """[W, num_stochastics] walker columns -> {attr: [W] or [W, k]} for
        every declared attribute, constants broadcast."""
        block = np.asarray(block, dtype=np.float64)
        n_w = block.shape[0]
        cols = self._columns()
        out = {}
        for k in self._declared:
            if k in self._priors:
                out[k] = self._priors[k].coerce(block[:, cols[k]])
            elif k in self._constants:
                const = np.asarray(self._constants[k], dtype=np.float64)
                out[k] = np.broadcast_to(const, (n_w,) + const.shape)
        return out

    def log_priors_batch(self, block):
        """[W, num_stochastics] -> [W] joint log-prior of this component."""
        block = np.asarray(block, dtype=np.float64)
        total = np.zeros(block.shape[0])
        for k, cols in self._columns().items():
            total = total + self._priors[k].logp_batch(block[:, cols])
        return total
