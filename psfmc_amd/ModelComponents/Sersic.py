import numpy as np
from scipy.special import gamma, gammaincinv

from .ComponentBase import ComponentBase, StochasticProperty


class Sersic(ComponentBase):
    """Elliptical Sersic profile (reference: ModelComponents/Sersic.py:16-45).
    `reff` / `reff_b` are the semi-major / semi-minor effective radii, `angle`
    the position angle (CCW of up; radians unless `angle_degrees`)."""
    device_kind = 'sersic'
    _fits_abbrs = [('Sersic', 'SER'), ('reff_b', 'REB'), ('reff', 'RE'),
                   ('index', 'N'), ('angle', 'ANG')]

    xy = StochasticProperty()
    mag = StochasticProperty()
    reff = StochasticProperty()
    reff_b = StochasticProperty()
    index = StochasticProperty()
    angle = StochasticProperty()

    def __init__(self, xy=None, mag=None, reff=None, reff_b=None, index=None,
                 angle=None, angle_degrees=False):
        super(Sersic, self).__init__()
        self.xy = xy
        self.mag = mag
        self.reff = reff
        self.reff_b = reff_b
        self.index = index
        self.angle = angle
        self.angle_degrees = angle_degrees

    # axis-ratio constraint: reff_b <= reff (Sersic.py:41-45)
    def log_priors(self):
        logp = super(Sersic, self).log_priors()
        return logp + (-np.inf if self.reff_b > self.reff else 0)

    def log_priors_batch(self, block):
        logp = super(Sersic, self).log_priors_batch(block)
        vals = self.values_batch(block)
        return np.where(vals['reff_b'] > vals['reff'], -np.inf, logp)

    @staticmethod
    def kappa(index):
        """b_n of Ciotti & Bertin (1999): gammaincinv(2n, 1/2) (Sersic.py:47-53)."""
        return gammaincinv(2 * index, 0.5)

    @staticmethod
    def sb_eff(flux_tot, index, reff, reff_b, kappa=None):
        """Surface brightness at the effective radius (Sersic.py:55-71)."""
        if kappa is None:
            kappa = Sersic.kappa(index)
        return flux_tot / (np.pi * reff * reff_b * 2 * index *
                           np.exp(kappa + np.log(kappa) * -2 * index) *
                           gamma(2 * index))
