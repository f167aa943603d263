"""
psfmc_amd -- MI355X-native batched log-posterior for psfMC-style MCMC surface
brightness modelling.  Public names mirror the reference package `psfMC`.
"""
from .models import MultiComponentModel
from .batch import BatchLogPosterior

__version__ = '0.1.0'
__all__ = ['MultiComponentModel', 'BatchLogPosterior']
