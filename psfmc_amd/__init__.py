"""
psfmc_amd -- MI355X-native batched log-posterior for psfMC-style MCMC surface
brightness modelling.  Public names mirror the reference package `psfMC`.
"""
from .models import MultiComponentModel, FieldSet
from .batch import BatchLogPosterior
from .sampler import EnsembleSampler, DeviceEnsembleSampler, FieldSetSampler
from .parallel import RankGroup, ShardedLogPosterior
from .fitting import model_galaxy_mcmc, model_fields_mcmc
from .database import load_database

__version__ = '0.1.0'
__all__ = ['MultiComponentModel', 'FieldSet', 'BatchLogPosterior', 'EnsembleSampler', 'DeviceEnsembleSampler',
           'FieldSetSampler', 'RankGroup', 'ShardedLogPosterior', 'model_galaxy_mcmc', 'model_fields_mcmc',
           'load_database']
