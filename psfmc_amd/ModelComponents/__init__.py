"""
Model components: what a model file may contain.

`Configuration` (exactly one: the observed images, PSFs and zeropoint) and any
number of `Sky`, `PointSource` and `Sersic` components, each taking constants or
priors (`psfmc_amd.distributions`) as arguments.  The classes are parameter
containers; the images are rasterised on the GPU.
"""
from . import ComponentBase          # the module, so that `ComponentBase.ComponentBase` resolves
from .Sersic import Sersic
from .PointSource import PointSource
from .Sky import Sky
from .Configuration import Configuration

__all__ = ['Configuration', 'PointSource', 'Sersic', 'Sky']
