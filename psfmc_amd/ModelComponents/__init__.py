from . import ComponentBase as ComponentBase     # module, like the reference
from .Configuration import Configuration
from .PointSource import PointSource
from .Sersic import Sersic
from .Sky import Sky

__all__ = ['Configuration', 'PointSource', 'Sersic', 'Sky']
