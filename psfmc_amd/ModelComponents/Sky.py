"""
Sky background component.

A spatially constant level, in the same units as the observed image (ADU per
pixel).  Being constant it passes through the PSF convolution unchanged except
for the PSF's normalisation, and it contributes `adu**2 * sum(psf variance)` to
the model variance; on the GPU it is simply the starting value of every
rasterised pixel (`raster_row` in csrc/psfmc_device.h).  Several Sky components
in one model add up.  Reference: psfMC/ModelComponents/Sky.py:14-16.
"""
from .ComponentBase import ComponentBase, StochasticProperty


class Sky(ComponentBase):
    #: understood by the device rasteriser as an additive constant
    device_kind = 'sky'

    #: level in ADU: a number, or a prior (e.g. ``Normal(loc=0, scale=0.01)``)
    adu = StochasticProperty()

    def __init__(self, adu=None):
        ComponentBase.__init__(self)
        if adu is None:
            raise ValueError('Sky needs a level `adu` (a value or a prior)')
        self.adu = adu

    def __repr__(self):
        return 'Sky(adu={!r})'.format(self._priors.get('adu', self._constants.get('adu')))
