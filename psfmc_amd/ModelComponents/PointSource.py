from .ComponentBase import ComponentBase, StochasticProperty

SHIFT_METHODS = {'lanczos3': 0, 'bilinear': 1}     # include/psfmc_hip.h codes


class PointSource(ComponentBase):
    """Sub-pixel point source; `xy` is 0-based like numpy indices
    (reference: ModelComponents/PointSource.py:6-22).  The flux is spread with
    a Lanczos-3 (default) or bilinear kernel by the GPU rasteriser."""
    device_kind = 'ps'
    _fits_abbrs = [('PointSource', 'PS')]

    xy = StochasticProperty()
    mag = StochasticProperty()

    def __init__(self, xy=None, mag=None, shift_method='lanczos3'):
        super(PointSource, self).__init__()
        if shift_method not in SHIFT_METHODS:
            raise ValueError('Unknown shift method: {}'.format(shift_method))
        self.xy = xy
        self.mag = mag
        self.shift_method = shift_method
