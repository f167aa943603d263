import numpy as np

from .ComponentBase import ComponentBase, StochasticProperty
from ..distributions import DiscreteUniform
from ..utils import preprocess_psf, calculate_psf_variability


class PSFSelector(ComponentBase):
    """Holds the list of normalised PSFs and variance maps; with more than one
    PSF the index is a free `DiscreteUniform` parameter selecting the kernel
    per sample (reference: ModelComponents/PSFSelector.py:16-66).  Created by
    `Configuration`, always last in the component list.

    Unlike the reference the spectra are not computed here: the real-space
    arrays go to `psfmc_ctx_create`, which pads and transforms them on the GPU.
    """
    psf_index = StochasticProperty()

    def __init__(self, psf_list, ivm_list, data_shape):
        super(PSFSelector, self).__init__()
        if isinstance(psf_list, str) or isinstance(psf_list, np.ndarray):
            psf_list = [psf_list]
        if isinstance(ivm_list, str) or isinstance(ivm_list, np.ndarray):
            ivm_list = [ivm_list]
        if len(psf_list) != len(ivm_list):
            raise ValueError('PSF and IVM lists must be the same length')
        pairs = [preprocess_psf(p, v) for p, v in zip(psf_list, ivm_list)]
        shapes = {p.shape for p, _ in pairs}
        if len(shapes) != 1:
            raise ValueError('all PSFs must have the same shape: {}'.format(shapes))
        shape = shapes.pop()
        if shape[0] > data_shape[0] or shape[1] > data_shape[1]:
            raise NotImplementedError('PSF images larger than observation '
                                      'images are not yet supported')
        data, var = calculate_psf_variability([p for p, _ in pairs],
                                              [v for _, v in pairs])
        self.filenames = list(psf_list)
        self.psf_data = data          # real space, unit sum
        self.psf_var = var
        self.psf_index = (DiscreteUniform(low=0, high=len(data))
                          if len(data) > 1 else 0)

    def update_stochastic_names(self, count=None):
        if 'psf_index' in self._priors:
            self._priors['psf_index'].name = 'PSF_Index'
            self._priors['psf_index'].fitsname = 'PSF_IDX'

    @property
    def filename(self):
        return self.filenames[self.psf_index]
