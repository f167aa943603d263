from .ComponentBase import ComponentBase
from .PSFSelector import PSFSelector
from ..utils import preprocess_obs


class Configuration(ComponentBase):
    """Input images and control parameters of a model (reference:
    ModelComponents/Configuration.py:10-52).  Every *_file argument is a FITS
    file name (relative to the model file) or an array.

    obs_file      observed image, in the units of the zeropoint
    obsivm_file   its inverse-variance (weight) map
    psf_files     one PSF image or a list (several -> `psf_index` is sampled)
    psfivm_files  matching inverse-variance map(s)
    mask_file     optional FITS mask, nonzero = excluded from the fit
    mag_zeropoint magnitude of one ADU
    """

    def __init__(self, obs_file, obsivm_file, psf_files, psfivm_files,
                 mask_file=None, mag_zeropoint=0):
        super(Configuration, self).__init__()
        self.mag_zeropoint = mag_zeropoint
        hdr, data, var, bad = preprocess_obs(obs_file, obsivm_file, mask_file)
        if data.ndim != 2 or data.shape[0] % 2 or data.shape[1] % 2:
            raise ValueError('observation must be a 2-D image with even sides, '
                             'got shape {}'.format(data.shape))
        self.obs_header = hdr
        self.obs_data = data
        self.obs_var = var
        self.bad_px = bad
        self.psf_selector = PSFSelector(psf_files, psfivm_files, data.shape)
