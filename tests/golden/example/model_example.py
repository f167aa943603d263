# Model definition for the J0005-0006 example field (HST data files shipped with
# the reference under examples/): sky + quasar point source + host galaxy +
# a faint companion.  Same components and priors as the reference's example
# model; written for this repo's tests (no ds9 region mask: pyregion is absent,
# so the reference ignores that mask as well -- SURVEY.md section 8(a) note C).
from numpy import array
from psfMC.ModelComponents import Configuration, Sky, PointSource, Sersic
from psfMC.distributions import Normal, Uniform, WeibullMinimum

Configuration(obs_file='sci_J0005-0006.fits', obsivm_file='ivm_J0005-0006.fits',
              psf_files='sci_psf.fits', psfivm_files='ivm_psf.fits',
              mag_zeropoint=25.9463)

Sky(adu=Normal(loc=0, scale=0.01))

qso_mag = 20.66
qso_xy = array((64.5, 64.5))
PointSource(xy=Uniform(loc=qso_xy - 8, scale=16 * array((1, 1))),
            mag=Uniform(loc=qso_mag - 0.2, scale=1.7))

Sersic(xy=Uniform(loc=qso_xy - 8, scale=16 * array((1, 1))),
       mag=Uniform(loc=qso_mag, scale=27.5 - qso_mag),
       reff=Uniform(loc=2.0, scale=10.0),
       reff_b=Uniform(loc=2.0, scale=10.0),
       index=WeibullMinimum(c=1.5, scale=4),
       angle=Uniform(loc=0, scale=180), angle_degrees=True)

blob_xy = array((46, 85.6))
Sersic(xy=Uniform(loc=blob_xy - 5, scale=10 * array((1, 1))),
       mag=Uniform(loc=23.5, scale=2.0),
       reff=Uniform(loc=2.0, scale=6.0),
       reff_b=Uniform(loc=2.0, scale=6.0),
       index=WeibullMinimum(c=1.5, scale=4),
       angle=Uniform(loc=0, scale=180), angle_degrees=True)
