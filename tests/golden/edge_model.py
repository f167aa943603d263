from numpy import array
Configuration(obs_file='sci.fits', obsivm_file='ivm.fits',
              psf_files=['psf0.fits', 'psf1.fits'],
              psfivm_files=['psfivm0.fits', 'psfivm1.fits'],
              mask_file='mask.fits', mag_zeropoint=24.0)
Sky(adu=Normal(loc=0, scale=0.05))
PointSource(xy=Uniform(loc=array((-2.0, -2.0)), scale=array((132.0, 68.0))),
            mag=Uniform(loc=17.0, scale=5.0), shift_method='bilinear')
PointSource(xy=Uniform(loc=array((-2.0, -2.0)), scale=array((132.0, 68.0))),
            mag=Uniform(loc=17.0, scale=5.0))
Sersic(xy=Uniform(loc=array((40.0, 10.0)), scale=array((50.0, 40.0))),
       mag=Uniform(loc=16.0, scale=6.0), reff=Uniform(loc=1.0, scale=20.0),
       reff_b=Uniform(loc=1.0, scale=20.0), index=Uniform(loc=0.4, scale=7.6),
       angle=Uniform(loc=-4, scale=8))
