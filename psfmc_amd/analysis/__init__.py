"""Posterior analysis: convergence diagnostics and posterior model images
(reference: psfMC/analysis; plotting is out of scope)."""
from .statistics import check_convergence_autocorr, potential_scale_reduction, num_effective_samples
from .images import save_posterior_images, default_filetypes

__all__ = ['check_convergence_autocorr', 'potential_scale_reduction', 'num_effective_samples',
           'save_posterior_images', 'default_filetypes']
