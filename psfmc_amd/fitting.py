"""
`model_galaxy_mcmc`: the reference's entry point (psfMC/fitting.py:13-113) on the
batched GPU log-posterior.  Same arguments and outputs (trace database
`<output_name>_db.fits`, posterior images `<output_name>_<type>.fits`); every
half-ensemble proposal is one GPU batch, and the per-iteration image accumulation
(`fitting.py:83`) evaluates the walkers' current positions in one batch instead of
carrying five images back from every proposal.
"""
import os
from collections import OrderedDict
from warnings import warn

import numpy as np

from .analysis import check_convergence_autocorr, save_posterior_images, default_filetypes
from .database import save_database, load_database
from .models import MultiComponentModel
from .sampler import EnsembleSampler, DeviceEnsembleSampler
from .utils import print_progress


def _rank_group(group, device):
    """The `parallel.RankGroup` of this run, or None for a single process.  'auto' looks for an
    initialised torch.distributed process group and quietly settles for one process when torch
    (or its distributed package) is not importable: the single-GPU entry point needs neither."""
    if group is None:
        return None
    auto = isinstance(group, str)
    try:
        from .parallel import RankGroup
        if isinstance(group, RankGroup):
            ranks = group
        else:
            if auto:
                import torch.distributed as dist
                if not (dist.is_available() and dist.is_initialized()):
                    return None
            ranks = RankGroup(None if auto else group, 'cuda:%d' % device)
    except ImportError:
        if auto:
            return None
        raise
    return None if ranks.single else ranks


def model_galaxy_mcmc(model_file, output_name=None, write_fits=default_filetypes, iterations=0,
                      burn=0, chains=None, max_iterations=1,
                      convergence_check=check_convergence_autocorr, sampler_class=None,
                      device=0, backend='auto', random_state=None, accumulate=True, quiet=False,
                      group='auto'):
    """Model a galaxy's surface brightness with MCMC.

    model_file, output_name, write_fits, iterations, burn, chains, max_iterations,
    convergence_check: as in the reference.  Extra keywords: `sampler_class` (an
    emcee-compatible EnsembleSampler; default the built-in one), `device`,
    `backend`, `random_state` (RandomState state tuple or seed for reproducible
    runs), `accumulate` (posterior images during sampling), `group`: with an initialised
    torch.distributed process group ('auto': the default group if there is one; or a group
    object) the walkers of every half-step are sharded over the ranks, one process per GPU
    (`device` should then be the rank's LOCAL_RANK); rank 0 draws the start positions and the
    sampler's random state and writes the outputs, every rank returns the same database."""
    if output_name is None:
        output_name = 'out_' + model_file.replace('.py', '')
    output_name += '_{}'

    mc_model = model_file if isinstance(model_file, MultiComponentModel) else None
    if mc_model is None:
        n_hint = max(chains or 0, 64)
        mc_model = MultiComponentModel(model_file, device=device, backend=backend,
                                       max_walkers=n_hint)
    ranks = _rank_group(group, mc_model._device)
    if chains is None:
        chains = 2 * mc_model.num_params + 2
    if chains > mc_model._max_walkers:
        raise ValueError('model was built for at most {} walkers'.format(mc_model._max_walkers))

    cls = sampler_class
    if cls is None:           # walkers resident on the GPU whenever every prior can be
        mc_model.engine
        cls = EnsembleSampler if mc_model._host_priors else DeviceEnsembleSampler
    device_acc = False
    if cls is DeviceEnsembleSampler:
        sampler = cls(chains, mc_model, group=ranks)
        device_acc = accumulate
    elif cls is EnsembleSampler:
        if ranks:
            from .parallel import ShardedLogPosterior
            evaluate = ShardedLogPosterior(mc_model, group=ranks)
        else:
            evaluate = mc_model.log_posterior_batch
        sampler = cls(chains, mc_model.num_params, batch_lnpostfn=evaluate)
    else:                      # a real emcee: batch through its pool hook
        from .batch import BatchLogPosterior
        sampler = cls(chains, mc_model.num_params, mc_model.log_posterior,
                      kwargs={'model': mc_model}, pool=BatchLogPosterior(mc_model).as_pool())
    if random_state is not None:
        if isinstance(random_state, (int, np.integer)):
            random_state = np.random.RandomState(int(random_state)).get_state()
        sampler.random_state = random_state
    writer = ranks is None or ranks.rank == 0
    if ranks is not None:                      # one random stream for all ranks
        sampler.random_state = ranks.broadcast_object(sampler.random_state)
        quiet = quiet or not writer

    db_name = output_name.format('db') + '.fits'
    have_db = os.path.exists(db_name)
    if ranks is not None:
        have_db = ranks.broadcast_object(have_db)
    if not have_db:
        param_vec = mc_model.init_params_from_priors(chains)
        if ranks is not None:
            param_vec = ranks.broadcast_object(param_vec)
        lnprob = None
        for step, result in enumerate(sampler.sample(param_vec, iterations=burn)):
            param_vec, lnprob = result[0], result[1]
            sampler.clear_blobs()
            if not quiet:
                print_progress(step, burn, 'Burning')
        sampler.reset()

        converged = False
        if device_acc:              # images are summed inside the device sampling loop
            sampler.accumulate = True
        for sampling_iter in range(max_iterations):
            for step, result in enumerate(sampler.sample(param_vec, lnprob0=lnprob,
                                                         iterations=iterations)):
                param_vec, lnprob = result[0], result[1]
                if accumulate and not device_acc and writer:   # current positions, summed on the GPU
                    mc_model.accumulate_samples(param_vec)
                sampler.clear_blobs()
                if not quiet:
                    print_progress(step, iterations, 'Sampling')
            if convergence_check(sampler):
                converged = True
                break
            warn('Not yet converged after {:d} iterations:'.format((sampling_iter + 1) * iterations))
            convergence_check(sampler, verbose=0 if quiet else 1)

        if device_acc:
            mc_model.reduce_accumulated(ranks)
        meta = OrderedDict([('MCITER', sampler.chain.shape[1]), ('MCBURN', burn),
                            ('MCCHAINS', chains), ('MCCONVRG', bool(converged)),
                            ('MCACCEPT', float(sampler.acceptance_fraction.mean()))])
        database = save_database(sampler, mc_model, db_name, meta_dict=meta) if writer else None
        if ranks is not None:
            ranks.broadcast_object(True)          # the file is complete
            database = database if writer else load_database(db_name)
    else:
        if writer:
            print('Database already contains sampled chains, skipping sampling')
        database = load_database(db_name)

    if writer:
        save_posterior_images(mc_model, database, output_name=output_name, filetypes=write_fits)
    if ranks is not None:
        ranks.broadcast_object(True)              # outputs written
    return mc_model, database
