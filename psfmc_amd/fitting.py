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
        # room for batches beyond the ensemble: the posterior images are recomputed from the filtered database in
        # slices of max_walkers samples (152 k samples in slices of 64 were 2300 calls, a fifth of a default fit)
        n_hint = max(chains or 0, 1024)
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


def model_fields_mcmc(model_files, output_names=None, write_fits=default_filetypes, iterations=0, burn=0,
                      chains=None, max_iterations=1, convergence_check=check_convergence_autocorr,
                      device=0, random_states=None, start_positions=None, accumulate=True, quiet=False):
    """`model_galaxy_mcmc` for SEVERAL fields of one image shape and one model structure at once, in
    one GPU context (`models.FieldSet`): every field is fitted exactly as its own `model_galaxy_mcmc`
    run would fit it -- its own ensemble of `chains` walkers, its own random stream, burn-in, sampling
    with posterior images accumulated, convergence check, trace database `<output_name>_db.fits` and
    posterior images -- but all fields advance together and every half-step's proposals of all fields
    are evaluated as ONE batch.  For surveys of many small fields (BASELINE config 5: 64 fields of
    256 x 256 with 256 walkers each, 8 per GPU), where a fit per field is dominated by fixed costs.
    The reference has no counterpart: one field per process (psfMC/fitting.py:13-113).

    model_files      model files (or MultiComponentModel objects) with the same component lists
    output_names     one per field (default 'out_<model file>')
    random_states    one RandomState state tuple or seed per field (default: numpy's global generator
                     seeds them in field order)
    start_positions  optional [F][chains, P] start positions (default: drawn from each field's priors)
    Returns a list of (model, database), one per field."""
    from .models import FieldSet
    from .sampler import FieldSetSampler
    n_f = len(model_files)
    if output_names is None:
        output_names = ['out_' + str(f).replace('.py', '') for f in model_files]
    if len(output_names) != n_f:
        raise ValueError('one output name per field')
    output_names = [o + '_{}' for o in output_names]
    first = model_files[0] if isinstance(model_files[0], MultiComponentModel) else None
    if chains is None:
        if first is None:
            first = MultiComponentModel(model_files[0], device=device, backend='fused', max_walkers=1)
            model_files = [first] + list(model_files[1:])
        chains = 2 * first.num_params + 2
    if chains < 2 or chains % 2:
        raise ValueError('chains must be an even number >= 2 (got {}): the stretch move updates the ensemble in '
                         'two halves'.format(chains))
    if random_states is not None and len(random_states) != n_f:
        raise ValueError('random_states: one per field ({} given for {} fields)'.format(len(random_states), n_f))
    if start_positions is not None and len(start_positions) != n_f:
        raise ValueError('start_positions: one [chains, P] array per field ({} given for {} fields)'.format(
            len(start_positions), n_f))
    fieldset = FieldSet(model_files, max_walkers=chains * n_f, device=device)
    models = fieldset.models
    if start_positions is not None:
        for f, p in enumerate(start_positions):
            if np.shape(p) != (chains, fieldset.num_params):
                raise ValueError('start_positions[{}] must be [{}, {}], got {}'.format(
                    f, chains, fieldset.num_params, np.shape(p)))
    sampler = FieldSetSampler(chains, fieldset, accumulate=False)
    for f, sub in enumerate(sampler.fields):
        state = None if random_states is None else random_states[f]
        if state is None:
            state = int(np.random.randint(0, 2 ** 31 - 1))
        if isinstance(state, (int, np.integer)):
            state = np.random.RandomState(int(state)).get_state()
        sub.random_state = state

    db_names = [o.format('db') + '.fits' for o in output_names]
    have = [os.path.exists(d) for d in db_names]
    if any(have) and not all(have):
        raise ValueError('some of the fields already have a database and some do not: {}'.format(
            [d for d, h in zip(db_names, have) if h]))
    if not all(have):
        if start_positions is None:
            start_positions = [m.init_params_from_priors(chains) for m in models]
        pos = np.array([np.asarray(p, dtype=np.float64) for p in start_positions])
        lnprob = None
        for step, result in enumerate(sampler.sample(pos, iterations=burn)):
            pos = np.array([r[0] for r in result])
            lnprob = np.array([r[1] for r in result])
            if not quiet:
                print_progress(step, burn, 'Burning')
        sampler.reset()
        sampler.accumulate = bool(accumulate)
        converged = [False] * n_f
        for sampling_iter in range(max_iterations):
            for step, result in enumerate(sampler.sample(pos, lnprob0=lnprob, iterations=iterations)):
                pos = np.array([r[0] for r in result])
                lnprob = np.array([r[1] for r in result])
                if not quiet:
                    print_progress(step, iterations, 'Sampling')
            converged = [bool(convergence_check(sub)) for sub in sampler.fields]
            if all(converged):
                break
            warn('Not yet converged after {:d} iterations: fields {}'.format(
                (sampling_iter + 1) * iterations, [f for f, ok in enumerate(converged) if not ok]))
        databases = []
        for f, sub in enumerate(sampler.fields):
            meta = OrderedDict([('MCITER', sub.chain.shape[1]), ('MCBURN', burn), ('MCCHAINS', chains),
                                ('MCCONVRG', converged[f]),
                                ('MCACCEPT', float(sub.acceptance_fraction.mean()))])
            databases.append(save_database(sub, models[f], db_names[f], meta_dict=meta))
    else:
        if not quiet:
            print('Databases already contain sampled chains, skipping sampling')
        databases = [load_database(d) for d in db_names]
    for m, db, out in zip(models, databases, output_names):
        save_posterior_images(m, db, output_name=out, filetypes=write_fits)
    return list(zip(models, databases))
