"""BASELINE config 5 as a FIT: several fields of one shape sampled together in one context
(`FieldSetSampler` / `model_fields_mcmc`, psfmc_stretch_run_fields) -- every field's chain, trace
database and posterior images must equal what the field's own one-field run produces from the same
start positions and random state, bit for bit (images to summation order)."""
import os

import numpy as np
import pytest

import helpers
import synth_field

pytestmark = pytest.mark.gpu


def _models(n_fields, n_side=256, n_sersic=1, max_walkers=256):
    from test_gpu_fullsize import make_model
    return [make_model(n_side, n_sersic, 'fused', max_walkers=max_walkers, seed=s) for s in range(n_fields)]


def test_eight_fields_sampled_together_equal_their_own_runs():
    """8 fields x 256 walkers at 256^2 (config 5's per-GPU share): the joint device sampler against
    eight `DeviceEnsembleSampler`s, posterior images accumulated inside the sampling loop in both."""
    from psfmc_amd import FieldSet, FieldSetSampler, DeviceEnsembleSampler
    n_f, n_w, n_iter = 8, 256, 7
    own = _models(n_f)
    p0 = [synth_field.draw_walkers(256, 1, n_w, seed=90 + f, near_truth=fld['truth'])
          for f, (_, fld) in enumerate(own)]
    fs = FieldSet([m for m, _ in _models(n_f)], max_walkers=n_f * n_w)
    joint = FieldSetSampler(n_w, fs, block=3, accumulate=True)
    for f, sub in enumerate(joint.fields):
        sub.random_state = np.random.RandomState(500 + f).get_state()
    last = None
    for last in joint.sample(p0, iterations=n_iter):
        pass
    for f, (model, _) in enumerate(own):
        solo = DeviceEnsembleSampler(n_w, model, block=3, accumulate=True)
        solo.random_state = np.random.RandomState(500 + f).get_state()
        res = None
        for res in solo.sample(p0[f], iterations=n_iter):
            pass
        sub = joint.fields[f]
        assert np.array_equal(sub.chain, solo.chain), f
        assert np.array_equal(sub.lnprobability, solo.lnprobability), f
        assert np.array_equal(sub.naccepted, solo.naccepted), f
        assert np.array_equal(last[f][0], res[0]) and np.array_equal(last[f][1], res[1])
        assert last[f][2][2] == res[2][2] and np.array_equal(last[f][2][1], res[2][1])     # generator state
        assert solo.naccepted.sum() > 0
        want = model.collect_posterior_images()
        got = fs.models[f].collect_posterior_images()
        assert fs.models[f].accumulated_samples == model.accumulated_samples == n_w * n_iter
        for kind, img in want.items():
            fin = np.isfinite(img)
            assert np.array_equal(np.isfinite(got[kind]), fin), (f, kind)
            assert np.abs(got[kind][fin] - img[fin]).max() <= 1e-12 * np.abs(img[fin]).max(), (f, kind)
        model.close()
    # the fields' posterior sums are separate: clearing one leaves the others
    fs.models[0].accumulate_samples(p0[0][:5])
    fs.models[1].accumulate_samples(p0[1][:7])
    fs.models[0].reset_images()
    assert fs.context.accumulated(0)[1] == 0 and fs.context.accumulated(1)[1] == 7
    fs.close()


def test_field_models_images_go_through_the_shared_context():
    """A FieldSet's models serve per-sample images and recomputed posterior images from their field of
    the shared context (psfmc_eval_images_field, psfmc_accumulate_theta_field): equal to the same
    model on a context of its own."""
    from psfmc_amd import FieldSet
    own = _models(3, n_side=128, n_sersic=2, max_walkers=32)
    fs = FieldSet([m for m, _ in _models(3, n_side=128, n_sersic=2, max_walkers=32)], max_walkers=96)
    for f, (model, fld) in enumerate(own):
        theta = synth_field.draw_walkers(128, 2, 6, seed=11 + f, near_truth=fld['truth'])
        a = model.sample_images(theta)
        b = fs.models[f].sample_images(theta)
        for kind in a:
            assert np.array_equal(a[kind], b[kind]), (f, kind)
        model.accumulate_samples(theta)
        fs.models[f].accumulate_samples(theta)
    for f, (model, _) in enumerate(own):
        want, got = model.collect_posterior_images(), fs.models[f].collect_posterior_images()
        for kind in want:
            fin = np.isfinite(want[kind])
            assert np.abs(got[kind][fin] - want[kind][fin]).max() <= 1e-12 * np.abs(want[kind][fin]).max()
        model.close()
    fs.close()


def test_model_fields_mcmc_writes_each_fields_own_outputs(tmp_path):
    """The entry point: three fields fitted together write the databases and posterior images their own
    `model_galaxy_mcmc` runs write (same start positions, same random states)."""
    from psfmc_amd import model_fields_mcmc, model_galaxy_mcmc, fits_io
    n_f, chains = 3, 24
    own = _models(n_f, n_side=128, n_sersic=1, max_walkers=chains)
    p0 = [synth_field.draw_walkers(128, 1, chains, seed=40 + f, near_truth=fld['truth'])
          for f, (_, fld) in enumerate(own)]
    joint_models = [m for m, _ in _models(n_f, n_side=128, n_sersic=1, max_walkers=chains)]
    with pytest.warns(UserWarning):                     # 6 iterations do not converge
        results = model_fields_mcmc(joint_models, output_names=[str(tmp_path / ('j%d' % f)) for f in range(n_f)],
                                    iterations=6, burn=3, chains=chains, random_states=[7 + f for f in range(n_f)],
                                    start_positions=p0, quiet=True,
                                    write_fits=('convolved_model', 'composite_ivm'))
    assert len(results) == n_f
    for f, (model, _) in enumerate(own):
        # the one-field run from the same start: model_galaxy_mcmc draws its start from numpy's global
        # generator, so the prior draw is replaced by the given positions
        model.init_params_from_priors = lambda n, p=p0[f]: p
        with pytest.warns(UserWarning):
            _, db = model_galaxy_mcmc(model, output_name=str(tmp_path / ('s%d' % f)), iterations=6, burn=3,
                                      chains=chains, random_state=7 + f, quiet=True, group=None,
                                      write_fits=('convolved_model', 'composite_ivm'))
        jdb = results[f][1]
        assert jdb.colnames == db.colnames
        for name in db.colnames:
            assert np.array_equal(np.asarray(jdb[name]), np.asarray(db[name])), (f, name)
        for kind in ('convolved_model', 'composite_ivm'):
            a = fits_io.read_image(str(tmp_path / ('j%d_%s.fits' % (f, kind))))
            b = fits_io.read_image(str(tmp_path / ('s%d_%s.fits' % (f, kind))))
            assert np.abs(a - b).max() <= 1e-12 * np.abs(b).max(), (f, kind)
        model.close()
    for m, _ in results:
        m.close()


def test_field_set_leaves_the_callers_models_alone_and_serves_likelihood_calls():
    """Round-3 advice: a model object handed to a `FieldSet` keeps its own context and every call it had; the
    set's own copies answer the likelihood-level calls through their field of the shared context
    (psfmc_eval_batch_field) and refuse -- by name -- what a shared context cannot do per field."""
    from psfmc_amd import FieldSet, DeviceEnsembleSampler, engine
    from psfmc_amd.fitting import model_fields_mcmc
    n_f, n_w = 3, 64
    own = _models(n_f, max_walkers=n_w)
    thetas = [synth_field.draw_walkers(256, 1, n_w, seed=20 + f, near_truth=fld['truth'])
              for f, (_, fld) in enumerate(own)]
    thetas[1][:5] = synth_field.draw_walkers(256, 1, 5, seed=3)          # some prior draws, some of them -inf
    before = [m.log_posterior_batch(t) for (m, _), t in zip(own, thetas)]   # the callers' models own contexts now
    fs = FieldSet([m for m, _ in own], max_walkers=n_f * n_w)
    for f, (model, _) in enumerate(own):
        assert fs.models[f] is not model and isinstance(model._engine, engine.Context)
        # the caller's model: same results as before, every call still there
        assert np.array_equal(model.log_posterior_batch(thetas[f]), before[f])
        ll_own = model.log_likelihood_batch(thetas[f])
        host_own = model.log_posterior_batch_host(thetas[f])
        # the set's copy of it: the same numbers through the shared context
        view = fs.models[f]
        assert np.array_equal(view.log_posterior_batch(thetas[f]), before[f])
        assert np.array_equal(view.log_likelihood_batch(thetas[f]), ll_own)
        assert np.array_equal(view.log_posterior_batch_host(thetas[f]), host_own)
        with pytest.raises(NotImplementedError, match='FieldSet'):
            DeviceEnsembleSampler(n_w, view)
        with pytest.raises(NotImplementedError, match='accumulated_sums'):
            view.engine.accumulated_sums()
        solo = DeviceEnsembleSampler(n_w, model, block=2)                 # ... and the caller's still samples
        for _ in solo.sample(thetas[f], iterations=2):
            pass
    with pytest.raises(ValueError, match='even'):
        model_fields_mcmc([m for m, _ in own], chains=7, iterations=1, quiet=True)
    with pytest.raises(ValueError, match='one per field'):
        model_fields_mcmc([m for m, _ in own], chains=8, iterations=1, random_states=[1, 2], quiet=True)
    with pytest.raises(ValueError, match='start_positions'):
        model_fields_mcmc([m for m, _ in own], chains=8, iterations=1, quiet=True,
                          start_positions=[np.zeros((6, 10))] * n_f)
    fs.close()
    for model, _ in own:
        model.close()
