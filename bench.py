#!/usr/bin/env python
"""
bench.py -- walker log-posterior evaluations per second (BASELINE.json metric).

A *step* is one pass of the hot path over one batch of synthetic walkers: the
256x256 field with 1 PointSource + 1 Sersic that BASELINE.json's `metric` is quoted on
(BASELINE.md / SURVEY.md section 8(d) headline), W walkers per GPU, derived-parameter
rows already resident in HBM; the step ends with the all-gather of the log-likelihoods
across ranks (N > 1).  BASELINE.json configs[1] (the reference's J0005-0006 example
model, 256 walkers in one batch) is a parity case (tests/golden/example); its throughput
is reported next to the headline as `example_model_256_walkers` (N = 1 only).  One process per GPU (`python -m torch.distributed.run`
for N > 1); weak scaling (per-GPU batch fixed).

Prints ONE JSON line on rank 0 (contract in the task description), including
  roofline      algorithmic bytes (96 N^2 + 64 N per evaluation)
                over the live-measured device time of the evaluation pipeline
  cpu_baseline  the numpy oracle (a port of the reference algorithm) timed on
                this host, one walker per call like the reference
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'tools')):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBPS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md, chip-level table


def algorithmic_bytes_per_eval(n):
    return 96 * n * n + 64 * n


def pmc_traffic(args, kernel):
    """HBM bytes per walker of `kernel` from the committed rocprofv3 PMC passes
    (profiles/pmc_traffic.json; FETCH_SIZE doubled per MI355X_MICROARCH.md section HBM,
    plus WRITE_SIZE), or None when that shape was not profiled."""
    path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    try:
        with open(path) as f:
            table = json.load(f)
        return table['%dx%d' % (args.size, args.size)][kernel]['hbm_bytes_per_walker']
    except (IOError, OSError, KeyError, ValueError):
        return None


def build_problem(args, device, seed=0):
    """Synthetic field + walker rows.  Returns (model, theta[W,P])."""
    import tempfile
    import synth_field
    from psfmc_amd import MultiComponentModel, fits_io
    fld = synth_field.make_field(args.size, args.sersic, seed=seed)
    tmp = tempfile.mkdtemp(prefix='psfmc_bench_')
    for key, name in (('sci', 'sci.fits'), ('ivm', 'ivm.fits'), ('psf', 'psf.fits'),
                      ('psf_ivm', 'psf_ivm.fits')):
        fits_io.write_image(os.path.join(tmp, name), fld[key])
    path = os.path.join(tmp, 'model.py')
    with open(path, 'w') as f:
        f.write(synth_field.model_file_text(args.size, args.sersic))
    model = MultiComponentModel(path, device=device, backend=args.backend,
                                max_walkers=args.walkers)
    half = args.walkers // 2
    theta = np.vstack([
        synth_field.draw_walkers(args.size, args.sersic, half, seed=1),
        synth_field.draw_walkers(args.size, args.sersic, args.walkers - half, seed=2,
                                 near_truth=fld['truth'])])
    return model, theta, fld


def kernel_profile(eng, args, step, torch, dev, steps):
    """Per-kernel device time from HIP events recorded inside the library around
    every launch (set_option 'profile').  Run with ONE pass in flight so that a
    kernel's events bracket that kernel alone -- the same quantity rocprofv3
    --kernel-trace reports.  Returns a list of dicts, longest total first."""
    n = args.size
    nxh = n // 2 + 1
    t_bytes = 2 * nxh * n * 16                  # transposed half-spectra of one walker
    designed = {'rows_fwd': t_bytes, 'cols': 2 * t_bytes, 'rows_inv': t_bytes}
    names = {'rows_fwd': 'k_rows_fwd<%d, false>' % n, 'cols': 'k_cols<%d, true>' % n,
             'rows_inv': 'k_rows_inv<%d>' % n}
    streams = eng.get_option('streams')
    eng.set_option('streams', 1)
    eng.set_option('profile', 1)
    for _ in range(steps):
        step()
    torch.cuda.synchronize(dev)
    chunk = int(eng.get_option('chunk_walkers'))
    out = []
    for key in ('rows_fwd', 'cols', 'rows_inv'):
        ms = eng.get_option('prof_ms_' + key)
        cnt = eng.get_option('prof_n_' + key)
        if not cnt:
            continue
        avg_s = ms * 1e-3 / cnt
        walkers = args.walkers * steps / cnt          # walkers one launch processes (avg)
        out.append({'kernel': names[key], 'launches': int(cnt), 'avg_ms': avg_s * 1e3,
                    'total_ms': ms, 'walkers_per_launch': walkers,
                    'bytes_per_walker': designed[key],
                    'GBps': designed[key] * walkers / avg_s / 1e9})
    eng.set_option('profile', 0)
    eng.set_option('streams', streams)
    out.sort(key=lambda d: -d['total_ms'])
    return out, chunk


def example_model_rate(walkers=256, reps=40):
    """BASELINE.json configs[1]: the reference's example field (128^2 HST data, Sky +
    PointSource + 2 Sersic, 18 parameters; data files under tests/golden/example), 256
    walkers drawn from the priors, one batch per call through the Python entry point
    (`MultiComponentModel.log_posterior_batch`: host vectors in, log-posteriors out)."""
    mfile = os.path.join(ROOT, 'tests', 'golden', 'example', 'model_example.py')
    if not os.path.exists(mfile):
        return None
    from psfmc_amd import MultiComponentModel
    m = MultiComponentModel(mfile, max_walkers=walkers)
    np.random.seed(7)
    theta = m.init_params_from_priors(walkers)
    out = m.log_posterior_batch(theta)                     # warm-up (context, layout)
    t0 = time.perf_counter()
    for _ in range(reps):
        out = m.log_posterior_batch(theta)
    dt = time.perf_counter() - t0
    m.close()
    return {'evals_per_s': walkers * reps / dt, 'ms_per_batch': dt / reps * 1e3, 'walkers': walkers,
            'finite': int(np.isfinite(out).sum()),
            'workload': 'J0005-0006 example model (128x128, Sky + PointSource + 2 Sersic), 256 walkers '
                        'per batch, host vectors in / log-posteriors out, fp64'}


def cpu_baseline(args, fld, theta, budget_s):
    """The oracle, one walker per call (the reference's execution model,
    psfMC/fitting.py:55), on a bounded sample of the same walkers."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import psfmc_oracle as orc
    import helpers
    field = orc.make_field(fld['sci'], fld['ivm'], [fld['psf']], [fld['psf_ivm']],
                           mag_zp=fld['mag_zp'])
    layout = helpers.synth_layout(args.sersic)
    done, t0 = 0, time.perf_counter()
    vals = []
    while True:
        vals.append(helpers.oracle_loglike(field, layout, theta[done % len(theta)]))
        done += 1
        el = time.perf_counter() - t0
        if el >= budget_s:
            break
    return {'value': done / el, 'unit': 'evals/s', 'cores': 1, 'kind': 'port',
            'sample': '%d walkers of the same batch, one per call, %.1f s' % (done, el)}, vals


def cpu_worker(args):
    """Child process of cpu_baseline_multi: evaluate walkers with the oracle for
    --cpu-seconds and print how many were done (no GPU, no torch)."""
    import synth_field
    fld = synth_field.make_field(args.size, args.sersic, seed=0)
    half = args.walkers // 2
    theta = np.vstack([
        synth_field.draw_walkers(args.size, args.sersic, half, seed=1),
        synth_field.draw_walkers(args.size, args.sersic, args.walkers - half, seed=2,
                                 near_truth=fld['truth'])])
    _, vals = cpu_baseline(args, fld, theta[args.cpu_worker::7], args.cpu_seconds)
    print('CPU_WORKER_DONE %d' % len(vals))


def cpu_baseline_multi(args, n_procs):
    """The best the reference could do had `threads=n` worked: one
    single-threaded process per core, walkers split between them."""
    import subprocess
    env = dict(os.environ, OMP_NUM_THREADS='1', MKL_NUM_THREADS='1', OPENBLAS_NUM_THREADS='1')
    cmd = [sys.executable, os.path.abspath(__file__), '--size', str(args.size), '--sersic',
           str(args.sersic), '--walkers', str(args.walkers), '--cpu-seconds', str(args.cpu_seconds)]
    t0 = time.perf_counter()
    procs = [subprocess.Popen(cmd + ['--cpu-worker', str(i)], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.DEVNULL, text=True) for i in range(n_procs)]
    done = 0
    for p in procs:
        out, _ = p.communicate(timeout=args.cpu_seconds * 4 + 120)
        for line in out.splitlines():
            if line.startswith('CPU_WORKER_DONE'):
                done += int(line.split()[1])
    el = time.perf_counter() - t0
    return {'value': done / args.cpu_seconds, 'unit': 'evals/s', 'cores': n_procs, 'kind': 'port',
            'sample': '%d single-threaded processes x %.0f s of walkers of the same batch '
                      '(wall %.1f s incl. start-up)' % (n_procs, args.cpu_seconds, el)}


def many_fields(args, torch, dist, world, rank, local, dev):
    """BASELINE config 5: every rank owns --fields independent fields (own context,
    own streams, own walkers); a step evaluates all of them.  Fields shard across
    ranks with no data-path collective."""
    probs = [build_problem(args, local, seed=rank * args.fields + f) for f in range(args.fields)]
    engs, rows, outs = [], [], []
    for model, theta, _ in probs:
        engs.append(model.engine)
        rows.append(torch.from_numpy(model.derived_rows(theta)).to(dev))
        outs.append(torch.empty(args.walkers, dtype=torch.float64, device=dev))

    def step():
        for eng, r, o in zip(engs, rows, outs):       # each context enqueues on its own stream
            eng.loglike_device(args.walkers, r.data_ptr(), 0, o.data_ptr(), None)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    finite = int(sum(int(torch.isfinite(o).sum().item()) for o in outs))
    if rank == 0:
        total = args.walkers * args.fields * world * args.steps
        b_eval = algorithmic_bytes_per_eval(args.size)
        achieved = b_eval * total / elapsed / world / 1e9
        print(json.dumps({
            'metric': 'walker log-posterior evals/sec, 256x256 image, 1 PSF+1 Sersic',
            'value': total / elapsed, 'unit': 'evals/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64',
            'data': 'synthetic',
            'config': {'workload': '%d independent synthetic %dx%d fields per GPU x %d walkers each, '
                                   '1 PointSource + %d Sersic, fp64' % (args.fields, args.size,
                                                                       args.size, args.walkers,
                                                                       args.sersic),
                       'fields_per_gpu': args.fields, 'walkers_per_field': args.walkers,
                       'backend': args.backend, 'parallelism': 'fields sharded x%d' % world},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBPS, 'traffic': None,
                         'kernel': 'evaluation pipelines of all fields (per GPU)',
                         'bytes_per_eval': b_eval},
            'finite_loglikes': finite}))
    for model, _, _ in probs:
        model.close()
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--walkers', type=int, default=4096, help='walkers per GPU per step')
    ap.add_argument('--size', type=int, default=256)
    ap.add_argument('--sersic', type=int, default=1)
    ap.add_argument('--backend', default=os.environ.get('PSFMC_BACKEND', 'fused'))
    ap.add_argument('--fields', type=int, default=1,
                    help='independent fields per GPU, each with its own context and --walkers '
                         'walkers (BASELINE config 5: 64 fields x 256 walkers over 8 GPUs)')
    ap.add_argument('--chunk', type=int, default=0, help='walkers per internal pass (0 = library default)')
    ap.add_argument('--opt', action='append', default=[], help='library option key=value (tuning)')
    ap.add_argument('--dist-backend', default='nccl',
                    help="'nccl' (RCCL over xGMI; the real thing) or 'gloo' (rehearsal of the "
                         "multi-rank path on a box with fewer GPUs than ranks: ranks share devices)")
    ap.add_argument('--cpu-seconds', type=float, default=12.0)
    ap.add_argument('--no-example', action='store_true', help='skip the configs[1] (example model) rate')
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--cpu-procs', type=int, default=min(16, os.cpu_count() or 1),
                    help='processes of the all-cores CPU baseline (0 = skip)')
    ap.add_argument('--cpu-worker', type=int, default=-1, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_worker >= 0:
        return cpu_worker(args)
    # the all-cores CPU baseline runs first, in child processes, before this process
    # touches the GPU (rank 0, N = 1 only)
    multi = None
    if (not args.no_cpu and args.cpu_procs > 1 and int(os.environ.get('WORLD_SIZE', '1')) == 1):
        multi = cpu_baseline_multi(args, args.cpu_procs)

    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (there is no CPU fallback)')
    if args.dist_backend == 'gloo':          # rehearsal: several ranks may share one device
        local = local % torch.cuda.device_count()
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if args.dist_backend == 'gloo':
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)

    if args.fields > 1:
        return many_fields(args, torch, dist, world, rank, local, dev)
    model, theta, fld = build_problem(args, local)
    eng = model.engine
    if args.chunk:
        eng.set_option('chunk_walkers', args.chunk)
    for kv in args.opt:
        key, val = kv.split('=')
        eng.set_option(key, float(val))
    rows = torch.from_numpy(model.derived_rows(theta)).to(dev)
    out = torch.empty(args.walkers, dtype=torch.float64, device=dev)
    gathered = torch.empty(args.walkers * world, dtype=torch.float64, device=dev)
    # a real (non-NULL) stream: the library launches on the stream it is handed,
    # and the HIP events below must sit on that same stream
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)

    gloo = world > 1 and args.dist_backend == 'gloo'
    if gloo:
        gathered = gathered.cpu()

    def step():
        eng.loglike_device(args.walkers, rows.data_ptr(), 0, out.data_ptr(), stream.cuda_stream)
        if gloo:                         # host staging only in the rehearsal back end
            dist.all_gather_into_tensor(gathered, out.cpu())
        elif world > 1:
            dist.all_gather_into_tensor(gathered, out)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        step()
    ev1.record(stream)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device='cpu' if gloo else dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        # every rank evaluated the same walkers of the same field: the gathered blocks agree
        blocks = gathered.reshape(world, args.walkers)
        assert bool((blocks == blocks[0]).all()), 'ranks disagree on the gathered log-likelihoods'

    # sanity: the batch the timing ran on is numerically right
    lnlike = out.cpu().numpy()
    n_finite = int(np.isfinite(lnlike).sum())

    kernels, chunk = ([], 0)
    if rank == 0 and args.backend == 'fused':
        kernels, chunk = kernel_profile(eng, args, step if world == 1 else
                                        (lambda: eng.loglike_device(args.walkers, rows.data_ptr(), 0,
                                                                    out.data_ptr(), stream.cuda_stream)),
                                        torch, dev, max(2, min(args.steps, 5)))

    if rank == 0:
        total_evals = args.walkers * world * args.steps
        value = total_evals / elapsed
        b_eval = algorithmic_bytes_per_eval(args.size)
        launch_s = dev_ms * 1e-3 / args.steps            # device time of one pipeline pass
        achieved = b_eval * args.walkers / launch_s / 1e9
        line = {
            'metric': 'walker log-posterior evals/sec, 256x256 image, 1 PSF+1 Sersic',
            'value': value, 'unit': 'evals/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'synthetic %dx%d field, 1 PointSource + %d Sersic, %d walkers '
                                   'per GPU per step, fp64, rows resident in HBM'
                                   % (args.size, args.size, args.sersic, args.walkers),
                       'image': args.size, 'walkers_per_gpu': args.walkers,
                       'backend': args.backend, 'parallelism': 'walkers sharded x%d' % world},
            'finite_loglikes': n_finite,
        }
        # whole pipeline against SURVEY.md section 8(d)'s algorithmic bytes per evaluation
        pipe = {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                'frac': achieved / HBM_PEAK_GBPS, 'traffic': None,
                'kernel': 'evaluation pipeline (k_rows_fwd + k_cols + k_rows_inv, one '
                          'eval_batch_device call of %d walkers)' % args.walkers,
                'bytes_per_eval': b_eval, 'launch_ms': launch_s * 1e3}
        if kernels:
            # dominant kernel: its own algorithmic bytes (= the HBM bytes the design
            # moves: it reads and writes the transposed half-spectra once) over its
            # event-timed average launch duration
            k = kernels[0]
            traffic = pmc_traffic(args, k['kernel'])
            line['roofline'] = {
                'bound': 'hbm', 'achieved': k['GBps'], 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                'frac': k['GBps'] / HBM_PEAK_GBPS,
                'traffic': traffic * k['walkers_per_launch'] if traffic else None,
                'kernel': k['kernel'], 'launch_ms': k['avg_ms'], 'launches': k['launches'],
                'bytes_per_launch': k['bytes_per_walker'] * k['walkers_per_launch'],
                'walkers_per_launch': k['walkers_per_launch']}
            line['roofline_pipeline'] = pipe
            line['kernels'] = kernels
        else:
            line['roofline'] = pipe
        # host-inclusive rate: the Python batch call (scipy priors, kappa, rows, H2D, D2H)
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            model.log_posterior_batch(theta)
        line['host_path_evals_per_s'] = args.walkers * reps / (time.perf_counter() - t0)
        if world == 1 and not args.no_example:
            ex = example_model_rate()
            if ex:
                line['example_model_256_walkers'] = ex
        if not args.no_cpu and world == 1:      # CPU baseline: rank 0 at N = 1 only
            base, vals = cpu_baseline(args, fld, theta, args.cpu_seconds)
            line['cpu_baseline'] = base
            if multi:
                line['cpu_baseline_all_cores'] = multi
            ref = np.array(vals)[:len(lnlike)]
            got = lnlike[:len(ref)]
            fin = np.isfinite(ref)
            line['check_vs_cpu_rel'] = float(np.max(np.abs(got[fin] - ref[fin]) / np.abs(ref[fin])))
        print(json.dumps(line))
    model.close()
    if world > 1:
        dist.barrier()              # rank 0 may still have been in its profile pass
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
