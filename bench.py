#!/usr/bin/env python
"""
bench.py -- walker log-posterior evaluations per second (BASELINE.json metric).

A *step* is --batches passes of the hot path, each over one batch of --walkers
synthetic walkers on the 256x256 field with 1 PointSource + 1 Sersic that
BASELINE.json's `metric` is quoted on (SURVEY.md section 8(d) headline: W = 4096).
The timed call is the WHOLE log-posterior: raw emcee parameter vectors resident in
HBM -> unpacking, priors, early-out, Sersic constants, rasteriser, both convolutions,
composite variance, masked Gaussian likelihood, NaN guard -> one double per walker
(`psfmc_eval_theta_device`, psfMC/models.py:193-243 for every walker of the batch).
With N > 1 every batch ends with the all-gather of the log-posteriors across ranks.

`python bench.py --gpus N` with N > 1 starts `python -m torch.distributed.run` with N
ranks as a CHILD process (before this process touches the GPU) and relays its output;
the driver's own `torch.distributed.run ... bench.py --gpus N` form runs the ranks
directly.  One process per GPU; weak scaling (per-GPU batch fixed).

Prints ONE JSON line on rank 0 (contract in the task description), including
  roofline       the dominant kernel (k_cols): its algorithmic bytes per launch (read +
                 write of the transposed half-spectra, = the column pass's share of SURVEY
                 section 8(d)'s per-evaluation figure) over its live HIP-event-timed launch
                 duration; `traffic` from the committed rocprofv3 PMC passes
  roofline_step  the whole step by MEASURED bytes (PMC bytes per walker of the three
                 pipeline kernels x evals/s) against the 8 TB/s peak;
                 `algorithmic_equiv_GBps` is SURVEY section 8(d)'s 96 N^2 + 64 N bytes x
                 evals/s -- an equivalent rate of an unfused design, not a fraction of peak
  cpu_baseline   the numpy oracle (a port of the reference algorithm) timed on this host,
                 one walker per call like the reference
  configs        (N = 1) the other single-GPU workloads of BASELINE.json, each timed for >= 1 s after the
                 headline: config 3 (512^2, 2 Sersic, 1024 walkers), config 4's per-GPU share (1024^2, 4
                 Sersic, 256 walkers), config 5's share (8 fields x 256 walkers: evaluations and stretch-move
                 iterations), one embedded side (170^2) -- value, roofline by measured bytes, the two
                 ceilings and a check of >= 4 walkers against the CPU oracle
  multi_gpu      the product's sharded paths at this N: config 4 strong-scaled
                 (1024^2, 2048 walkers over the ranks), the device sampler with sharded half-steps at 256
                 walkers, config 5 with its 64 fields dealt to the ranks
With N > 1 the timed step itself goes through `parallel.ShardedLogPosterior.evaluate_device`: N x --walkers
vectors resident on every rank, each rank evaluating its block, one RCCL all-gather per batch.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

# the host driver only supports dmabuf IPC: RCCL needs this before the HIP runtime starts
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'tools')):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBPS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md, chip-level table
BYTES_NOTE = ('FETCH_SIZE/WRITE_SIZE count requests at the L2 memory side (fabric); the pass size keeps '
              'the transposed half-spectra inside the 256 MiB Infinity Cache, so part of these bytes '
              'is served on-die, not from DRAM')


def algorithmic_bytes_per_eval(n):
    return 96 * n * n + 64 * n


def pmc_table(args):
    """HBM-side bytes per walker of each pipeline kernel from the committed rocprofv3 PMC
    passes (profiles/pmc_traffic.json; FETCH_SIZE doubled per MI355X_MICROARCH.md section
    HBM, plus WRITE_SIZE), or {} when that shape was not profiled."""
    path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    size = args if isinstance(args, int) else args.size
    try:
        with open(path) as f:
            return json.load(f)['%dx%d' % (size, size)]
    except (IOError, OSError, KeyError, ValueError):
        return {}


def pmc_lookup(table, prefix):
    # (the one-row-per-wave row kernels of psfmc_rows3_path.h are named k_rows3_*)
    for pre in (prefix, prefix.replace('k_rows_', 'k_rows3_')):
        for name, rec in table.items():
            if name.startswith(pre):
                return rec['hbm_bytes_per_walker']
    return None


def physical_cores():
    """(cores this process may use, logical CPUs of the machine): one entry per distinct
    (package, core) among the CPUs of the affinity mask."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        allowed = list(range(os.cpu_count() or 1))
    seen = set()
    for cpu in allowed:
        base = '/sys/devices/system/cpu/cpu%d/topology/' % cpu
        try:
            with open(base + 'physical_package_id') as f:
                pkg = f.read().strip()
            with open(base + 'core_id') as f:
                core = f.read().strip()
            seen.add((pkg, core))
        except (IOError, OSError):
            seen.add(('cpu', cpu))
    return max(len(seen), 1), os.cpu_count() or 1


def cgroup_cpu_quota():
    """CPUs the cgroup's bandwidth controller lets this job use at once (cpu.max = "quota period",
    cgroup v2; cpu.cfs_quota_us / cpu.cfs_period_us, v1), or None when unlimited / unreadable.  The
    affinity mask alone says nothing about it: the GPU boxes of this pool show 256 logical CPUs in
    the mask and a quota of 16."""
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:
            quota, period = f.read().split()[:2]
        return None if quota == 'max' else float(quota) / float(period)
    except (IOError, OSError, ValueError):
        pass
    try:
        with open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us') as f:
            quota = float(f.read())
        with open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as f:
            period = float(f.read())
        return None if quota <= 0 else quota / period
    except (IOError, OSError, ValueError):
        return None


def cgroup_throttled_usec():
    try:
        with open('/sys/fs/cgroup/cpu.stat') as f:
            for line in f:
                if line.startswith('throttled_usec'):
                    return int(line.split()[1])
    except (IOError, OSError, ValueError):
        pass
    return None


def usable_cores():
    """Processes the all-cores CPU baseline may run without oversubscribing the job's CPU share:
    min(physical cores in the affinity mask, cgroup CPU quota).  Returns (n, detail dict)."""
    n_phys, n_logical = physical_cores()
    quota = cgroup_cpu_quota()
    n = n_phys if quota is None else max(1, min(n_phys, int(quota)))
    return n, {'physical_cores_in_affinity_mask': n_phys, 'host_logical_cpus': n_logical,
               'cgroup_cpu_quota': quota}


def draw_theta(args, fld, n, seed=1, size=None, sersic=None):
    import synth_field
    size, sersic = size or args.size, args.sersic if sersic is None else sersic
    half = n // 2
    return np.vstack([
        synth_field.draw_walkers(size, sersic, half, seed=seed),
        synth_field.draw_walkers(size, sersic, n - half, seed=seed + 1,
                                 near_truth=fld['truth'])])


def build_problem(args, device, seed=0, size=None, sersic=None, walkers=None, max_walkers=None):
    """Synthetic field + walker vectors.  Returns (model, theta[W,P], field dict).  size / sersic / walkers
    default to the command line's; max_walkers (the context's capacity) to `walkers`."""
    import tempfile
    import synth_field
    from psfmc_amd import MultiComponentModel, fits_io
    size, sersic = size or args.size, args.sersic if sersic is None else sersic
    walkers = walkers or args.walkers
    fld = synth_field.make_field(size, sersic, seed=seed)
    tmp = tempfile.mkdtemp(prefix='psfmc_bench_')
    for key, name in (('sci', 'sci.fits'), ('ivm', 'ivm.fits'), ('psf', 'psf.fits'),
                      ('psf_ivm', 'psf_ivm.fits')):
        fits_io.write_image(os.path.join(tmp, name), fld[key])
    path = os.path.join(tmp, 'model.py')
    with open(path, 'w') as f:
        f.write(synth_field.model_file_text(size, sersic))
    model = MultiComponentModel(path, device=device, backend=args.backend,
                                max_walkers=max_walkers or walkers)
    return model, draw_theta(args, fld, walkers, size=size, sersic=sersic), fld


def column_kernel_name(eng, n):
    """The column kernel this context launches NOW (the library's own decision: options such as cols3 and
    storage_f32 and the row-group size go into it), or psfmc_amd.engine.column_engine's static answer for a
    library that does not report it."""
    code = eng.get_option('column_engine')
    if code == code:
        return {0: 'k_cols', 1: 'k_cols3', 2: 'k_cols3g', 3: 'k_cols3f'}[int(code)]
    from psfmc_amd.engine import column_engine
    return column_engine(n)[0]


def kernel_profile(eng, args, one_batch, torch, dev, reps):
    """Per-kernel device time from HIP events recorded inside the library around
    every launch (set_option 'profile').  Run with ONE pass in flight so that a
    kernel's events bracket that kernel alone -- the same quantity rocprofv3
    --kernel-trace reports.  Returns a list of dicts, longest total first."""
    n = args.size
    nxh = n // 2 + 1
    t_bytes = 2 * nxh * n * 16                  # transposed half-spectra of one walker
    designed = {'rows_fwd': t_bytes, 'cols': 2 * t_bytes, 'rows_inv': t_bytes}
    rows3 = eng.get_option('rows3')                 # the one-row-per-wave three-stage row kernels: bit 0 forward, bit 1 inverse
    rows3 = int(rows3) if rows3 == rows3 else 0
    names = {'rows_fwd': ('k_rows3_fwd<%d, false>' if rows3 & 1 else 'k_rows_fwd<%d, false>') % n,
             'rows_inv': ('k_rows3_inv<%d>' if rows3 & 2 else 'k_rows_inv<%d>') % n,
             'cols': '%s<%d, true>' % (column_kernel_name(eng, n), n)}
    streams = eng.get_option('streams')
    eng.set_option('streams', 1)
    eng.set_option('profile', 1)
    for _ in range(reps):
        one_batch()
    torch.cuda.synchronize(dev)
    out = []
    for key in ('rows_fwd', 'cols', 'rows_inv'):
        ms = eng.get_option('prof_ms_' + key)
        cnt = eng.get_option('prof_n_' + key)
        if not cnt or cnt != cnt:            # 0, or NaN from a library that does not know the key
            continue
        avg_s = ms * 1e-3 / cnt
        walkers = args.walkers * reps / cnt          # walkers one launch processes (avg)
        out.append({'kernel': names[key], 'launches': int(cnt), 'avg_ms': avg_s * 1e3,
                    'total_ms': ms, 'walkers_per_launch': walkers,
                    'bytes_per_walker': designed[key],
                    'GBps': designed[key] * walkers / avg_s / 1e9})
    eng.set_option('profile', 0)
    eng.set_option('streams', streams)
    out.sort(key=lambda d: -d['total_ms'])
    return out


def sweep_ceiling(engine, device, nbytes, walkers, rate_gpu, reps=20):
    """What this GPU gives the data flow of one pass at best: three plain sweeps (the library's
    psfmc_debug_sweep kernels, no arithmetic) over a buffer the size of one pass's transposed
    half-spectra T -- written once (k_rows_fwd), read and written back in place (k_cols), read
    once (k_rows_inv).  Their summed time is a floor for the three-kernel pass on this box;
    `step_over_floor` compares the timed step with it."""
    out = {'buffer_bytes': nbytes, 'walkers_per_pass': walkers}
    floor_us = 0.0
    for name, passes in (('write', 1), ('read_write', 2), ('read', 1)):
        us = engine.debug_sweep(name, nbytes, reps, device)
        out[name + '_us'] = us
        out[name + '_GBps'] = passes * nbytes / us / 1e3
        floor_us += us
    out['pass_floor_us'] = floor_us
    out['step_us_per_pass'] = walkers / rate_gpu * 1e6
    out['step_over_floor'] = out['step_us_per_pass'] / floor_us
    out['note'] = ('plain write / in-place read-modify-write / read sweeps over one pass of T (which the '
                   'library sizes for the Infinity Cache), timed with HIP events inside the library')
    return out


def valu_ceiling(engine, device, table, walkers, rate_gpu, n_simd):
    """The other ceiling: the vector-instruction issue rate of THIS chip with every SIMD busy (psfmc_debug_valu_rate:
    64 v_fma_f64 per loop iteration, two waves per SIMD, HIP events) times the vector wave-instructions the three
    kernels issue per walker (SQ_INSTS_VALU of the committed PMC passes, profiles/pmc_traffic.json).  With two and
    four Sersic components the 512^2 / 1024^2 passes carry more VALU time than sweep time."""
    per_walker = [rec.get('valu_wave_instructions_per_walker') for name, rec in table.items()
                  if name.startswith(('k_rows_fwd<', 'k_cols', 'k_rows_inv<'))]
    if len(per_walker) != 3 or not all(per_walker):
        return None
    ns = engine.debug_valu_rate(2, 20000, device)
    instr = float(sum(per_walker))
    floor_us = instr * walkers * ns / n_simd / 1e3
    step_us = walkers / rate_gpu * 1e6
    return {'ns_per_fp64_wave_instruction_per_simd': ns, 'valu_wave_instructions_per_walker': instr,
            'simds': n_simd, 'walkers_per_pass': walkers, 'pass_floor_us': floor_us, 'step_us_per_pass': step_us,
            'step_over_valu_floor': step_us / floor_us,
            'note': 'fp64 vector issue rate measured on every SIMD of the chip at two waves per SIMD x the PMC '
                    'instruction counts of the three kernels; a floor only if the instruction mix issued like v_fma_f64 '
                    '(v_rcp_f64 takes 4x as long)'}


def small_ensembles(eng, args, torch, dev, theta_dev, out_dev, stream):
    """Half-steps of the reference's default ensembles (chains = 2 P + 2, psfMC/fitting.py:
    52-53 -> 11 and 19 walkers per half-step for 10 / 18 parameters) and a few more: device
    time per call of the whole log-posterior, back to back on one stream."""
    res = {}
    for w in (11, 19, 32, 64, 128, 256):
        if w > args.walkers:
            continue
        for _ in range(5):
            eng.logpost_theta_device(w, theta_dev.data_ptr(), 0, out_dev.data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize(dev)
        reps = 200
        t0 = time.perf_counter()
        for _ in range(reps):
            eng.logpost_theta_device(w, theta_dev.data_ptr(), 0, out_dev.data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) / reps
        res[str(w)] = {'us_per_call': dt * 1e6, 'evals_per_s': w / dt}
    return res


def default_ensemble_sampler(model, theta, n_iter=64):
    """The reference's default fit: `chains = 2 P + 2` walkers (psfMC/fitting.py:52-53) stepped by the
    device-resident sampler (psfmc_stretch_run), ms per stretch-move iteration -- as the library runs it (small
    ensembles: ONE pipeline pass per iteration over the first half's proposals and both candidate proposals
    of every second-half walker) and as two half-steps (option speculate = 0); same chain bit for bit."""
    from psfmc_amd.sampler import DeviceEnsembleSampler
    eng = model.engine
    n_w = 2 * model.num_params + 2
    p0 = np.ascontiguousarray(theta[:n_w])
    lnp = model.log_posterior_batch(p0)
    if not np.all(np.isfinite(lnp)):
        good = theta[np.isfinite(model.log_posterior_batch(theta[:8 * n_w]))]
        if len(good) < n_w:
            return None
        p0 = np.ascontiguousarray(good[:n_w])
        lnp = model.log_posterior_batch(p0)
    s = DeviceEnsembleSampler(n_w, model, block=n_iter)
    s.random_state = np.random.RandomState(5).get_state()
    draws, _ = s._draw(n_iter)
    out, chains = {}, {}
    for key, mode in (('ms_per_iteration', -1), ('ms_per_iteration_two_half_steps', 0)):
        eng.set_option('speculate', mode)
        best = None
        for _ in range(4):
            t0 = time.perf_counter()
            res = eng.stretch_run(p0.copy(), lnp.copy(), *draws, np.zeros(n_w, dtype=np.int64), store=True)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        out[key] = best * 1e3 / n_iter
        chains[key] = res[2]
    eng.set_option('speculate', -1)
    out['walkers'] = n_w
    out['evals_per_s'] = n_w * n_iter / (out['ms_per_iteration'] * 1e-3 * n_iter)
    out['chains_identical'] = bool(np.array_equal(*chains.values()))
    out['what'] = ('%d walkers (2 P + 2), %d iterations per psfmc_stretch_run call, upload and chain download '
                   'included; best of 4' % (n_w, n_iter))
    return out


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
            'sample': '%d walkers of the same batch, one per call, %.1f s (log-likelihood; the '
                      'reference adds ~15 scipy.stats prior calls per walker on top)' % (done, el)}, vals


def cpu_worker(args):
    """Child process of cpu_baseline_multi: evaluate walkers with the oracle for
    --cpu-seconds and print how many were done (no GPU, no torch)."""
    import synth_field
    fld = synth_field.make_field(args.size, args.sersic, seed=0)
    theta = draw_theta(args, fld, args.walkers)
    t0 = time.perf_counter()
    _, vals = cpu_baseline(args, fld, theta[args.cpu_worker::7], args.cpu_seconds)
    print('CPU_WORKER_DONE %d %.3f' % (len(vals), time.perf_counter() - t0))


def cpu_baseline_multi(args, n_procs, detail, single_rate=None):
    """The best the reference could do had `threads=n` worked: one
    single-threaded process per core this job may really use (usable_cores: affinity mask AND cgroup
    quota), walkers split between them.  The per-process rates are on the line so that a collapse
    (oversubscription, a throttled cgroup) is visible; `valid` is false when the aggregate does not
    even reach the single-process rate."""
    env = dict(os.environ, OMP_NUM_THREADS='1', MKL_NUM_THREADS='1', OPENBLAS_NUM_THREADS='1')
    cmd = [sys.executable, os.path.abspath(__file__), '--size', str(args.size), '--sersic',
           str(args.sersic), '--walkers', str(args.walkers), '--cpu-seconds', str(args.cpu_seconds)]
    thr0 = cgroup_throttled_usec()
    t0 = time.perf_counter()
    procs = [subprocess.Popen(cmd + ['--cpu-worker', str(i)], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.DEVNULL, text=True) for i in range(n_procs)]
    rates = []
    for p in procs:
        out, _ = p.communicate(timeout=args.cpu_seconds * 4 + 120)
        for line in out.splitlines():
            if line.startswith('CPU_WORKER_DONE'):
                _, n_done, secs = line.split()
                rates.append(int(n_done) / float(secs))
    el = time.perf_counter() - t0
    thr1 = cgroup_throttled_usec()
    total = float(sum(rates))
    res = {'value': total, 'unit': 'evals/s', 'cores': n_procs, 'kind': 'port',
           'per_process_evals_per_s': {'min': min(rates) if rates else None,
                                       'median': float(np.median(rates)) if rates else None,
                                       'max': max(rates) if rates else None},
           'cgroup_throttled_s_during_run': (thr1 - thr0) * 1e-6 if thr0 is not None and thr1 is not None else None,
           'sample': '%d single-threaded processes (min of the physical cores in the affinity mask and the '
                     'cgroup CPU quota) x %.0f s of walkers of the same batch, each timing its own loop '
                     '(wall %.1f s incl. start-up)' % (n_procs, args.cpu_seconds, el)}
    res.update(detail)
    if single_rate:
        res['speedup_over_one_process'] = total / single_rate
        res['valid'] = bool(total >= single_rate and len(rates) == n_procs)
    return res


class Watchdog(object):
    """Per-rank deadline around the phases of a multi-rank run that can hang (rendezvous, RCCL
    communicator set-up, the first collective): when a phase overruns, the rank prints ONE JSON error
    line naming itself and the phase and leaves with a non-zero code -- no retry, no re-exec; the
    launcher (torch.distributed.run) then tears the other ranks down."""

    def __init__(self, rank, world):
        self.rank, self.world, self._timer, self.name = rank, world, None, None

    def _expired(self, name, seconds):
        print(json.dumps({'error': 'timeout', 'rank': self.rank, 'world_size': self.world, 'phase': name,
                          'limit_s': seconds, 'host': socket.gethostname(),
                          'hint': 'rendezvous: MASTER_ADDR / MASTER_PORT; communicator: '
                                  'HSA_ENABLE_IPC_MODE_LEGACY=0, one visible GPU per LOCAL_RANK'}), flush=True)
        os._exit(3)

    def phase(self, name, seconds):
        import threading
        self.done()
        self.name = name
        self._timer = threading.Timer(seconds, self._expired, (name, seconds))
        self._timer.daemon = True
        self._timer.start()

    def done(self):
        if self._timer is not None:
            self._timer.cancel()
            self._timer = None

    def failed(self, exc):
        self.done()
        print(json.dumps({'error': type(exc).__name__, 'rank': self.rank, 'world_size': self.world,
                          'phase': self.name, 'message': str(exc)[:400]}), flush=True)
        os._exit(4)


class Salvage(object):
    """Deadline around the side figures of a multi-rank run (collectives the timed region did not use): when they
    overrun, rank 0 prints the headline line it already holds -- with the overrun named under `multi_gpu` -- and
    every rank leaves with code 0, so that a hang in a side figure cannot cost the measured headline."""

    def __init__(self, rank, seconds):
        import threading
        self.rank, self.seconds, self.line = rank, seconds, None
        self._timer = threading.Timer(seconds, self._expired)
        self._timer.daemon = True

    def arm(self, line):
        self.line = line
        self._timer.start()

    def _expired(self):
        if self.rank == 0 and self.line is not None:
            self.line['multi_gpu'] = {'error': 'timeout', 'limit_s': self.seconds,
                                      'note': 'the side figures of the sharded paths did not finish; the headline '
                                              'above was measured before they started'}
            print(json.dumps(self.line), flush=True)
        os._exit(0)

    def done(self):
        self._timer.cancel()


def guarded(name, fn, *a, **kw):
    """A side figure must not cost the headline: an exception becomes an `error` entry under its key (and a line
    on stderr)."""
    try:
        return fn(*a, **kw)
    except Exception as exc:               # noqa: BLE001 -- whatever it was, name it and go on
        import traceback
        traceback.print_exc()
        sys.stderr.write('bench.py: side figure %s failed: %r\n' % (name, exc))
        return {'error': type(exc).__name__, 'message': str(exc)[:400]}


def spawn_ranks(args, argv):
    """`bench.py --gpus N` outside a torch.distributed launch: start the N ranks as a child
    process (this process has not touched the GPU), relay the output, return its exit code."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'),
               OMP_NUM_THREADS=os.environ.get('OMP_NUM_THREADS', '1'))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd, env=env)


def timed_region(args, torch, dist, world, dev, step, gloo):
    """W untimed steps, then exactly K steps between barrier + synchronize on both sides;
    returns the MAX over ranks of the wall time."""
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
        tmax = torch.tensor([elapsed], dtype=torch.float64, device='cpu' if gloo else dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    return elapsed


def fields_mcmc_rates(args, probs, merged, n_iter=60):
    """Config 5 as a FIT (psfMC/fitting.py:56-86 for every field): stretch-move iterations per second
    with the fields' ensembles stepped TOGETHER in the shared context (`FieldSetSampler`) and one after
    the other in their own contexts (`DeviceEnsembleSampler` each); an iteration moves every walker of
    every field once (two half-steps), posterior images accumulated every iteration."""
    import synth_field
    from psfmc_amd import FieldSetSampler, DeviceEnsembleSampler
    p0 = [synth_field.draw_walkers(args.size, args.sersic, args.walkers, seed=300 + f, near_truth=fld['truth'])
          for f, (_, _, fld) in enumerate(probs)]
    out = {}
    for accumulate in (False, True):
        joint = FieldSetSampler(args.walkers, merged, accumulate=accumulate)
        for f, sub in enumerate(joint.fields):
            sub.random_state = np.random.RandomState(900 + f).get_state()
        for _ in joint.sample(p0, iterations=4):
            pass
        t0 = time.perf_counter()
        for _ in joint.sample(p0, iterations=n_iter):
            pass
        t_joint = time.perf_counter() - t0
        t_own = 0.0
        for f, (model, _, _) in enumerate(probs):
            solo = DeviceEnsembleSampler(args.walkers, model, accumulate=accumulate)
            solo.random_state = np.random.RandomState(900 + f).get_state()
            for _ in solo.sample(p0[f], iterations=4):
                pass
            t0 = time.perf_counter()
            for _ in solo.sample(p0[f], iterations=n_iter):
                pass
            t_own += time.perf_counter() - t0
        evals = args.fields * args.walkers * n_iter
        out['accumulating_images' if accumulate else 'sampling_only'] = {
            'one_context_iterations_per_s': n_iter / t_joint, 'one_context_evals_per_s': evals / t_joint,
            'own_contexts_iterations_per_s': n_iter / t_own, 'own_contexts_evals_per_s': evals / t_own}
    out['what'] = ('%d fields x %d walkers, %d iterations after 4 of warm-up; an iteration = two half-steps of '
                   'every field' % (args.fields, args.walkers, n_iter))
    return out


def many_fields(args, torch, dist, world, rank, local, dev, gloo):
    """BASELINE config 5: every rank owns --fields independent fields (own context,
    own streams, own walkers); a step evaluates all of them (raw vectors -> log-posterior).
    Fields shard across ranks with no data-path collective."""
    probs = [build_problem(args, local, seed=rank * args.fields + f) for f in range(args.fields)]
    merged = None
    if args.fields_merge:
        # the fields' walkers share ONE context and one batch (psfmc_ctx_create_fields)
        # (its own model objects: a FieldSet routes its models through the shared context, and the
        # fields' own contexts are evaluated next to it)
        from psfmc_amd import FieldSet
        twins = [build_problem(args, local, seed=rank * args.fields + f)[0] for f in range(args.fields)]
        merged = FieldSet(twins, max_walkers=args.fields * args.walkers, device=local)
        all_theta = torch.from_numpy(np.concatenate([t for _, t, _ in probs])).to(dev)
        all_out = torch.empty(args.fields * args.walkers, dtype=torch.float64, device=dev)
        seg_f, seg_n = list(range(args.fields)), [args.walkers] * args.fields
        if args.chunk:
            merged.context.set_option('chunk_walkers', args.chunk)
        for kv in args.opt:
            key, val = kv.split('=')
            merged.context.set_option(key, float(val))
    engs, thetas, outs = [], [], []
    for model, theta, _ in probs:
        engs.append(model.engine)
        if args.chunk:
            model.engine.set_option('chunk_walkers', args.chunk)
        for kv in args.opt:
            key, val = kv.split('=')
            model.engine.set_option(key, float(val))
        thetas.append(torch.from_numpy(theta).to(dev))
        outs.append(torch.empty(args.walkers, dtype=torch.float64, device=dev))

    def step():
        for _ in range(args.batches):
            if merged is not None:
                merged.context.logpost_theta_device(seg_f, seg_n, all_theta.data_ptr(), all_out.data_ptr(), None)
                continue
            for eng, t, o in zip(engs, thetas, outs):       # each context enqueues on its own stream
                eng.logpost_theta_device(args.walkers, t.data_ptr(), 0, o.data_ptr(), None)

    elapsed = timed_region(args, torch, dist, world, dev, step, gloo)
    finite = int(sum(int(torch.isfinite(o).sum().item()) for o in outs))
    agree = None
    if merged is not None:
        finite = int(torch.isfinite(all_out).sum().item())
        # the shared batch against each field's own context, bit for bit
        for eng, t, o in zip(engs, thetas, outs):
            eng.logpost_theta_device(args.walkers, t.data_ptr(), 0, o.data_ptr(), None)
        torch.cuda.synchronize(dev)
        agree = bool(torch.equal(torch.cat(outs), all_out))
    mcmc = None
    if rank == 0 and merged is not None and not args.no_extras:
        mcmc = fields_mcmc_rates(args, probs, merged)
    if rank == 0:
        total = args.walkers * args.fields * args.batches * world * args.steps
        b_eval = algorithmic_bytes_per_eval(args.size)
        table = pmc_table(args)
        per_walker = [pmc_lookup(table, k) for k in ('k_rows_fwd<', 'k_cols', 'k_rows_inv<')]
        measured = sum(per_walker) if all(per_walker) else None
        rate_gpu = total / elapsed / world
        print(json.dumps({
            'metric': 'walker log-posterior evals/sec, 256x256 image, 1 PSF+1 Sersic',
            'value': total / elapsed, 'unit': 'evals/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64',
            'data': 'synthetic',
            'config': {'workload': '%d independent synthetic %dx%d fields per GPU x %d walkers each x %d '
                                   'batches per step, 1 PointSource + %d Sersic, raw vectors resident in '
                                   'HBM -> log-posterior, fp64' % (args.fields, args.size, args.size,
                                                                   args.walkers, args.batches, args.sersic),
                       'fields_per_gpu': args.fields, 'walkers_per_field': args.walkers,
                       'one_context_for_all_fields': bool(args.fields_merge),
                       'shared_batch_equals_own_contexts_bitwise': agree,
                       'backend': args.backend, 'parallelism': 'fields sharded x%d' % world},
            'roofline': {'bound': 'hbm', 'achieved': measured * rate_gpu / 1e9 if measured else None,
                         'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                         'frac': measured * rate_gpu / 1e9 / HBM_PEAK_GBPS if measured else None,
                         'traffic': measured * args.walkers * args.fields * args.batches if measured else None,
                         'kernel': 'evaluation pipelines of all fields (per GPU), measured PMC bytes per walker',
                         'algorithmic_equiv_GBps': b_eval * rate_gpu / 1e9, 'bytes_note': BYTES_NOTE},
            'finite_logposts': finite, 'mcmc': mcmc}))
    if merged is not None:
        merged.context.close()
    for model, _, _ in probs:
        model.close()
    if world > 1:
        dist.destroy_process_group()



def timed_calls(call, sync, min_s=1.0, decide=None):
    """Seconds per call of `call()` (asynchronous launches; `sync()` waits for the device): two calls
    sized, then enough of them for >= min_s between two synchronisations.  `decide(n)` lets rank 0's
    count be every rank's (multi-rank runs).  Returns (seconds per call, calls timed)."""
    call()
    sync()
    t0 = time.perf_counter()
    call()
    call()
    sync()
    one = max((time.perf_counter() - t0) / 2, 1e-6)
    n = max(3, int(np.ceil(1.3 * min_s / one)))        # (the sized pair includes a synchronisation the loop does not)
    if decide is not None:
        n = decide(n)
    t0 = time.perf_counter()
    for _ in range(n):
        call()
    sync()
    return (time.perf_counter() - t0) / n, n


def oracle_check(model, fld, sersic, theta, got, count=4):
    """max relative difference of `count` walkers' log-posteriors (spread over the batch) from the CPU
    oracle's log-likelihood + the host priors (scipy): the bench's own parity check of a timed batch."""
    for p_ in (os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
        if p_ not in sys.path:
            sys.path.insert(0, p_)
    import psfmc_oracle as orc
    import helpers
    field = orc.make_field(fld['sci'], fld['ivm'], [fld['psf']], [fld['psf_ivm']], mag_zp=fld['mag_zp'])
    layout = helpers.synth_layout(sersic)
    idx = np.unique(np.linspace(0, len(theta) - 1, count * 3).astype(int))
    prior = model.log_priors_batch(theta[idx])
    idx, prior = idx[np.isfinite(prior)][:count], prior[np.isfinite(prior)][:count]   # (walkers that reach the likelihood)
    worst = 0.0
    for i, lp in zip(idx, prior):
        want = helpers.oracle_loglike(field, layout, theta[i]) + lp
        if np.isfinite(want):
            worst = max(worst, abs(got[i] - want) / abs(want))
        elif np.isfinite(got[i]):
            return float('inf'), len(idx)
    return float(worst), len(idx)


def step_rooflines(size, rate, per_pass, engine_mod, device, n_simd):
    """The step's roofline by measured bytes and the two measured ceilings for a rate of `rate` evals/s at
    `per_pass` walkers per pass (objects as on the headline line)."""
    table = pmc_table(size)
    per_walker = [pmc_lookup(table, k) for k in ('k_rows_fwd<', 'k_cols', 'k_rows_inv<')]
    measured = sum(per_walker) if all(per_walker) else None
    out = {'roofline_step': {
        'bound': 'hbm', 'achieved': measured * rate / 1e9 if measured else None, 'peak': HBM_PEAK_GBPS,
        'unit': 'GB/s', 'frac': measured * rate / 1e9 / HBM_PEAK_GBPS if measured else None,
        'measured_bytes_per_eval': measured,
        'algorithmic_equiv_GBps': algorithmic_bytes_per_eval(size) * rate / 1e9}}
    t_bytes = 2 * (size // 2 + 1) * size * 16
    sc = sweep_ceiling(engine_mod, device, t_bytes * per_pass, per_pass, rate)
    out['sweep_ceiling'] = {k: sc[k] for k in ('pass_floor_us', 'step_us_per_pass', 'step_over_floor', 'walkers_per_pass')}
    vc = valu_ceiling(engine_mod, device, table, per_pass, rate, n_simd)
    if vc:
        out['valu_ceiling'] = {k: vc[k] for k in ('pass_floor_us', 'step_over_valu_floor',
                                                  'valu_wave_instructions_per_walker')}
        out['step_over_larger_floor'] = sc['step_us_per_pass'] / max(vc['pass_floor_us'], sc['pass_floor_us'])
    return out


def other_configs(args, torch, dev, local):
    """N = 1: the single-GPU workloads of BASELINE.json besides the headline, each for >= 1 s of timed
    device-resident evaluations after a warm-up, on this same box in this same run."""
    from psfmc_amd import engine as engine_mod
    n_simd = 4 * torch.cuda.get_device_properties(dev).multi_processor_count
    sync = lambda: torch.cuda.synchronize(dev)
    out = {}
    for key, size, sersic, walkers, what in (
            ('config3_512', 512, 2, 1024, 'BASELINE config 3: synthetic 512x512, 1 PointSource + 2 Sersic, 1024 walkers'),
            ('config4_share_1024', 1024, 4, 256, "BASELINE config 4's per-GPU share: synthetic 1024x1024, 1 PointSource + "
                                                 '4 Sersic, 256 of the 2048 walkers'),
            ('embedded_170', 170, 1, 4096, 'a side the transforms are not built for (170x170, embedded), 1 PointSource + '
                                           '1 Sersic, 4096 walkers'),
            # not BASELINE configs: one mixed-radix side and one side above 1024, so that the driver's own run holds a
            # number for each family of kernels (general two-stage rows + three-stage columns; one-row-per-wave rows)
            ('general_300', 300, 1, 2048, 'a general side (300x300 = 15 x 20 rows, 5 x (5 x 12) columns), 1 PointSource + '
                                          '1 Sersic, 2048 walkers'),
            ('large_2048', 2048, 1, 64, 'a side above 1024 (2048x2048: one-row-per-wave three-stage row kernels, '
                                        'k_cols3f columns), 1 PointSource + 1 Sersic, 64 walkers')):
        model, theta, fld = build_problem(args, local, size=size, sersic=sersic, walkers=walkers)
        eng = model.engine
        th = torch.from_numpy(theta).to(dev)
        res = torch.empty(walkers, dtype=torch.float64, device=dev)
        stream = torch.cuda.Stream(dev)
        call = lambda: eng.logpost_theta_device(walkers, th.data_ptr(), 0, res.data_ptr(), stream.cuda_stream)
        per_call, n_calls = timed_calls(call, sync)
        got = res.cpu().numpy()
        rel, n_chk = oracle_check(model, fld, sersic, theta, got, count=2 if size > 1024 else 4)
        rec = {'workload': what + ', raw vectors resident in HBM -> log-posterior, fp64', 'value': walkers / per_call,
               'unit': 'evals/s', 'timed_s': per_call * n_calls, 'calls': n_calls, 'walkers_per_call': walkers,
               'finite_logposts': int(np.isfinite(got).sum()), 'check_vs_cpu_rel': rel, 'walkers_checked': n_chk,
               'transform': [int(eng.get_option('transform_ny')), int(eng.get_option('transform_nx'))]}
        if args.backend == 'fused':
            rec.update(step_rooflines(int(eng.get_option('transform_nx')), rec['value'], eng.pass_size(walkers),
                                      engine_mod, local, n_simd))
        out[key] = rec
        model.close()
        del th, res
    # config 5's per-GPU share: 8 independent 256^2 fields x 256 walkers in ONE context -- evaluations, and as a fit
    import synth_field
    from psfmc_amd import FieldSet, FieldSetSampler
    n_f, n_w = 8, 256
    probs = [build_problem(args, local, seed=100 + f, size=256, sersic=1, walkers=n_w) for f in range(n_f)]
    fs = FieldSet([m for m, _, _ in probs], max_walkers=n_f * n_w, device=local)
    all_theta = torch.from_numpy(np.concatenate([t for _, t, _ in probs])).to(dev)
    all_out = torch.empty(n_f * n_w, dtype=torch.float64, device=dev)
    seg_f, seg_n = list(range(n_f)), [n_w] * n_f
    call = lambda: fs.context.logpost_theta_device(seg_f, seg_n, all_theta.data_ptr(), all_out.data_ptr(), None)
    per_call, n_calls = timed_calls(call, sync)
    got = all_out.cpu().numpy()
    rel = max(oracle_check(probs[f][0], probs[f][2], 1, probs[f][1], got[f * n_w:(f + 1) * n_w], count=1)[0]
              for f in (0, 3, 5, 7))
    rec = {'workload': "BASELINE config 5's per-GPU share: 8 independent synthetic 256x256 fields x 256 walkers in one "
                       'context (psfmc_ctx_create_fields), 1 PointSource + 1 Sersic, raw vectors resident in HBM -> '
                       'log-posterior, fp64',
           'value': n_f * n_w / per_call, 'unit': 'evals/s', 'timed_s': per_call * n_calls, 'calls': n_calls,
           'finite_logposts': int(np.isfinite(got).sum()), 'check_vs_cpu_rel': rel, 'walkers_checked': 4}
    rec.update(step_rooflines(256, rec['value'], int(fs.context.get_option('chunk_walkers')), engine_mod, local, n_simd))
    sampler = FieldSetSampler(n_w, fs, accumulate=False)
    p0 = [synth_field.draw_walkers(256, 1, n_w, seed=300 + f, near_truth=probs[f][2]['truth']) for f in range(n_f)]
    for f, sub in enumerate(sampler.fields):
        sub.random_state = np.random.RandomState(900 + f).get_state()
    for _ in sampler.sample(p0, iterations=4):
        pass
    n_iter, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < 1.0:
        for _ in sampler.sample(p0, iterations=32):
            pass
        n_iter += 32
    dt = time.perf_counter() - t0
    rec['mcmc'] = {'stretch_move_iterations_per_s': n_iter / dt, 'evals_per_s': n_f * n_w * n_iter / dt,
                   'iterations': n_iter, 'timed_s': dt,
                   'what': 'the 8 ensembles stepped together (FieldSetSampler); an iteration moves every walker of '
                           'every field once'}
    out['config5_share_8_fields'] = rec
    fs.close()
    for m, _, _ in probs:
        m.close()
    return out


def multi_gpu_extras(args, torch, dist, world, rank, local, dev, gloo):
    """The product's sharded paths at this world size, every rank taking part:
    config 4 strong-scaled, the device sampler with sharded half-steps at 256 walkers, config 5's 64 fields
    dealt to the ranks.  Times are the MAX over ranks; rank 0 returns the object."""
    import synth_field
    from psfmc_amd import FieldSet
    from psfmc_amd.parallel import RankGroup, ShardedLogPosterior, shard_bounds
    from psfmc_amd.sampler import DeviceEnsembleSampler
    rg = RankGroup(None, dev) if world > 1 else None
    sync = lambda: torch.cuda.synchronize(dev)

    def decide(n):
        return int(rg.broadcast_object(n)) if rg is not None else n

    def rank_max(seconds):
        if world == 1:
            return seconds
        t = torch.tensor([seconds], dtype=torch.float64, device='cpu' if gloo else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    out = {'n_gpus': world, 'dist_backend': (args.dist_backend if world > 1 else None)}
    # (1) config 4, strong scaling: 1024^2, 1 PS + 4 Sersic, 2048 walkers over the ranks, one all-gather per batch
    n_w = args.config4_walkers
    block = -(-n_w // world)
    model, theta, fld = build_problem(args, local, size=1024, sersic=4, walkers=n_w, max_walkers=block)
    th = torch.from_numpy(theta).to(dev)
    sharded = ShardedLogPosterior(model, group=rg, device=dev)
    keep = {}

    def call():
        keep['out'] = sharded.evaluate_device(th)
    if world > 1:
        dist.barrier()
    per_call, n_calls = timed_calls(call, sync, decide=decide)
    per_call = rank_max(per_call)
    got = keep['out'].cpu().numpy()
    rel, n_chk = (oracle_check(model, fld, 4, theta, got) if rank == 0 else (None, 0))
    out['config4_strong'] = {
        'workload': 'BASELINE config 4, strong scaling: synthetic 1024x1024, 1 PointSource + 4 Sersic, %d walkers '
                    'resident on every rank, %d per rank per batch, one all-gather of the log-posteriors per batch '
                    '(parallel.ShardedLogPosterior.evaluate_device), fp64' % (n_w, block),
        'value': n_w / per_call, 'unit': 'evals/s', 'scaling': 'strong', 'timed_s': per_call * n_calls, 'calls': n_calls,
        'finite_logposts': int(np.isfinite(got).sum()), 'check_vs_cpu_rel': rel, 'walkers_checked': n_chk}
    model.close()
    del th, keep['out']
    # (2) the device sampler with every half-step's proposals sharded: 256 walkers on the headline field
    n_w = 256
    model, theta, fld = build_problem(args, local, size=256, sersic=1, walkers=n_w)
    p0 = synth_field.draw_walkers(256, 1, n_w, seed=41, near_truth=fld['truth'])
    sampler = DeviceEnsembleSampler(n_w, model, block=32, group=rg)
    sampler.random_state = np.random.RandomState(17).get_state()
    for _ in sampler.sample(p0, iterations=32):
        pass
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in sampler.sample(p0, iterations=64):
        pass
    per_iter = max((time.perf_counter() - t0) / 64, 1e-6)
    n_iter = decide(32 * int(np.ceil(1.5 / per_iter / 32)))          # >= 1 s of iterations, rank 0's count for all
    t0 = time.perf_counter()
    for _ in sampler.sample(p0, iterations=n_iter):
        pass
    dt = rank_max(time.perf_counter() - t0)
    out['sampler_w256'] = {
        'workload': 'stretch-move iterations of 256 walkers on the 256x256 headline field, walkers resident on the '
                    'device, every half-step\'s 128 proposals sharded over the ranks and all-gathered '
                    '(DeviceEnsembleSampler(group=...)); every rank holds the same chain',
        'stretch_move_iterations_per_s': n_iter / dt, 'evals_per_s': n_w * n_iter / dt, 'iterations': n_iter,
        'timed_s': dt, 'acceptance_fraction': float(np.mean(sampler.acceptance_fraction))}
    model.close()
    # (3) config 5: 64 independent 256^2 fields x 256 walkers, fields dealt to the ranks, no collective
    n_fields, n_w = args.config5_fields, 256
    lo, hi = shard_bounds(n_fields, world, rank)
    probs = [build_problem(args, local, seed=100 + f, size=256, sersic=1, walkers=n_w) for f in range(lo, hi)]
    mine = hi - lo
    fs = FieldSet([m for m, _, _ in probs], max_walkers=mine * n_w, device=local)
    all_theta = torch.from_numpy(np.concatenate([t for _, t, _ in probs])).to(dev)
    all_out = torch.empty(mine * n_w, dtype=torch.float64, device=dev)
    seg_f, seg_n = list(range(mine)), [n_w] * mine
    call = lambda: fs.context.logpost_theta_device(seg_f, seg_n, all_theta.data_ptr(), all_out.data_ptr(), None)
    if world > 1:
        dist.barrier()
    per_call, n_calls = timed_calls(call, sync, decide=decide)
    per_call = rank_max(per_call)
    got = all_out.cpu().numpy()
    rel = oracle_check(probs[0][0], probs[0][2], 1, probs[0][1], got[:n_w], count=2)[0] if rank == 0 else None
    finite = torch.tensor([float(np.isfinite(got).sum())], dtype=torch.float64, device='cpu' if gloo else dev)
    if world > 1:
        dist.all_reduce(finite, op=dist.ReduceOp.SUM)
    out['config5_fields'] = {
        'workload': 'BASELINE config 5: %d independent synthetic 256x256 fields x 256 walkers, %d fields per rank in '
                    'one context each (FieldSet), 1 PointSource + 1 Sersic, raw vectors resident in HBM -> '
                    'log-posterior, no data-path collective, fp64' % (n_fields, -(-n_fields // world)),
        'value': n_fields * n_w / per_call, 'unit': 'evals/s', 'scaling': 'strong', 'timed_s': per_call * n_calls,
        'calls': n_calls, 'finite_logposts': int(finite.item()), 'check_vs_cpu_rel': rel}
    fs.close()
    for m, _, _ in probs:
        m.close()
    return out if rank == 0 else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--fields-merge', type=int, default=1,
                    help='many-fields mode: 1 = all fields of a GPU in one context and one batch '
                         '(psfmc_ctx_create_fields), 0 = a context and a batch per field')
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--walkers', type=int, default=4096, help='walkers per batch (per GPU)')
    ap.add_argument('--batches', type=int, default=0,
                    help='batches per step (0: enough for ~250 ms per step, so that the default 20 '
                         'steps time >= 5 s: 84 at 256^2 x 4096 walkers)')
    ap.add_argument('--step-ms', type=float, default=250.0, help='target step duration of --batches 0')
    ap.add_argument('--distinct-batches', type=int, default=10,
                    help='sets of walker vectors resident in HBM; the batches of a step cycle through them')
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
    ap.add_argument('--init-timeout', type=float, default=180.0,
                    help='seconds a rank waits for the rendezvous / the first collective before it prints a '
                         'JSON error line and exits (N > 1)')
    ap.add_argument('--cpu-seconds', type=float, default=12.0)
    ap.add_argument('--no-example', action='store_true', help='skip the configs[1] (example model) rate')
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--no-extras', action='store_true',
                    help='skip the side figures (small ensembles, likelihood-only, host path)')
    ap.add_argument('--cpu-procs', type=int, default=-1,
                    help='processes of the all-cores CPU baseline (-1 = one per physical core, 0 = skip)')
    ap.add_argument('--cpu-worker', type=int, default=-1, help=argparse.SUPPRESS)
    ap.add_argument('--no-configs', action='store_true', help='skip the other single-GPU workloads (N = 1)')
    ap.add_argument('--no-multi', action='store_true', help="skip the sharded paths' side figures (multi_gpu)")
    ap.add_argument('--config4-walkers', type=int, default=2048, help='walkers of the strong-scaled config 4')
    ap.add_argument('--config5-fields', type=int, default=64, help='fields of config 5, dealt to the ranks')
    ap.add_argument('--extras-timeout', type=float, default=420.0,
                    help='N > 1: seconds the multi_gpu side figures may take before rank 0 prints the headline line '
                         'without them and every rank leaves')
    args = ap.parse_args()
    if args.cpu_worker >= 0:
        return cpu_worker(args)
    env_world = os.environ.get('WORLD_SIZE')
    if env_world is None and args.gpus > 1:
        sys.exit(spawn_ranks(args, sys.argv[1:]))
    world = int(env_world or '1')
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d (launch with --nproc-per-node %d, or '
                         'run plain `python bench.py --gpus N`)' % (args.gpus, world, args.gpus))
    if args.batches <= 0:
        # ~3.3 ms per 4096-walker batch at 256^2 on one MI355X; scale by the bytes of a walker
        per_batch_ms = 3.0 * (args.walkers / 4096.0) * (args.size / 256.0) ** 2 * max(args.fields, 1)
        args.batches = max(1, min(512, int(np.ceil(args.step_ms / per_batch_ms))))
    # The GPU timing comes FIRST (the driver samples GPU utilisation while the command runs); the
    # CPU baselines follow, the all-cores one in child processes (started with fork + exec: this
    # process itself is never replaced)
    n_usable, cpu_detail = usable_cores()
    if args.cpu_procs < 0:
        args.cpu_procs = n_usable

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (there is no CPU fallback)')
    gloo = world > 1 and args.dist_backend == 'gloo'
    if gloo:                                 # rehearsal: several ranks may share one device
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1:
        import datetime
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dog = Watchdog(rank, world)
        try:
            dog.phase('init_process_group(%s)' % args.dist_backend, args.init_timeout)
            limit = datetime.timedelta(seconds=args.init_timeout)
            if gloo:
                dist.init_process_group('gloo', timeout=limit)
            else:
                dist.init_process_group('nccl', device_id=dev, timeout=limit)
            # the first collective builds the RCCL communicator (xGMI / dmabuf IPC set-up happens here)
            dog.phase('first collective (all_gather of one double per rank)', args.init_timeout)
            probe = torch.full((1,), float(rank), dtype=torch.float64, device='cpu' if gloo else dev)
            got = torch.empty(world, dtype=torch.float64, device=probe.device)
            dist.all_gather_into_tensor(got, probe)
            if not gloo:
                torch.cuda.synchronize(dev)
            ranks_seen = [int(v) for v in got.cpu().tolist()]
            if ranks_seen != list(range(world)):
                raise RuntimeError('all_gather returned %r' % (got.cpu().tolist(),))
        except Exception as exc:           # noqa: BLE001 -- report and leave, whatever it was
            dog.failed(exc)
        dog.done()
    else:
        ranks_seen = [0]

    if args.fields > 1:
        return many_fields(args, torch, dist, world, rank, local, dev, gloo)
    # N ranks: a batch is N x --walkers vectors, the SAME on every rank and resident in every rank's HBM; a rank
    # evaluates its contiguous block of --walkers and the blocks are all-gathered (weak scaling)
    w_all = args.walkers * world
    model, theta, fld = build_problem(args, local, walkers=w_all, max_walkers=args.walkers)
    eng = model.engine
    if args.chunk:
        eng.set_option('chunk_walkers', args.chunk)
    for kv in args.opt:
        key, val = kv.split('=')
        eng.set_option(key, float(val))
    # the batches of a step cycle through n_sets sets of walkers of their own (same field); all resident in HBM
    n_sets = max(1, min(args.batches, args.distinct_batches))
    thetas = [theta] + [draw_theta(args, fld, w_all, seed=11 + 2 * b) for b in range(1, n_sets)]
    theta_dev = torch.from_numpy(np.ascontiguousarray(np.stack(thetas))).to(dev)      # [n_sets, N W, P]
    out = torch.empty((n_sets, args.walkers), dtype=torch.float64, device=dev)
    sharded, gathered = None, {}
    if world > 1:
        # the product's sharded evaluator (psfmc_amd/parallel.py): this rank's block through
        # psfmc_eval_theta_device, ONE all-gather of the log-posteriors per batch (RCCL with 'nccl'; host-staged
        # with the 'gloo' rehearsal back end), every rank ends with all N W values
        from psfmc_amd.parallel import RankGroup, ShardedLogPosterior
        rgroup = RankGroup(None, dev)
        sharded = ShardedLogPosterior(model, group=rgroup, device=dev)
        stream_cm = rgroup.on_stream()
        stream_cm.__enter__()                # torch's current stream = the sharded path's stream from here on
        stream = torch.cuda.current_stream(dev)
    else:
        # a real (non-NULL) stream: the library launches on the stream it is handed,
        # and the HIP events below must sit on that same stream
        stream = torch.cuda.Stream(dev)
        torch.cuda.set_stream(stream)
    sptr = stream.cuda_stream

    def one_batch(b=0):
        eng.logpost_theta_device(args.walkers, theta_dev[b].data_ptr(), 0, out[b].data_ptr(), sptr)

    def step():
        for i in range(args.batches):
            b = i % n_sets
            if sharded is None:
                one_batch(b)
            else:
                gathered[b] = sharded.evaluate_device(theta_dev[b])

    elapsed = timed_region(args, torch, dist, world, dev, step, gloo)
    sharded_check = None
    if world > 1:
        # every rank holds every block: this rank's own block evaluated alone must be what the gather put in
        # its place, and (rank 0) 64 walkers of the LAST rank's block evaluated here must equal what that rank sent
        # (per-walker results do not depend on the batch: same bits)
        full = gathered[0]
        lo = rank * args.walkers
        eng.logpost_theta_device(args.walkers, theta_dev[0][lo:lo + args.walkers].data_ptr(), 0, out[0].data_ptr(), sptr)
        far = (world - 1) * args.walkers
        probe_out = torch.empty(64, dtype=torch.float64, device=dev)
        eng.logpost_theta_device(64, theta_dev[0][far:far + 64].data_ptr(), 0, probe_out.data_ptr(), sptr)
        torch.cuda.synchronize(dev)
        ok = bool(torch.equal(out[0], full[lo:lo + args.walkers]) and torch.equal(probe_out, full[far:far + 64]))
        flag = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device='cpu' if gloo else dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        sharded_check = bool(flag.item() == 1.0)
        assert sharded_check, 'a rank found the gathered log-posteriors different from its own evaluation'

    # sanity: the batch the timing ran on is numerically right
    lnpost = out[0].cpu().numpy()
    if world == 1:
        n_finite = int(np.isfinite(out.cpu().numpy()).sum())
    else:
        n_finite = int(sum(int(torch.isfinite(g).sum().item()) for g in gathered.values()))

    kernels = []
    if rank == 0 and args.backend == 'fused':
        kernels = kernel_profile(eng, args, one_batch, torch, dev, max(2, min(args.steps, 5)))
    line = None
    if rank == 0:
        evals_per_step = args.walkers * args.batches
        total_evals = evals_per_step * world * args.steps
        value = total_evals / elapsed
        rate_gpu = value / world
        b_eval = algorithmic_bytes_per_eval(args.size)
        table = pmc_table(args)
        line = {
            'metric': 'walker log-posterior evals/sec, 256x256 image, 1 PSF+1 Sersic',
            'value': value, 'unit': 'evals/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'synthetic %dx%d field, 1 PointSource + %d Sersic, %d batches x %d walkers '
                                   'per GPU per step (cycling through %d resident sets of walkers), raw emcee '
                                   'vectors resident in HBM -> full log-posterior (priors, early-out, Sersic '
                                   'constants, likelihood), fp64'
                                   % (args.size, args.size, args.sersic, args.batches, args.walkers, n_sets),
                       'image': args.size, 'walkers_per_batch': args.walkers, 'batches_per_step': args.batches,
                       'value_is': 'device-resident vectors (psfmc_eval_theta_device); the Python call '
                                   'log_posterior_batch(theta) with HOST vectors is python_entry_point below',
                       'evals_per_gpu_per_step': evals_per_step,
                       'entry_point': ('psfmc_eval_theta_device' if world == 1 else
                                       'psfmc_amd.parallel.ShardedLogPosterior.evaluate_device: a batch = %d vectors '
                                       'resident on every rank, each rank evaluates its block of %d '
                                       '(psfmc_eval_theta_device), one all-gather of the log-posteriors per batch '
                                       '(RankGroup.all_gather_blocks)' % (w_all, args.walkers)),
                       'backend': args.backend, 'parallelism': 'walkers sharded x%d' % world},
            'finite_logposts': n_finite, 'ranks_seen': ranks_seen,
            'dist_backend': args.dist_backend if world > 1 else None,
            'gathered_equals_own_evaluation_bitwise': sharded_check,
        }
        per_walker = [pmc_lookup(table, k) for k in ('k_rows_fwd<', 'k_cols', 'k_rows_inv<')]
        measured = sum(per_walker) if all(per_walker) else None
        step_roof = {'bound': 'hbm', 'achieved': measured * rate_gpu / 1e9 if measured else None,
                     'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                     'frac': measured * rate_gpu / 1e9 / HBM_PEAK_GBPS if measured else None,
                     'traffic': measured * evals_per_step if measured else None,
                     'kernel': 'whole step per GPU: (k_rows_fwd + k_cols + k_rows_inv) PMC bytes per walker '
                               'x evals/s; profiles/pmc_traffic.json',
                     'measured_bytes_per_eval': measured,
                     'algorithmic_equiv_GBps': b_eval * rate_gpu / 1e9,
                     'algorithmic_bytes_per_eval': b_eval, 'bytes_note': BYTES_NOTE}
        if kernels:
            # dominant kernel: its own algorithmic bytes (= the bytes the design moves: it reads
            # and writes the transposed half-spectra once; the column pass's 64 F of SURVEY section
            # 8(d)'s figure) over its event-timed average launch duration
            k = kernels[0]
            traffic = pmc_lookup(table, k['kernel'].split('<')[0])
            prof_wpl = next((rec.get('walkers_per_launch') for name, rec in table.items()
                             if name.startswith(k['kernel'].split('<')[0])), None)
            line['roofline'] = {
                'bound': 'hbm', 'achieved': k['GBps'], 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                'frac': k['GBps'] / HBM_PEAK_GBPS,
                'traffic': traffic * k['walkers_per_launch'] if traffic else None,
                'kernel': k['kernel'], 'launch_ms': k['avg_ms'], 'launches': k['launches'],
                'bytes_per_launch': k['bytes_per_walker'] * k['walkers_per_launch'],
                'walkers_per_launch': k['walkers_per_launch'],
                'timing': 'HIP events around every launch on its own stream, one pass in flight',
                'traffic_source': 'profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of '
                                  'this command at %s walkers per launch (this run: %.1f), 2 x FETCH_SIZE + '
                                  'WRITE_SIZE per launch' % (prof_wpl, k['walkers_per_launch']),
                'bytes_note': BYTES_NOTE}
            line['roofline_step'] = step_roof
            line['kernels'] = kernels
        else:
            line['roofline'] = step_roof
    # the sharded paths' side figures: every rank takes part (rank 0 comes from its profile pass).  With N > 1 they
    # run under a deadline that prints the headline line as it stands if they hang
    multi = None
    if world > 1:
        stream_cm.__exit__(None, None, None)
        dist.barrier()
    if not args.no_multi and not args.no_extras and args.backend == 'fused':
        t0 = time.perf_counter()
        salvage = None
        if world > 1:
            salvage = Salvage(rank, args.extras_timeout)
            salvage.arm(line)
        multi = guarded('multi_gpu', multi_gpu_extras, args, torch, dist, world, rank, local, dev, gloo)
        if salvage is not None:
            salvage.done()
        if multi is not None:
            multi['wall_s'] = time.perf_counter() - t0
    if rank == 0:
        if world == 1 and not args.no_extras:
            # side figures: likelihood only from pre-derived rows (round 1's headline), the
            # Python entry point with host vectors, small ensembles
            rows = torch.from_numpy(model.derived_rows(theta)).to(dev)
            like = torch.empty(args.walkers, dtype=torch.float64, device=dev)
            for _ in range(2):
                eng.loglike_device(args.walkers, rows.data_ptr(), 0, like.data_ptr(), sptr)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(10):
                eng.loglike_device(args.walkers, rows.data_ptr(), 0, like.data_ptr(), sptr)
            torch.cuda.synchronize(dev)
            line['loglike_from_rows_evals_per_s'] = args.walkers * 10 / (time.perf_counter() - t0)
            model.log_posterior_batch(theta)
            n_calls, t0 = 0, time.perf_counter()
            while n_calls < 5 or time.perf_counter() - t0 < 1.0:
                model.log_posterior_batch(theta)
                n_calls += 1
            host_rate = args.walkers * n_calls / (time.perf_counter() - t0)
            line['host_path_evals_per_s'] = host_rate
            line['python_entry_point'] = {
                'evals_per_s': host_rate, 'calls': n_calls, 'walkers_per_call': args.walkers,
                'what': 'SURVEY.md section 8(d) literally: MultiComponentModel.log_posterior_batch(theta) with '
                        'HOST numpy vectors in and HOST log-posteriors out (host-to-device copy, launch, '
                        'device-to-host copy and synchronisation inside every call)'}
            line['small_ensembles'] = small_ensembles(eng, args, torch, dev, theta_dev[0], out[0], stream)
            if args.backend == 'fused' and not model._host_priors:
                line['default_ensemble_sampler'] = default_ensemble_sampler(model, theta)
            if args.backend == 'fused':
                per_pass = eng.pass_size(args.walkers)
                t_bytes = 2 * (args.size // 2 + 1) * args.size * 16
                from psfmc_amd import engine as engine_mod
                line['sweep_ceiling'] = sweep_ceiling(engine_mod, local, t_bytes * per_pass, per_pass, rate_gpu)
                n_simd = 4 * torch.cuda.get_device_properties(dev).multi_processor_count
                vc = valu_ceiling(engine_mod, local, table, per_pass, rate_gpu, n_simd)
                if vc:
                    vc['step_over_larger_floor'] = vc['step_us_per_pass'] / max(vc['pass_floor_us'],
                                                                               line['sweep_ceiling']['pass_floor_us'])
                    line['valu_ceiling'] = vc
            if args.backend == 'fused' and args.size in (64, 128, 256, 512, 1024):
                # opt-in storage mode, NOT the headline: complex64 half-spectra between the kernels,
                # fp64 arithmetic (include/psfmc_hip.h "storage_f32")
                ref64 = out[0].cpu().numpy()
                eng.set_option('storage_f32', 1)
                for _ in range(2):
                    one_batch()
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for _ in range(10):
                    one_batch()
                torch.cuda.synchronize(dev)
                rate = args.walkers * 10 / (time.perf_counter() - t0)
                got32 = out[0].cpu().numpy()
                fin = np.isfinite(ref64)
                line['storage_f32_option'] = {
                    'evals_per_s': rate, 'max_rel_diff_vs_f64': float(np.max(np.abs(got32[fin] - ref64[fin]) /
                                                                             np.abs(ref64[fin]))),
                    'max_abs_diff_vs_f64': float(np.max(np.abs(got32[fin] - ref64[fin]))),
                    'note': 'opt-in: complex64 storage of the intermediate half-spectra, fp64 arithmetic; '
                            'not an fp64 result and not the headline'}
                eng.set_option('storage_f32', 0)
                one_batch()
                torch.cuda.synchronize(dev)
        if multi is not None:
            line['multi_gpu'] = multi
        if world == 1 and not args.no_configs and not args.no_extras and args.backend == 'fused':
            t0 = time.perf_counter()
            line['configs'] = guarded('configs', other_configs, args, torch, dev, local)
            line['configs']['wall_s'] = time.perf_counter() - t0
        if world == 1 and not args.no_example:
            ex = guarded('example_model_256_walkers', example_model_rate)
            if ex:
                line['example_model_256_walkers'] = ex
        if not args.no_cpu and world == 1:      # CPU baseline: rank 0 at N = 1 only
            base, vals = cpu_baseline(args, fld, theta, args.cpu_seconds)
            line['cpu_baseline'] = base
            if args.cpu_procs > 1:
                line['cpu_baseline_all_cores'] = cpu_baseline_multi(args, args.cpu_procs, cpu_detail, base['value'])
            if args.size == 256 and args.sersic == 1:
                line['reference_cpu_survey'] = {
                    'value': 79.0, 'unit': 'evals/s', 'cores': 1, 'kind': 'reference',
                    'note': 'a labelled constant, not measured in this run: the unmodified reference '
                            '(mmechtley/psfMC, numpy 1.26 + numexpr, one process) on this workload in the '
                            'survey container, BASELINE.md section 2 / SURVEY.md section 6; the reference '
                            'cannot travel to the GPU box'}
            ref = np.array(vals)[:len(lnpost)]
            ref = ref + model.log_priors_batch(theta[:len(ref)])
            got = lnpost[:len(ref)]
            fin = np.isfinite(ref)
            assert np.array_equal(np.isfinite(got), fin), 'GPU and oracle disagree on finiteness'
            line['check_vs_cpu_rel'] = float(np.max(np.abs(got[fin] - ref[fin]) / np.abs(ref[fin])))
        print(json.dumps(line))
    if world > 1 and isinstance(multi, dict) and 'error' in multi:
        # a side figure failed on this rank: the other ranks may sit in one of its collectives until their own
        # deadline; the line is out, leave without another collective
        sys.stdout.flush()
        os._exit(0)
    model.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
