/*
 * psfmc_hip.h -- C ABI of libpsfmc_hip.so: the MI355X (gfx950) implementation of
 * psfMC's per-sample log-likelihood, batched over ensemble-sampler walkers.
 *
 * Drop-in boundary.  The reference evaluates one walker at a time inside
 *   MultiComponentModel.log_posterior          psfMC/models.py:193-243
 * which emcee reaches through `lnpostfn` / `pool.map` (psfMC/fitting.py:56-58).
 * Everything below the prior early-out (models.py:213-241) is replaced by
 * psfmc_eval_batch(); the one-time setup that feeds it (Configuration.py:41-52,
 * PSFSelector.py:32-43, utils.py:9-22 `pad_and_rfft_image`, :126-133
 * `pre_fft_psf`) is replaced by psfmc_ctx_create().  The Python host side
 * (psfmc_amd/engine.py) binds these entry points with ctypes; see INTEGRATION.md
 * for the stub a maintainer of the reference would add.
 *
 * Conventions
 *   - plain pointers and sizes only; caller owns every host buffer for the
 *     duration of the call; the context owns all device memory.
 *   - return 0 on success, a negative PSFMC_E* code otherwise, message in
 *     psfmc_last_error().  Numeric trouble is never an error: a NaN / inf
 *     log-likelihood is written to the output and the host maps it to -inf
 *     (psfMC/models.py:238-241).
 *   - a context is bound to one device and is not thread-safe.
 *   - images are row-major [ny][nx], x = column index = fastest axis, pixel
 *     centres at integer coordinates (psfMC/utils.py:35-42).  ny and nx must
 *     be even (psfMC/models.py:276 is broken for odd sizes too).
 */
#ifndef PSFMC_HIP_H
#define PSFMC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct psfmc_ctx psfmc_ctx;

/* error codes */
#define PSFMC_OK            0
#define PSFMC_EINVAL       -1   /* bad argument / unsupported shape        */
#define PSFMC_EHIP         -2   /* HIP runtime / hipFFT failure            */
#define PSFMC_ENOMEM       -3
#define PSFMC_ENODEV       -4   /* no usable gfx950 device                 */

/* convolution back ends (both run entirely on the GPU) */
#define PSFMC_BACKEND_FUSED   0  /* hand-written LDS FFT fused with rasteriser / spectral multiply / chi^2.
                                    BUILT sides: the powers of two 64..1024 and the even 5-smooth sides 96 100 120
                                    144 150 160 180 192 200 240 250 288 300 320 360 384 400 480 500 576 600
                                    640 720 768 800 900 960 and, with a factor 7, 84 98 112 126 140 168 196
                                    210 224 252 280 294 336 350 392 420 448 504 560 630 672 700 784 840 896,
                                    with a factor 11 or 13: 88 104 110 130 132 156 176 208 220 260 264 286 308 312
                                    330 352 364 390 416 440 484 520 528 572 616 624 650 660 676 704 728 780 832
                                    (nx and ny independently, any combination).  Any OTHER even side is EMBEDDED
                                    (round 3): it runs on the kernels of a built side >= side + PSF side - 1 of
                                    that axis (overlap-save; every array at this boundary keeps the image's own
                                    shape).  psfmc_ctx_create returns PSFMC_EINVAL only for odd sides and for an
                                    unbuilt side with side + PSF side - 1 above the largest built side */
#define PSFMC_BACKEND_HIPFFT  1  /* batched hipFFT D2Z/Z2D between separate kernels: any even shape
                                    (psfMC/utils.py:25-32 accepts those); also the cross-check path */

/* point-source shift methods (psfMC/ModelComponents/PointSource.py:40-51) */
#define PSFMC_PS_LANCZOS3   0
#define PSFMC_PS_BILINEAR   1

/*
 * Per-walker derived-parameter row (doubles), length psfmc_row_len():
 *   [0]                      sky level, ADU                      (Sky.py:14-16)
 *   per point source  (4):   flux, x0, y0, method                (PointSource.py:24-57; flux = utils.py:160-164)
 *   per Sersic        (9):   x0, y0, m00, m01, m10, m11, kappa, p, sb_eff
 *                            (M = inverse scale * inverse rotation, Sersic.py:80-91;
 *                             kappa = gammaincinv(2n, 1/2), Sersic.py:47-53; p = 0.5/n,
 *                             Sersic.py:121; sb_eff Sersic.py:55-71)
 *   [last]                   psf index (already rounded; PSFSelector.py:54-66)
 * Component groups are summed in the order sky, point sources, Sersics.
 */
#define PSFMC_ROW_SKY      1
#define PSFMC_ROW_PS       4
#define PSFMC_ROW_SERSIC   9

/*
 * Create a context for one observed field.
 *   sci, obs_var, bad_px  [ny][nx]  Configuration.obs_data / obs_var / bad_px
 *                                   (Configuration.py:44-46; obs_var = +inf at bad pixels,
 *                                   utils.py:68-70; bad_px nonzero = excluded from the sum)
 *   psf, psf_var          [n_psf][psf_ny][psf_nx]  normalised PSFs and their variance maps
 *                                   as returned by preprocess_psf / calculate_psf_variability
 *                                   (utils.py:106-157).  They are centre-padded to [ny][nx]
 *                                   at offset pad/2 and Fourier transformed ON THE DEVICE
 *                                   (replaces utils.py:9-22, :126-133).
 *   n_ps, n_sersic        component counts of the model (fixed per context)
 *   max_walkers           largest W a later call may pass
 */
int psfmc_ctx_create(psfmc_ctx** out, int device, int ny, int nx,
                     const double* sci, const double* obs_var, const uint8_t* bad_px,
                     int n_psf, int psf_ny, int psf_nx,
                     const double* psf, const double* psf_var,
                     int n_ps, int n_sersic, int max_walkers, int backend);

int psfmc_ctx_destroy(psfmc_ctx* ctx);

/* doubles per walker row: 2 + 4*n_ps + 9*n_sersic */
int psfmc_row_len(const psfmc_ctx* ctx);

/*
 * Walkers per internal pass the library uses for a batch of W walkers (the batch is
 * evaluated as ceil(W / pass) passes of at most this size; per-walker results do not depend
 * on it).  For tests that place known vectors at pass boundaries and for profiling scripts
 * that normalise per-launch counters; there is no reference counterpart.  < 0 on error.
 */
int psfmc_pass_size(const psfmc_ctx* ctx, int W);

/*
 * Log-likelihood of W walkers (models.py:213-216, 233-236 for each):
 *   loglike[w] = -0.5 * sum_{good px} ( resid^2 * ivm - ln(ivm / 2pi) )
 * rows [W][row_len] and skip [W] (nonzero = prior was not finite: the walker is
 * not evaluated and gets -inf, models.py:208-211; may be NULL) are HOST buffers;
 * the call returns after loglike[W] has been copied back.
 */
int psfmc_eval_batch(psfmc_ctx* ctx, int W, const double* rows,
                     const uint8_t* skip, double* loglike);

/*
 * Same, with DEVICE pointers, enqueued on `stream` (a hipStream_t, NULL = the
 * context's own stream) and not synchronised: for callers that keep walkers
 * resident in HBM (bench.py, the multi-GPU all-gather path).
 */
int psfmc_eval_batch_device(psfmc_ctx* ctx, int W, const double* d_rows,
                            const uint8_t* d_skip, double* d_loglike, void* stream);

/*
 * The five per-sample images of models.py:222-226 for W walkers, each
 * [W][ny][nx] host output or NULL: raw_model (models.py:245-253),
 * convolved_model (:255-263), residual (:282-294), composite_ivm (:265-280),
 * point_source_subtracted (:296-306).
 */
int psfmc_eval_images(psfmc_ctx* ctx, int W, const double* rows,
                      double* raw, double* conv, double* resid, double* ivm,
                      double* ps_sub);

/*
 * Raw emcee parameter vectors on the device: with the model's parameter layout and
 * its priors registered, psfmc_eval_theta computes the complete log-posterior
 * (models.py:193-243) of W vectors without any host arithmetic -- joint log-prior
 * (distributions.py:112-127 for the families below, Sersic.py:41-45), early-out,
 * Sersic kappa = gammaincinv(2n, 1/2) and Sigma_e (Sersic.py:47-71), flux
 * (utils.py:160-164), ellipse matrix (Sersic.py:80-91), likelihood, NaN -> -inf.
 *
 * Slots, in this order: n_sky x [adu] | n_ps x [mag, x, y] | n_sersic x [angle, index,
 * mag, reff, reff_b, x, y] | [psf_index].  slot_col[i] is the column of the vector that
 * feeds slot i (packing contract of ComponentBase.py:45-74 / models.py:174-185) or -1
 * for a constant slot_const[i].
 * Prior families per vector column: 0 = evaluated by the host (passed per walker in
 * `extra_lnprior`), 1 uniform(loc=p0, scale=p1), 2 normal(loc=p0, scale=p1),
 * 3 weibull_min(c=p0, loc=p1, scale=p2), 4 randint(low=p0, high=p1) on the rounded value.
 */
#define PSFMC_PRIOR_HOST         0
#define PSFMC_PRIOR_UNIFORM      1
#define PSFMC_PRIOR_NORMAL       2
#define PSFMC_PRIOR_WEIBULL_MIN  3
#define PSFMC_PRIOR_RANDINT      4
int psfmc_set_layout(psfmc_ctx* ctx, int n_sky, int n_params, const int* slot_col,
                     const double* slot_const, const int* ps_method, const int* sersic_degrees,
                     double mag_zeropoint, const int* family, const double* p0, const double* p1,
                     const double* p2);
/* host buffers theta [W][n_params], extra_lnprior [W] or NULL, lnprob [W] */
int psfmc_eval_theta(psfmc_ctx* ctx, int W, const double* theta, const double* extra_lnprior,
                     double* lnprob);
/* device buffers, enqueued on `stream` (NULL = the context's stream), not synchronised */
int psfmc_eval_theta_device(psfmc_ctx* ctx, int W, const double* d_theta, const double* d_extra_lnprior,
                            double* d_lnprob, void* stream);

/*
 * Several observed fields of ONE shape in one context (fused back end).  The reference fits one
 * field per process (psfMC/fitting.py:13-113 builds one MultiComponentModel per model file); a
 * survey of many small fields -- BASELINE config 5: independent 256 x 256 fields x 256 walkers each
 * -- then makes many small batches, each paying the fixed cost of a call.  Here the fields' walkers
 * share the batches: every walker's record carries (field * n_psf + PSF) as its kernel-spectrum index,
 * the row kernels pick the field's pixels from it, and 8 x 256 walkers run at the rate of one 2048-walker
 * ensemble.
 *   sci / obs_var / bad_px  [n_fields][ny][nx];  psf / psf_var  [n_fields][n_psf][psf_ny][psf_nx]
 *   the same component counts (n_ps, n_sersic) and parameter layout structure for every field;
 *   psfmc_set_layout (= field 0) first, then psfmc_set_layout_field for fields 1..: own constants, priors
 *   psfmc_eval_theta[_device]_fields: segment i = seg_count[i] consecutive walkers of field seg_field[i];
 *   theta [W][n_params], lnprob [W] in segment order, W = sum of the counts <= max_walkers.
 * Round 3: a context of several fields is also FITTED as one (psfMC/fitting.py:56-113 for every field at once):
 *   psfmc_stretch_run_fields      every field's ensemble (W walkers each, own random numbers) sampled together,
 *                                 the half-step proposals of all fields in one batch of n_fields W / 2 walkers;
 *                                 arrays as psfmc_stretch_run's with the field as the leading dimension
 *   psfmc_accumulate_theta_field  posterior-image sums of ONE field from raw vectors (analysis/images.py:62-74)
 *   psfmc_get_accumulated_field   that field's five posterior images (models.py:74-97) and sample count
 *   psfmc_eval_images_field       the five per-sample images (models.py:222-226) of walkers of one field
 *   psfmc_eval_batch_field        psfmc_eval_batch for derived rows of one field (round 4: models.py:213-216,
 *                                 233-236 on caller-derived scalars; the rows' PSF index counts within the field)
 * psfmc_reset_accumulated clears every field's sums, psfmc_reset_accumulated_field one field's.  The raw-sum exchange for sharded ranks
 * (psfmc_get/set_accumulated_sums) and the half-step API (psfmc_stretch_open ...) serve one-field contexts.
 */
int psfmc_ctx_create_fields(psfmc_ctx** out, int device, int ny, int nx, int n_fields, const double* sci,
                            const double* obs_var, const uint8_t* bad_px, int n_psf, int psf_ny, int psf_nx,
                            const double* psf, const double* psf_var, int n_ps, int n_sersic, int max_walkers);
int psfmc_set_layout_field(psfmc_ctx* ctx, int field, int n_sky, int n_params, const int* slot_col,
                           const double* slot_const, const int* ps_method, const int* sersic_degrees,
                           double mag_zeropoint, const int* family, const double* p0, const double* p1,
                           const double* p2);
int psfmc_stretch_run_fields(psfmc_ctx* ctx, int W, int n_iter, double* pos, double* lnprob, int lnprob_valid,
                             const double* z, const double* lz, const int* partner, const double* log_u,
                             double* chain, double* lnprob_chain, long long* naccepted, int accumulate);
int psfmc_accumulate_theta_field(psfmc_ctx* ctx, int field, int W, const double* theta);
int psfmc_reset_accumulated_field(psfmc_ctx* ctx, int field);
int psfmc_get_accumulated_field(psfmc_ctx* ctx, int field, double* raw, double* conv, double* resid, double* ivm,
                                double* ps_sub, long long* count);
int psfmc_eval_images_field(psfmc_ctx* ctx, int field, int W, const double* rows, double* raw, double* conv,
                            double* resid, double* ivm, double* ps_sub);
int psfmc_eval_batch_field(psfmc_ctx* ctx, int field, int W, const double* rows, const uint8_t* skip,
                           double* loglike);
int psfmc_eval_theta_fields(psfmc_ctx* ctx, int n_seg, const int* seg_field, const int* seg_count,
                            const double* theta, const double* extra_lnprior, double* lnprob);
int psfmc_eval_theta_device_fields(psfmc_ctx* ctx, int n_seg, const int* seg_field, const int* seg_count,
                                   const double* d_theta, const double* d_extra_lnprior, double* d_lnprob,
                                   void* stream);
/* test hook: the derived rows [W][row_len], log-priors [W] and skip flags [W] the device
 * computes for W vectors */
int psfmc_debug_theta_rows(psfmc_ctx* ctx, int W, const double* theta, double* rows, double* lnprior,
                           uint8_t* skip);

/*
 * Stretch-move ensemble sampling with the walkers resident on the device: n_iter
 * iterations of the two half-ensemble proposals of emcee 2.2.1's EnsembleSampler
 * (the sampler the reference drives, psfMC/fitting.py:56-86; algorithm restated in
 * SURVEY.md Appendix A).  The caller supplies the random numbers in emcee's draw order
 * (per half-step: z, partner index, ln u), so a run reproduces the host-side sampler;
 * nothing is copied to the host between iterations.  Needs psfmc_set_layout with every
 * prior on the device (no PSFMC_PRIOR_HOST column).
 *   pos [W][P], lnprob [W]      in/out (host); lnprob is computed first if !lnprob_valid
 *   lz, log_u [n_iter][2][W/2]  (P-1) ln z and ln u;  z [n_iter][2][W/2]
 *   partner [n_iter][2][W/2]    index into the complementary half-ensemble
 *   chain [W][n_iter][P], lnprob_chain [W][n_iter]   outputs (host, may be NULL)
 *   naccepted [W]               in/out acceptance counters
 *   accumulate                  nonzero: after every iteration add the images of all W
 *                               positions to the posterior sums (psfmc_accumulate_images)
 */
int psfmc_stretch_run(psfmc_ctx* ctx, int W, int n_iter, double* pos, double* lnprob,
                      int lnprob_valid, const double* z, const double* lz, const int* partner,
                      const double* log_u, double* chain, double* lnprob_chain,
                      long long* naccepted, int accumulate);

/*
 * The same sampler one half-step at a time, for walkers sharded over several GPUs (one
 * process per GPU).  Every rank opens the SAME ensemble with the SAME
 * random numbers; per half-step each rank calls psfmc_stretch_half_eval for its contiguous
 * block [lo, lo + n) of the half-ensemble's proposals (the proposals, priors and prep records
 * of the whole half are formed on every rank; only the block's likelihood pipeline runs), the
 * caller all-gathers the blocks' log-posteriors (torch.distributed / RCCL: the one collective
 * of the path) and hands the gathered half-ensemble vector to psfmc_stretch_half_accept, which
 * applies the identical accept / move / chain entry on every rank.  There is no reference
 * counterpart (psfMC/fitting.py:55 gave up on parallel evaluation); the result equals
 * psfmc_stretch_run's chain bit for bit.  d_* are device pointers, `stream` a hipStream_t
 * (NULL = the context's stream); nothing is synchronised except in open / close.
 *   psfmc_stretch_accumulate   add the images of walkers [lo, lo + n) of the current
 *                              ensemble to this rank's posterior sums
 *   psfmc_get/set_accumulated_sums   the raw sums [4][ny][nx] (raw, convolved, model
 *                              variance, PS-only convolved) + sample count, so that ranks'
 *                              shares can be added up (all-reduce) before
 *                              psfmc_get_accumulated turns them into means
 */
int psfmc_stretch_open(psfmc_ctx* ctx, int W, int n_iter, const double* pos, const double* lnprob,
                       const double* z, const double* lz, const int* partner, const double* log_u,
                       const long long* naccepted, int store_chain);
int psfmc_stretch_half_eval(psfmc_ctx* ctx, int it, int h, int lo, int n, double* d_newlnp_block,
                            void* stream);
int psfmc_stretch_half_accept(psfmc_ctx* ctx, int it, int h, const double* d_newlnp_half, void* stream);
int psfmc_stretch_accumulate(psfmc_ctx* ctx, int lo, int n, void* stream);
int psfmc_stretch_close(psfmc_ctx* ctx, double* pos, double* lnprob, double* chain,
                        double* lnprob_chain, long long* naccepted, void* stream);
int psfmc_get_accumulated_sums(psfmc_ctx* ctx, double* sums, long long* count);
int psfmc_set_accumulated_sums(psfmc_ctx* ctx, const double* sums, long long count);

/*
 * One process driving several GPUs (the form SURVEY.md section 8(b) sketched; the
 * one-process-per-GPU form with torch.distributed / RCCL is psfmc_amd/parallel.py).  The
 * field is replicated on every listed device (a device may be listed more than once),
 * walkers are split into contiguous blocks (block r = walkers [r W/n ...)), every device's
 * upload, evaluation and download are enqueued before any is waited for, and each block
 * lands at its offset of the caller's host array.  max_walkers bounds W of the whole
 * group.  Results equal the single-context ones bit for bit.  No reference counterpart
 * (psfMC/fitting.py:55).
 */
typedef struct psfmc_group psfmc_group;
int psfmc_group_create(psfmc_group** out, int n_dev, const int* devices, int ny, int nx,
                       const double* sci, const double* obs_var, const uint8_t* bad_px,
                       int n_psf, int psf_ny, int psf_nx, const double* psf, const double* psf_var,
                       int n_ps, int n_sersic, int max_walkers, int backend);
int psfmc_group_destroy(psfmc_group* group);
int psfmc_group_size(const psfmc_group* group);
int psfmc_group_set_layout(psfmc_group* group, int n_sky, int n_params, const int* slot_col,
                           const double* slot_const, const int* ps_method, const int* sersic_degrees,
                           double mag_zeropoint, const int* family, const double* p0, const double* p1,
                           const double* p2);
int psfmc_group_eval_batch(psfmc_group* group, int W, const double* rows, const uint8_t* skip,
                           double* loglike);
int psfmc_group_eval_theta(psfmc_group* group, int W, const double* theta, const double* extra_lnprior,
                           double* lnprob);

/*
 * Posterior-image accumulation on the device (replaces the per-sample blob
 * hand-over and the running mean of MultiComponentModel.accumulate_images,
 * models.py:74-97, fed from fitting.py:83).  psfmc_accumulate_images adds the images
 * of W walkers (rows as above) to device-resident sums; composite_ivm is averaged as
 * a variance (models.py:81-97).  psfmc_get_accumulated returns the means ([ny][nx]
 * each, any pointer may be NULL; ivm = 1 / mean variance) and the sample count.
 */
int psfmc_accumulate_images(psfmc_ctx* ctx, int W, const double* rows);
/* the same for W raw parameter vectors [W][n_params] (psfmc_set_layout): records derived on the device */
int psfmc_accumulate_theta(psfmc_ctx* ctx, int W, const double* theta);
int psfmc_get_accumulated(psfmc_ctx* ctx, double* raw, double* conv, double* resid, double* ivm,
                          double* ps_sub, long long* count);
int psfmc_reset_accumulated(psfmc_ctx* ctx);

/*
 * Device-computed PSF spectra, for checking the on-device replacement of
 * pre_fft_psf (utils.py:126-133): out arrays [n_psf][ny][nx/2+1][2] (re, im),
 * equal to numpy.fft.rfft2 of the centre-padded images.
 */
int psfmc_get_spectra(psfmc_ctx* ctx, double* psf_spec, double* var_spec);

/* tuning knobs: "chunk_walkers" (walkers per internal pass), "streams" (passes in
 * flight, 1..4), "cols_grid", "stagger" (0 / 1: the second pass in flight starts one
 * forward-row kernel after the first; default per image shape, results do not depend on it),
 * "linear_accumulation" (0 / 1, default 1, fused back end): posterior-image samples are added up as
 * raw / raw^2 / point-source-only raw and convolved once when the images are read, instead of every
 * sample going through the transforms (the five images are linear in those three),
 * "profile" (1: time every kernel with HIP events; read
 * back with get_option "prof_ms_rows_fwd" / "prof_n_rows_fwd", ..._cols, ..._rows_inv).
 * "storage_f32" (0 / 1, default 0): keep the fused path's intermediate half-spectra as
 * complex64 while every operation stays fp64 -- half the memory traffic; the log-posterior is
 * then good to ~1e-7 relative, the class of the reference's own float32 raw-model accumulator
 * (psfMC/models.py:249), not an fp64 result.  Power-of-two sides, fused back end only.
 * "cols3" (0 ... 4, default 1): which column kernel runs -- 0 the two-stage engine wherever a side has one, 1 the
 * wave-wide three-stage engines where they measured faster (round 4: k_cols3f at 512 / 1024 / 1536 / 2048, k_cols3g
 * at the other sides of psfmc_fft.h fft3g_pick), 2 k_cols3g at those four as well, 3 round 3's k_cols3 at 512 and
 * 1024, 4 the same as 1; results agree to rounding.  Sides above 1024 have only the three-stage kernels.
 * "exclusive" (bits 0 / 1 / 2 = forward rows / columns / inverse rows, default 0): chain the two passes in flight so
 * that two kernels of that kind never run side by side (measurement knob; slower in every combination).
 * "speculate" (device sampler, psfmc_stretch_run): ensembles of up to 2 n walkers run ONE pipeline pass per
 * iteration -- the first half's proposals and both candidate proposals of every second-half walker (partner
 * moved / partner stayed) -- instead of two half-steps; the chain is the same bit for bit.  0 never, n > 0 that
 * bound, -1 (default) n = 3e6 / transform pixels, the measured break-even; needs max_walkers >= 1.5 W.
 * get_option only: "pow_tabs" (1: this context's rasterising kernels take (rho^2)^p from per-walker power tables --
 * transforms of more than 256 pixels per row --, 0: log2 + exp2 per pixel; a property of the transform shape),
 * "transform_ny" / "transform_nx" (the transform shape: the image's own, or the built sides an
 * image of unbuilt sides is embedded in), "speculated_runs", "graph_launches", "row_group",
 * "partials_per_walker", "column_engine" (the column kernel this context launches now: 0 k_cols, 1 k_cols3, 2
 * k_cols3g, 3 k_cols3f), "rows3" (bit 0 / bit 1: the forward / inverse row kernel is the one-row-per-wave
 * three-stage one of csrc/psfmc_rows3_path.h -- both above 1024, the inverse one at seven general sides; the
 * environment variable PSFMC_ROWS3 = 1 / 0, read at context creation, forces every built one / none: A/B runs).
 */
int psfmc_set_option(psfmc_ctx* ctx, const char* key, double value);
double psfmc_get_option(const psfmc_ctx* ctx, const char* key);

/*
 * Diagnostic hook: evaluate one of the library's fp64 device functions on n host
 * values (op 0 log2, 1 exp2, 2 reciprocal, 3 single-Newton reciprocal, 4 exp2 without
 * clamp, 5 the rasteriser's table-driven log2, 6 exp2 with the lower clamp only; op 100: x^p through the
 * rasteriser's per-walker power tables, with p passed as in[n], i.e. `in` holds n + 1 values) so tests can
 * check the hand-written elementary functions of the rasteriser against numpy.
 */
int psfmc_debug_math(int device, int op, int n, const double* in, double* out);

/*
 * Diagnostic hook: time `reps` plain sweeps over a scratch buffer of nbytes (>= 1 MiB) -- mode 0
 * every byte written once, 1 read and written back in place, 2 read once: the memory traffic of
 * the three kernels of a pass without their arithmetic.  *us_per_sweep = average microseconds.
 * bench.py reports them beside the timed step ("sweep_ceiling").
 */
int psfmc_debug_sweep(int device, int mode, size_t nbytes, int reps, double* us_per_sweep);
/* measurement hook: nanoseconds per fp64 vector wave-instruction per SIMD with `waves_per_simd` waves on every
 * SIMD of the chip (64 v_fma_f64 per loop iteration): the VALU ceiling next to psfmc_debug_sweep's memory one */
int psfmc_debug_valu_rate(int device, int waves_per_simd, int iters, double* ns_per_instruction);

/* message of the last failing call on this thread ("" if none) */
const char* psfmc_last_error(void);

/* library/ABI version (bumped when a signature changes) */
int psfmc_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* PSFMC_HIP_H */
