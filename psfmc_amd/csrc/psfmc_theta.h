// psfmc_theta.h -- raw emcee parameter vectors on the device: priors for the
// common distribution families, the Sersic constants (kappa = b_n, Sigma_e) and
// the expansion into prep records, so that a batch needs no host arithmetic.
//
// Reference: packing contract ComponentBase.py:45-74 + models.py:174-185; priors
// distributions.py:112-127 (scipy.stats frozen logpdf / logpmf) and Sersic.py:41-45;
// kappa Sersic.py:47-53 (scipy.special.gammaincinv(2n, 1/2)); Sigma_e Sersic.py:55-71;
// flux utils.py:160-164; ellipse matrix Sersic.py:80-91.
#pragma once
#ifndef PSFMC_PART
#define PSFMC_PART 0      /* single translation unit (see psfmc_hip.hip) */
#endif
#include "psfmc_device.h"

namespace psfmc {

enum PriorFamily { PRIOR_HOST = 0, PRIOR_UNIFORM = 1, PRIOR_NORMAL = 2, PRIOR_WEIBULL_MIN = 3,
                   PRIOR_RANDINT = 4 };

// slots of a model, in this order: n_sky x [adu] | n_ps x [mag, x, y] |
// n_sersic x [angle, index, mag, reff, reff_b, x, y] | [psf_index]
__host__ __device__ inline int n_slots(int n_sky, int n_ps, int n_sersic) {
    return n_sky + 3 * n_ps + 7 * n_sersic + 1;
}

struct ThetaLayout {
    int n_sky, n_ps, n_sersic, n_params, n_psf;
    double mag_zp;
    const int* slot_col;        // [n_slots] column of theta, or -1
    const double* slot_const;   // [n_slots]
    const int* ps_method;       // [n_ps]
    const int* sersic_deg;      // [n_sersic] angle in degrees?
    const int* family;          // [n_params]
    const double* pa;           // [n_params] loc / c / low
    const double* pb;           // [n_params] scale / loc / high
    const double* pc;           // [n_params] - / scale / -
    const double* pk;           // [n_params] the family's constant log term (prior_log_norm),
                                //            filled by psfmc_set_layout
};

// ---------------------------------------------------------------------------
// regularised lower incomplete gamma P(a, x), series form (valid and fast for
// x < a + 1, which brackets the median), and its inverse at 1/2
// ---------------------------------------------------------------------------
// x^a e^-x / Gamma(a+1), the prefactor of the series.  Below a = 20 it is formed
// directly from Gamma(a) (`tg`, which the caller needs for Sigma_e anyway: one OCML
// tgamma instead of lgamma + tgamma); from a = 20 on, Gamma(a+1) = sqrt(2 pi a) (a/e)^a
// exp(corr(a)) avoids the cancellation of a ln x - x against ln Gamma, and `stirling`
// = ln sqrt(2 pi a) + corr(a) depends on a only.
__device__ inline double igam_stirling_term(double a) {
    const double ia = 1.0 / a, ia2 = ia * ia;
    const double corr = ia * (1.0 / 12.0 + ia2 * (-1.0 / 360.0 + ia2 * (1.0 / 1260.0 + ia2 *
                        (-1.0 / 1680.0 + ia2 * (1.0 / 1188.0 + ia2 * (-691.0 / 360360.0))))));
    return 0.5 * log(6.28318530717958647693 * a) + corr;
}
__device__ inline double igam_prefactor(double a, double x, double tg, double stirling) {
    if (a < 20.0) return exp(a * log(x) - x) / (a * tg);
    const double u = (x - a) / a;
    return exp(a * (log1p(u) - u) - stirling);
}

// kappa = gammaincinv(a, 1/2), a = 2n > 0.  `tg` = tgamma(a) (used for a < 20 only).
__device__ inline double gamma_median(double a, double tg) {
    if (!(a > 0.0) || !(a < 1e6)) return __builtin_nan("");
    double x;
    if (a < 1.0) {
        x = exp(log(0.5 * a * tg) / a);                         // P ~ x^a / Gamma(a+1)
    } else {
        // Ciotti & Bertin (1999) eq. 18, plus a fitted remainder i^3 (c0 + c1 i + c2 i^2)
        // (least squares against gammaincinv over n in [0.5, 40]): relative error of the
        // start <= 1.1e-7 instead of 3.6e-4 at n = 0.5, so ONE Halley step is enough
        const double n = 0.5 * a, i = 1.0 / n;
        x = a - 1.0 / 3.0 + i * (4.0 / 405.0 + i * (46.0 / 25515.0 + i * (131.0 / 1148175.0 -
            i * (2194697.0 / 30690717750.0))));
        x -= i * i * i * (2.47501670e-05 + i * (3.05002372e-05 - i * 1.35458779e-05));
    }
    const double stirling = a < 20.0 ? 0.0 : igam_stirling_term(a);
    // Halley on P(a, x) = 1/2: cubic convergence, so a step below 1e-6 x leaves an
    // error of order 1e-18 x and the iteration stops there
    for (int it = 0; it < 12; ++it) {
        const double pre = igam_prefactor(a, x, tg, stirling);    // x^a e^-x / Gamma(a+1)
        // four terms per round: their reciprocals do not depend on each other, so the dependent chain
        // is four multiplications instead of four reciprocals (k_theta_prep is latency-bound: one
        // lane per walker)
        double term = 1.0, sum = 1.0, ap = a;
        for (int k = 0; k < 500; ++k) {
            const double q1 = x * fast_rcp(ap + 1.0), q2 = x * fast_rcp(ap + 2.0);
            const double q3 = x * fast_rcp(ap + 3.0), q4 = x * fast_rcp(ap + 4.0);
            ap += 4.0;
            const double t1 = term * q1, t2 = t1 * q2, t3 = t2 * q3;
            term = t3 * q4;
            sum = (((sum + t1) + t2) + t3) + term;
            if (term < 1e-17 * sum) break;
        }
        const double f = sum * pre - 0.5;
        const double d1 = pre * a / x;                            // dP/dx = x^(a-1) e^-x / Gamma(a)
        const double r = f / d1;
        const double dx = r / (1.0 + 0.5 * r * (1.0 - (a - 1.0) / x));
        x -= dx;
        if (!(x > 0.0)) x = 0.5 * (x + dx);
        if (fabs(dx) <= 1e-6 * x) break;
    }
    return x;
}

// Sigma_e (Sersic.py:55-71); tg = tgamma(2n)
__device__ inline double sersic_sb_eff(double flux, double n, double reff, double reff_b, double kappa,
                                       double tg) {
    return flux / (M_PI * reff * reff_b * 2.0 * n * exp(kappa + log(kappa) * -2.0 * n) * tg);
}

// ---------------------------------------------------------------------------
// priors
// ---------------------------------------------------------------------------
// The x-independent term of each family's log-density, computed once on the host
// (ten log() calls per walker otherwise).
__host__ inline double prior_log_norm(int fam, double a, double b, double c) {
    switch (fam) {
        case PRIOR_UNIFORM: return -log(b);
        case PRIOR_NORMAL: return -0.91893853320467274178 - log(b);
        case PRIOR_WEIBULL_MIN: return log(a) - log(c);
        case PRIOR_RANDINT: return -log(b - a);
        default: return 0.0;
    }
}

__device__ inline double prior_logp(int fam, double x, double a, double b, double c, double k) {
    const double ninf = -INFINITY;
    switch (fam) {
        case PRIOR_UNIFORM: {                                 // loc a, scale b
            const double y = (x - a) / b;
            return (y >= 0.0 && y <= 1.0) ? k : (y == y ? ninf : y);
        }
        case PRIOR_NORMAL: {                                  // loc a, scale b
            const double z = (x - a) / b;
            return -0.5 * z * z + k;
        }
        case PRIOR_WEIBULL_MIN: {                             // c a, loc b, scale c
            const double y = (x - b) / c;
            if (!(y >= 0.0)) return y == y ? ninf : y;
            if (y == 0.0) return a == 1.0 ? k : (a < 1.0 ? INFINITY : ninf);
            return k + (a - 1.0) * log(y) - pow(y, a);
        }
        case PRIOR_RANDINT: {                                 // low a, high b (exclusive); x already rounded
            return (x >= a && x <= b - 1.0) ? k : ninf;
        }
        default:
            return 0.0;
    }
}

// The work of one walker is split into independent TASKS that run on different waves
// of the workgroup (blockDim = (64 walkers, n waves)): one walker per thread left the
// whole GPU with two waves crawling through ~9000 dependent fp64 instructions (35 us
// for a 128-walker half-ensemble, a sixth of an MCMC half-step).
//   task 0                       joint log-prior (device families + the host's extra),
//                                support and axis-ratio checks -> lnprior, skip; sky, PSF index
//   tasks 1 + 2k, 2 + 2k         PointSource k: flux and the y / the x axis of its window
//   then 2 per Sersic            the ellipse + flux, and Gamma(2n) + kappa
// after a barrier one wave per Sersic forms Sigma_e and its prep block, and after another
// task 0's wave writes the head of the prep record from the largest peak estimate.
// Walkers outside the prior support get skip = 1; their prep record is scratch.
__device__ inline double theta_slot(const ThetaLayout& L, const double* __restrict__ theta, int s) {
    const int col = L.slot_col[s];
    return col >= 0 ? theta[col] : L.slot_const[s];
}

__device__ inline double theta_log_prior(const ThetaLayout& L, const double* __restrict__ theta, double extra) {
    double lp = extra;
    for (int p = 0; p < L.n_params; ++p) {
        const int fam = L.family[p];
        if (fam == PRIOR_HOST) continue;
        const double x = fam == PRIOR_RANDINT ? rint(theta[p]) : theta[p];
        lp += prior_logp(fam, x, L.pa[p], L.pb[p], L.pc[p], L.pk[p]);
    }
    // Sersic axis-ratio constraint (Sersic.py:41-45)
    for (int k = 0; k < L.n_sersic; ++k) {
        const int s0 = L.n_sky + 3 * L.n_ps + 7 * k;
        if (theta_slot(L, theta, s0 + 4) > theta_slot(L, theta, s0 + 3)) lp = -INFINITY;
    }
    return lp;
}

// row pieces (the caller-row layout of include/psfmc_hip.h)
__device__ inline void theta_ps_row(const ThetaLayout& L, const double* __restrict__ theta, int k,
                                    double* __restrict__ r) {
    const int s0 = L.n_sky + 3 * k;
    r[0] = pow(10.0, -0.4 * (theta_slot(L, theta, s0) - L.mag_zp));
    r[1] = theta_slot(L, theta, s0 + 1);
    r[2] = theta_slot(L, theta, s0 + 2);
    r[3] = (double)L.ps_method[k];
}

// Sersic row in three independent pieces: the ellipse (sincos), kappa (tgamma + the
// incomplete-gamma inversion), and Sigma_e, which needs both.  scratch = {Gamma(2n), flux}.
__device__ inline void theta_sersic_geometry(const ThetaLayout& L, const double* __restrict__ theta, int k,
                                             double* __restrict__ r, double* __restrict__ scratch) {
    const int s0 = L.n_sky + 3 * L.n_ps + 7 * k;
    const double ang = theta_slot(L, theta, s0), mag = theta_slot(L, theta, s0 + 2);
    const double re = theta_slot(L, theta, s0 + 3), rb = theta_slot(L, theta, s0 + 4);
    const double th = (L.sersic_deg[k] ? ang * (M_PI / 180.0) : ang) + 0.5 * M_PI;
    const double sn = sin(th), cs = cos(th);
    r[0] = theta_slot(L, theta, s0 + 5);
    r[1] = theta_slot(L, theta, s0 + 6);
    r[2] = cs / re;
    r[3] = sn / re;
    r[4] = -sn / rb;
    r[5] = cs / rb;
    scratch[1] = pow(10.0, -0.4 * (mag - L.mag_zp));
}

__device__ inline void theta_sersic_kappa(const ThetaLayout& L, const double* __restrict__ theta, int k,
                                          double* __restrict__ r, double* __restrict__ scratch) {
    const double n = theta_slot(L, theta, L.n_sky + 3 * L.n_ps + 7 * k + 1);
    const double tg = tgamma(2.0 * n);
    r[6] = gamma_median(2.0 * n, tg);
    r[7] = 0.5 / n;
    scratch[0] = tg;
}

__device__ inline void theta_sersic_sigma(const ThetaLayout& L, const double* __restrict__ theta, int k,
                                          double* __restrict__ r, const double* __restrict__ scratch) {
    const int s0 = L.n_sky + 3 * L.n_ps + 7 * k;
    r[8] = sersic_sb_eff(scratch[1], theta_slot(L, theta, s0 + 1), theta_slot(L, theta, s0 + 3),
                         theta_slot(L, theta, s0 + 4), r[6], scratch[0]);
}

// Stretch-move proposal formed while the parameter tile is loaded (pos != nullptr):
// q[i] = c[j_i] - z_i (c[j_i] - s_i), s = half `h` of pos, c = the other half
// (Goodman & Weare 2010; emcee 2.2.1 `_propose_stretch`).  No FMA contraction: the
// same three roundings as the numpy expression emcee uses.  The iteration number
// comes from *d_iter when a captured hipGraph of one iteration is replayed, else `it`.
struct StretchIn {
    const double* pos;      // [2 half][P] current positions, or nullptr: theta is given
    double* q;              // [half][P] proposals (kept for the accept step)
    const double* z;        // [n_iter][2][half]
    const int* partner;     // [n_iter][2][half]
    const int* d_iter;
    int it, half, h;
    // Several fields in one launch (contexts of psfmc_ctx_create_fields; blockIdx.y = field): every
    // field is its own ensemble of 2 half walkers with its own random numbers.  Element strides
    // between consecutive fields of pos / q / z (and partner); 0 in a one-field launch.
    size_t pos_stride, q_stride, rand_stride;
    // spec = 1 (small ensembles, one field): ONE launch proposes a whole iteration -- 3 half walkers:
    //   [0, half)        the first half's proposals (as h = 0),
    //   [half, 2 half)   the second half's proposals IF the partner's proposal is accepted (the partner's
    //                    position is then its proposal, formed here again with the same three roundings),
    //   [2 half, 3 half) the second half's proposals if it is rejected (the partner stays where it is).
    // The accept step of the second half picks the row its partner's outcome selects (k_stretch_finish), so the
    // chain is the one of two half-steps run after each other, bit for bit, at one pipeline pass per iteration.
    int spec;
};

// the fields of one k_theta_prep launch (gridDim.y > 1): field f takes layouts[f], walkers
// [f W, (f + 1) W) of every per-walker array, kernel spectra from f * psf_stride on
struct FieldSegs {
    const ThetaLayout* layouts;   // device array [gridDim.y], or nullptr: one field, the layout passed by value
    int psf_stride;
};

__device__ inline double stretch_point(double s, double c, double z) {
#pragma clang fp contract(off)
    const double diff = c - s;
    const double step = z * diff;
    return c - step;
}

// LDS bytes of k_theta_prep: the layout tables, per walker its parameter vector and
// its derived row (the walk through the slots is a long chain of dependent small
// loads: from global memory it cost 45 us per call, from LDS it is a few us), and
// the components' peak estimates
constexpr int kThetaThreads = 64;       // walkers per workgroup
constexpr int kThetaMaxTasks = 8;       // waves per workgroup; further tasks loop
__host__ inline int theta_task_waves(int n_ps, int n_sersic) {
    const int t = 1 + 2 * n_ps + 2 * n_sersic;
    return t > kThetaMaxTasks ? kThetaMaxTasks : t;
}
__host__ inline size_t theta_prep_lds_bytes(int n_sky, int n_ps, int n_sersic, int n_params) {
    const size_t ns = n_slots(n_sky, n_ps, n_sersic);
    const size_t n_int = ((ns + n_ps + n_sersic + n_params + 1) / 2) * 2;           // 8-byte multiple
    const size_t n_dbl = ns + 4 * (size_t)n_params;
    return n_int * sizeof(int) +
           (n_dbl + (size_t)kThetaThreads * (n_params + row_len(n_ps, n_sersic) + n_ps + 3 * n_sersic)) *
               sizeof(double);
}

#if PSFMC_PART == 0          /* not a template: defined in the API part only */
__global__ void __launch_bounds__(kThetaThreads * kThetaMaxTasks)
k_theta_prep(ThetaLayout G, const double* __restrict__ theta,
             const double* __restrict__ extra, double* __restrict__ rows,
             double* __restrict__ prep, double* __restrict__ lnprior,
             uint8_t* __restrict__ skip, int W, int ny, int nx,
             const double* __restrict__ rho, StretchIn sp, int psf_base, FieldSegs segs) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    if (segs.layouts) {                                 // (wave-uniform) this workgroup's field
        const int f = blockIdx.y;
        G = segs.layouts[f];
        const size_t first = (size_t)f * W;             // its first walker in the per-walker arrays
        if (theta) theta += first * G.n_params;
        if (extra) extra += first;
        if (rows) rows += first * row_len(G.n_ps, G.n_sersic);
        prep += first * prep_len(G.n_ps, G.n_sersic);
        lnprior += first;
        skip += first;
        psf_base += f * segs.psf_stride;
        if (sp.pos) {
            sp.pos += (size_t)f * sp.pos_stride;
            sp.q += (size_t)f * sp.q_stride;
            sp.z += (size_t)f * sp.rand_stride;
            sp.partner += (size_t)f * sp.rand_stride;
        }
    }
    const int ns = n_slots(G.n_sky, G.n_ps, G.n_sersic);
    const int n_int = ((ns + G.n_ps + G.n_sersic + G.n_params + 1) / 2) * 2;
    const int n_dbl = ns + 4 * G.n_params;
    const int rlen = row_len(G.n_ps, G.n_sersic);
    const int P = G.n_params;
    int* li = reinterpret_cast<int*>(lds_raw);
    double* ld = reinterpret_cast<double*>(lds_raw + (size_t)n_int * sizeof(int));
    double* th_tile = ld + n_dbl;
    double* row_tile = th_tile + (size_t)kThetaThreads * P;
    double* peak_tile = row_tile + (size_t)kThetaThreads * rlen;       // [component][walker]
    const int tid = threadIdx.y * kThetaThreads + threadIdx.x, nthr = kThetaThreads * blockDim.y;
    // The four int tables and the four double tables are contiguous in the blob.  The layout
    // tables and the parameter tile are independent: every thread first ISSUES its (first) load
    // of each, then stores them to LDS -- one memory round trip instead of three in a row (the
    // kernel is a chain of latencies: 16 us for a 128-walker half-step whatever the batch).
    const int n_tab_i = ns + G.n_ps + G.n_sersic + P;
    const int w0 = blockIdx.x * kThetaThreads;
    const int n_here = W - w0 < kThetaThreads ? W - w0 : kThetaThreads;
    const int n_tile = n_here * P;
    int first_i = 0;
    double first_d = 0.0, first_t = 0.0;
    if (tid < n_tab_i) first_i = G.slot_col[tid];
    if (tid < n_dbl) first_d = G.slot_const[tid];
    size_t soff = 0;
    double s_own = 0.0, c_other = 0.0, zz = 0.0;
    // element i = lw P + d of the tile in a whole-iteration launch (sp.spec)
    auto propose_spec = [&](int i, size_t off0) -> double {
        const int lw = i / P, d = i - lw * P, w3 = w0 + lw;
        const int seg = w3 / sp.half, w = w3 - seg * sp.half;
        const size_t off1 = off0 + sp.half;
        if (seg == 0)
            return stretch_point(sp.pos[(size_t)w * P + d],
                                 sp.pos[(size_t)(sp.half + sp.partner[off0 + w]) * P + d], sp.z[off0 + w]);
        const int j = sp.partner[off1 + w];                          // a walker of the first half
        double cj = sp.pos[(size_t)j * P + d];
        if (seg == 1)
            cj = stretch_point(cj, sp.pos[(size_t)(sp.half + sp.partner[off0 + j]) * P + d], sp.z[off0 + j]);
        return stretch_point(sp.pos[(size_t)(sp.half + w) * P + d], cj, sp.z[off1 + w]);
    };
    if (sp.pos && sp.spec) {
        const int it = sp.d_iter ? *sp.d_iter : sp.it;
        soff = (size_t)it * 2 * sp.half;
    } else if (sp.pos) {                                            // propose into the tile
        const int it = sp.d_iter ? *sp.d_iter : sp.it;
        soff = ((size_t)it * 2 + sp.h) * sp.half;
        if (tid < n_tile) {
            const int lw = tid / P, d = tid - lw * P, w = w0 + lw;
            s_own = sp.pos[(size_t)(sp.h * sp.half + w) * P + d];
            c_other = sp.pos[(size_t)((1 - sp.h) * sp.half + sp.partner[soff + w]) * P + d];
            zz = sp.z[soff + w];
        }
    } else if (tid < n_tile) {
        first_t = theta[(size_t)w0 * P + tid];
    }
    if (tid < n_tab_i) li[tid] = first_i;
    if (tid < n_dbl) ld[tid] = first_d;
    for (int i = tid + nthr; i < n_tab_i; i += nthr) li[i] = G.slot_col[i];
    for (int i = tid + nthr; i < n_dbl; i += nthr) ld[i] = G.slot_const[i];
    if (sp.pos && sp.spec) {
        for (int i = tid; i < n_tile; i += nthr) {
            const double q = propose_spec(i, soff);
            th_tile[i] = q;
            sp.q[(size_t)w0 * P + i] = q;
        }
    } else if (sp.pos) {
        if (tid < n_tile) {
            const double q = stretch_point(s_own, c_other, zz);
            th_tile[tid] = q;
            sp.q[(size_t)w0 * P + tid] = q;
        }
        for (int i = tid + nthr; i < n_tile; i += nthr) {
            const int lw = i / P, d = i - lw * P, w = w0 + lw;
            const double s = sp.pos[(size_t)(sp.h * sp.half + w) * P + d];
            const double c = sp.pos[(size_t)((1 - sp.h) * sp.half + sp.partner[soff + w]) * P + d];
            const double q = stretch_point(s, c, sp.z[soff + w]);
            th_tile[i] = q;
            sp.q[(size_t)w0 * P + i] = q;
        }
    } else {
        if (tid < n_tile) th_tile[tid] = first_t;
        for (int i = tid + nthr; i < n_tile; i += nthr)             // coalesced tile load
            th_tile[i] = theta[(size_t)w0 * P + i];
    }
    __syncthreads();
    ThetaLayout L = G;
    L.slot_col = li; L.ps_method = li + ns; L.sersic_deg = li + ns + G.n_ps;
    L.family = li + ns + G.n_ps + G.n_sersic;
    L.slot_const = ld; L.pa = ld + ns; L.pb = ld + ns + P; L.pc = ld + ns + 2 * P; L.pk = ld + ns + 3 * P;
    const int lw = threadIdx.x, w = w0 + lw;
    const bool active = w < W;
    const double* th = th_tile + (size_t)lw * P;
    double* row = row_tile + (size_t)lw * rlen;
    double* my_prep = prep + (size_t)(active ? w : 0) * prep_len(G.n_ps, G.n_sersic);
    double* sersic_scratch = peak_tile + (size_t)(G.n_ps + G.n_sersic) * kThetaThreads;   // [sersic][2][walker]
    const int n_tasks = 1 + 2 * G.n_ps + 2 * G.n_sersic;
    bool ok = false;
    for (int task = threadIdx.y; task < n_tasks; task += blockDim.y) {      // wave-uniform
        if (!active) continue;
        if (task == 0) {
            const double lp = theta_log_prior(L, th, extra ? extra[w] : 0.0);
            ok = lp == lp && fabs(lp) != INFINITY;                          // finite
            lnprior[w] = lp;
            skip[w] = ok ? 0 : 1;
            double sky = 0.0;
            for (int k = 0; k < L.n_sky; ++k) sky += theta_slot(L, th, k);
            row[0] = sky;
            double psf = rint(theta_slot(L, th, ns - 1));
            psf = psf < 0.0 ? 0.0 : (psf > (double)(L.n_psf - 1) ? (double)(L.n_psf - 1) : psf);
            row[rlen - 1] = psf;
        } else if (task <= 2 * G.n_ps) {
            // both axis tasks derive the (identical) row piece; each writes its own half of
            // the prep block, the y task also the row and the peak estimate
            const int k = (task - 1) >> 1, axis = (task - 1) & 1;
            double rr[kRowPs];
            theta_ps_row(L, th, k, rr);
            prep_ps_axis(rr, my_prep + kPrepHead + kPrepPs * k, axis, axis ? nx : ny);
            if (axis == 0) {
                double* r = row + kRowSky + kRowPs * k;
                for (int j = 0; j < kRowPs; ++j) r[j] = rr[j];
                peak_tile[k * kThetaThreads + lw] = fabs(rr[0]);
            }
        } else {
            const int j = task - 1 - 2 * G.n_ps, k = j >> 1;
            double* r = row + kRowSky + kRowPs * G.n_ps + kRowSersic * k;
            double scr[2];
            if (j & 1) {
                theta_sersic_kappa(L, th, k, r, scr);
                sersic_scratch[(k * 2 + 0) * kThetaThreads + lw] = scr[0];
            } else {
                theta_sersic_geometry(L, th, k, r, scr);
                sersic_scratch[(k * 2 + 1) * kThetaThreads + lw] = scr[1];
            }
        }
    }
    __syncthreads();
    for (int k = threadIdx.y; k < G.n_sersic; k += blockDim.y) {            // wave-uniform
        if (!active) continue;
        double* r = row + kRowSky + kRowPs * G.n_ps + kRowSersic * k;
        const double scr[2] = {sersic_scratch[(k * 2 + 0) * kThetaThreads + lw],
                               sersic_scratch[(k * 2 + 1) * kThetaThreads + lw]};
        theta_sersic_sigma(L, th, k, r, scr);
        peak_tile[(G.n_ps + k) * kThetaThreads + lw] =
            prep_sersic_block(r, my_prep + kPrepHead + kPrepPs * G.n_ps + kPrepSersic * k);
    }
    __syncthreads();
    if (threadIdx.y != 0 || !active) return;
    double peak = fabs(row[0]);
    for (int k = 0; k < G.n_ps + G.n_sersic; ++k) peak = fmax(peak, peak_tile[k * kThetaThreads + lw]);
    // psf_base: first kernel-spectrum index of this walker's field (contexts holding several fields)
    prep_head(my_prep, row[0], psf_base + (int)row[rlen - 1], peak, rho);
    if (rows && ok) {
        double* out = rows + (size_t)w * rlen;
        for (int i = 0; i < rlen; ++i) out[i] = row[i];
    }
}
#endif

// lnprob = loglike + lnprior, non-finite likelihood -> -inf (models.py:238-243); all
// lanes of the walker's wave get the value
__device__ inline double walker_lnprob(const double* __restrict__ partial, const uint8_t* __restrict__ skip,
                                       const double* __restrict__ lnprior, int nblk, int w, int lane) {
    if (skip[w]) return -INFINITY;
    const double ll = -0.5 * wave_sum_partials(partial + (size_t)w * nblk, nblk, lane);
    const bool fin = ll == ll && fabs(ll) != INFINITY;
    return fin ? ll + lnprior[w] : -INFINITY;
}

// one wave per walker (launch: finish_blocks(W) x kFinishThreads)
#if PSFMC_PART == 0          /* not a template: defined in the API part only */
__global__ void k_finish_posterior(const double* __restrict__ partial, const uint8_t* __restrict__ skip,
                                   const double* __restrict__ lnprior, double* __restrict__ lnprob,
                                   int W, int nblk) {
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * (kFinishThreads / 64) + (threadIdx.x >> 6);
    if (w >= W) return;
    const double lp = walker_lnprob(partial, skip, lnprior, nblk, w, lane);
    if (lane == 0) lnprob[w] = lp;
}
#endif

// ---------------------------------------------------------------------------
// stretch move, second half of a half-step: the proposals' log-posteriors
// (k_finish_posterior's sum), acceptance where lz + newlnp - lnp > ln u (emcee 2.2.1
// `_propose_stretch`), the move, and the chain entry of this half's walkers -- the
// other half's positions do not change during this half-step, so iteration `it` of
// walker g is final after the half-step that owns g.  One wave per walker (launch:
// finish_blocks(half) x kFinishThreads).
// `newlnp_in` (multi-GPU): the proposals' log-posteriors were evaluated in blocks on several
// ranks and gathered; they are read from it instead of being summed here -- the values are the
// ones k_finish_posterior produced from the same partial sums, so the chain is the same bit for bit.
// ---------------------------------------------------------------------------
#if PSFMC_PART == 0          /* not a template: defined in the API part only */
__global__ void k_stretch_finish(const double* __restrict__ partial, const uint8_t* __restrict__ skip,
                                 const double* __restrict__ lnprior, int nblk,
                                 const double* __restrict__ newlnp_in,
                                 double* __restrict__ pos, double* __restrict__ lnprob,
                                 const double* __restrict__ q, const double* __restrict__ lz,
                                 const double* __restrict__ log_u, long long* __restrict__ nacc,
                                 double* __restrict__ chain, double* __restrict__ lnchain,
                                 const int* __restrict__ d_iter, int it_val, int n_iter, int half, int h,
                                 int P, size_t rand_stride, uint8_t* __restrict__ acc_out,
                                 const uint8_t* __restrict__ acc_in, const int* __restrict__ partner) {
#pragma clang fp contract(off)
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * (kFinishThreads / 64) + (threadIdx.x >> 6);
    if (w >= half) return;                                   // wave-uniform
    if (gridDim.y > 1) {                                     // several fields (psfmc_stretch_run_fields): blockIdx.y's ensemble
        const size_t f = blockIdx.y;
        partial += f * half * nblk;
        skip += f * half;
        lnprior += f * half;
        if (newlnp_in) newlnp_in += f * half;
        pos += f * 2 * half * P;
        lnprob += f * 2 * half;
        q += f * half * P;
        lz += f * rand_stride;
        log_u += f * rand_stride;
        nacc += f * 2 * half;
        if (chain) {
            chain += f * 2 * half * n_iter * P;
            lnchain += f * 2 * half * n_iter;
        }
    }
    const int it = d_iter ? *d_iter : it_val;
    const size_t off = ((size_t)it * 2 + h) * half;
    // whole-iteration launches (StretchIn::spec): the second half's walker takes the proposal row its
    // partner's outcome selects -- half + w if the partner moved, 2 half + w if it stayed
    int row = w;
    if (acc_in) row = w + (acc_in[partner[off + w]] ? half : 2 * half);
    const double newlnp = newlnp_in ? newlnp_in[w] : walker_lnprob(partial, skip, lnprior, nblk, row, lane);
    const int g = h * half + w;
    double lp = lnprob[g];
    const double diff = (lz[off + w] + newlnp) - lp;
    const bool accept = diff > log_u[off + w];               // the same in every lane
    if (accept) lp = newlnp;
    for (int d = lane; d < P; d += 64) {                     // lane d moves coordinate d
        double v;
        if (accept) {
            v = q[(size_t)row * P + d];
            pos[(size_t)g * P + d] = v;
        } else {
            v = pos[(size_t)g * P + d];
        }
        if (chain) chain[((size_t)g * n_iter + it) * P + d] = v;
    }
    if (lane == 0) {
        if (acc_out) acc_out[w] = accept ? 1 : 0;
        if (accept) {
            lnprob[g] = lp;
            nacc[g] += 1;
        }
        if (chain) lnchain[(size_t)g * n_iter + it] = lp;
    }
}
#endif

#if PSFMC_PART == 0          /* not a template: defined in the API part only */
__global__ void k_stretch_next(int* __restrict__ d_iter) { *d_iter += 1; }
#endif

}  // namespace psfmc
