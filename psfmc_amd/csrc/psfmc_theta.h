// psfmc_theta.h -- raw emcee parameter vectors on the device: priors for the
// common distribution families, the Sersic constants (kappa = b_n, Sigma_e) and
// the expansion into prep records, so that a batch needs no host arithmetic.
//
// Reference: packing contract ComponentBase.py:45-74 + models.py:174-185; priors
// distributions.py:112-127 (scipy.stats frozen logpdf / logpmf) and Sersic.py:41-45;
// kappa Sersic.py:47-53 (scipy.special.gammaincinv(2n, 1/2)); Sigma_e Sersic.py:55-71;
// flux utils.py:160-164; ellipse matrix Sersic.py:80-91.
#pragma once
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
};

// ---------------------------------------------------------------------------
// regularised lower incomplete gamma P(a, x), series form (valid and fast for
// x < a + 1, which brackets the median), and its inverse at 1/2
// ---------------------------------------------------------------------------
// log of x^a e^-x / Gamma(a+1)
__device__ inline double igam_log_prefactor(double a, double x) {
    if (a < 20.0) return a * log(x) - x - lgamma(a + 1.0);
    // Gamma(a+1) = sqrt(2 pi a) (a/e)^a exp(corr(a)):  avoids the cancellation of
    // a ln x - x against lgamma for large a
    const double u = (x - a) / a;
    const double ia = 1.0 / a, ia2 = ia * ia;
    const double corr = ia * (1.0 / 12.0 + ia2 * (-1.0 / 360.0 + ia2 * (1.0 / 1260.0 + ia2 *
                        (-1.0 / 1680.0 + ia2 * (1.0 / 1188.0 + ia2 * (-691.0 / 360360.0))))));
    return a * (log1p(u) - u) - 0.5 * log(6.28318530717958647693 * a) - corr;
}

__device__ inline double igam_series(double a, double x) {        // P(a, x)
    double term = 1.0, sum = 1.0, ap = a;
    for (int k = 0; k < 2000; ++k) {
        ap += 1.0;
        term *= x / ap;
        sum += term;
        if (term < 1e-17 * sum) break;
    }
    return sum * exp(igam_log_prefactor(a, x));
}

// kappa = gammaincinv(a, 1/2), a = 2n > 0
__device__ inline double gamma_median(double a) {
    if (!(a > 0.0) || !(a < 1e6)) return __builtin_nan("");
    double x;
    if (a < 1.0) {
        x = exp((log(0.5) + lgamma(a + 1.0)) / a);            // P ~ x^a / Gamma(a+1)
    } else {
        const double n = 0.5 * a, i = 1.0 / n;                  // Ciotti & Bertin (1999) eq. 18
        x = a - 1.0 / 3.0 + i * (4.0 / 405.0 + i * (46.0 / 25515.0 + i * (131.0 / 1148175.0 -
            i * (2194697.0 / 30690717750.0))));
    }
    // Halley on P(a, x) = 1/2: cubic convergence, so a step below 1e-6 x leaves an
    // error of order 1e-18 x and the iteration stops there (the starting values are
    // good to 1e-3 ... 1e-9, i.e. one or two steps)
    for (int it = 0; it < 12; ++it) {
        const double pre = exp(igam_log_prefactor(a, x));         // x^a e^-x / Gamma(a+1)
        double term = 1.0, sum = 1.0, ap = a;
        for (int k = 0; k < 2000; ++k) {
            ap += 1.0;
            term *= x * fast_rcp(ap);
            sum += term;
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

__device__ inline double sersic_sb_eff(double flux, double n, double reff, double reff_b, double kappa) {
    return flux / (M_PI * reff * reff_b * 2.0 * n * exp(kappa + log(kappa) * -2.0 * n) * tgamma(2.0 * n));
}

// ---------------------------------------------------------------------------
// priors
// ---------------------------------------------------------------------------
__device__ inline double prior_logp(int fam, double x, double a, double b, double c) {
    const double ninf = -INFINITY;
    switch (fam) {
        case PRIOR_UNIFORM: {                                 // loc a, scale b
            const double y = (x - a) / b;
            return (y >= 0.0 && y <= 1.0) ? -log(b) : (y == y ? ninf : y);
        }
        case PRIOR_NORMAL: {                                  // loc a, scale b
            const double z = (x - a) / b;
            return -0.5 * z * z - 0.91893853320467274178 - log(b);
        }
        case PRIOR_WEIBULL_MIN: {                             // c a, loc b, scale c
            const double y = (x - b) / c;
            if (!(y >= 0.0)) return y == y ? ninf : y;
            if (y == 0.0) return a == 1.0 ? -log(c) : (a < 1.0 ? INFINITY : ninf);
            return log(a) + (a - 1.0) * log(y) - pow(y, a) - log(c);
        }
        case PRIOR_RANDINT: {                                 // low a, high b (exclusive); x already rounded
            return (x >= a && x <= b - 1.0) ? -log(b - a) : ninf;
        }
        default:
            return 0.0;
    }
}

// One thread per walker: joint log-prior (device families + the host's extra),
// non-finite -> skip; otherwise derive the caller row and expand it to the prep
// record.  `row` is scratch of row_len doubles for this walker.
__device__ inline void theta_to_prep(const ThetaLayout& L, const double* __restrict__ theta,
                                     double extra, double* __restrict__ row, double* __restrict__ prep,
                                     double* lnprior_out, uint8_t* skip_out, int ny, int nx,
                                     const double* __restrict__ rho) {
    double lp = extra;
    const int ns = n_slots(L.n_sky, L.n_ps, L.n_sersic);
    auto slot = [&](int s) -> double {
        const int col = L.slot_col[s];
        return col >= 0 ? theta[col] : L.slot_const[s];
    };
    for (int p = 0; p < L.n_params; ++p) {
        const int fam = L.family[p];
        if (fam == PRIOR_HOST) continue;
        const double x = fam == PRIOR_RANDINT ? rint(theta[p]) : theta[p];
        lp += prior_logp(fam, x, L.pa[p], L.pb[p], L.pc[p]);
    }
    // Sersic axis-ratio constraint (Sersic.py:41-45)
    for (int k = 0; k < L.n_sersic; ++k) {
        const int s0 = L.n_sky + 3 * L.n_ps + 7 * k;
        if (slot(s0 + 4) > slot(s0 + 3)) lp = -INFINITY;
    }
    *lnprior_out = lp;
    const bool ok = lp == lp && fabs(lp) != INFINITY;        // finite
    *skip_out = ok ? 0 : 1;
    if (!ok) return;
    double sky = 0.0;
    for (int k = 0; k < L.n_sky; ++k) sky += slot(k);
    row[0] = sky;
    double* r = row + kRowSky;
    for (int k = 0; k < L.n_ps; ++k, r += kRowPs) {
        const int s0 = L.n_sky + 3 * k;
        r[0] = pow(10.0, -0.4 * (slot(s0) - L.mag_zp));
        r[1] = slot(s0 + 1);
        r[2] = slot(s0 + 2);
        r[3] = (double)L.ps_method[k];
    }
    for (int k = 0; k < L.n_sersic; ++k, r += kRowSersic) {
        const int s0 = L.n_sky + 3 * L.n_ps + 7 * k;
        const double ang = slot(s0), n = slot(s0 + 1), mag = slot(s0 + 2);
        const double re = slot(s0 + 3), rb = slot(s0 + 4);
        const double th = (L.sersic_deg[k] ? ang * (M_PI / 180.0) : ang) + 0.5 * M_PI;
        const double sn = sin(th), cs = cos(th);
        const double kappa = gamma_median(2.0 * n);
        r[0] = slot(s0 + 5);
        r[1] = slot(s0 + 6);
        r[2] = cs / re;
        r[3] = sn / re;
        r[4] = -sn / rb;
        r[5] = cs / rb;
        r[6] = kappa;
        r[7] = 0.5 / n;
        r[8] = sersic_sb_eff(pow(10.0, -0.4 * (mag - L.mag_zp)), n, re, rb, kappa);
    }
    double psf = rint(slot(ns - 1));
    psf = psf < 0.0 ? 0.0 : (psf > (double)(L.n_psf - 1) ? (double)(L.n_psf - 1) : psf);
    r[0] = psf;
    build_prep(row, prep, L.n_ps, L.n_sersic, ny, nx, rho);
}

// LDS bytes of k_theta_prep: the layout tables, and per thread its parameter vector
// and its derived row (the walk through the slots is a long chain of dependent small
// loads: from global memory it cost 45 us per call, from LDS it is a few us)
constexpr int kThetaThreads = 64;
__host__ inline size_t theta_prep_lds_bytes(int n_sky, int n_ps, int n_sersic, int n_params) {
    const size_t ns = n_slots(n_sky, n_ps, n_sersic);
    const size_t n_int = ((ns + n_ps + n_sersic + n_params + 1) / 2) * 2;           // 8-byte multiple
    const size_t n_dbl = ns + 3 * (size_t)n_params;
    return n_int * sizeof(int) + (n_dbl + (size_t)kThetaThreads * (n_params + row_len(n_ps, n_sersic))) *
                                     sizeof(double);
}

__global__ void __launch_bounds__(kThetaThreads)
k_theta_prep(ThetaLayout G, const double* __restrict__ theta,
             const double* __restrict__ extra, double* __restrict__ rows,
             double* __restrict__ prep, double* __restrict__ lnprior,
             uint8_t* __restrict__ skip, int W, int ny, int nx,
             const double* __restrict__ rho) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int ns = n_slots(G.n_sky, G.n_ps, G.n_sersic);
    const int n_int = ((ns + G.n_ps + G.n_sersic + G.n_params + 1) / 2) * 2;
    const int n_dbl = ns + 3 * G.n_params;
    const int rlen = row_len(G.n_ps, G.n_sersic);
    int* li = reinterpret_cast<int*>(lds_raw);
    double* ld = reinterpret_cast<double*>(lds_raw + (size_t)n_int * sizeof(int));
    double* th_tile = ld + n_dbl;
    double* row_tile = th_tile + (size_t)kThetaThreads * G.n_params;
    // the four int tables and the four double tables are contiguous in the blob
    for (int i = threadIdx.x; i < ns + G.n_ps + G.n_sersic + G.n_params; i += kThetaThreads) li[i] = G.slot_col[i];
    for (int i = threadIdx.x; i < n_dbl; i += kThetaThreads) ld[i] = G.slot_const[i];
    const int w0 = blockIdx.x * kThetaThreads;
    const int n_here = W - w0 < kThetaThreads ? W - w0 : kThetaThreads;
    for (int i = threadIdx.x; i < n_here * G.n_params; i += kThetaThreads)      // coalesced tile load
        th_tile[i] = theta[(size_t)w0 * G.n_params + i];
    __syncthreads();
    ThetaLayout L = G;
    L.slot_col = li; L.ps_method = li + ns; L.sersic_deg = li + ns + G.n_ps;
    L.family = li + ns + G.n_ps + G.n_sersic;
    L.slot_const = ld; L.pa = ld + ns; L.pb = ld + ns + G.n_params; L.pc = ld + ns + 2 * G.n_params;
    const int w = w0 + threadIdx.x;
    if (w >= W) return;
    double* row = row_tile + (size_t)threadIdx.x * rlen;
    theta_to_prep(L, th_tile + (size_t)threadIdx.x * G.n_params, extra ? extra[w] : 0.0, row,
                  prep + (size_t)w * prep_len(G.n_ps, G.n_sersic), lnprior + w, skip + w, ny, nx, rho);
    if (rows) {
        double* out = rows + (size_t)w * rlen;
        for (int i = 0; i < rlen; ++i) out[i] = row[i];
    }
}

// lnprob[w] = loglike + lnprior, non-finite likelihood -> -inf (models.py:238-243)
__global__ void k_finish_posterior(const double* __restrict__ partial, const uint8_t* __restrict__ skip,
                                   const double* __restrict__ lnprior, double* __restrict__ lnprob,
                                   int W, int nblk) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= W) return;
    if (skip[w]) { lnprob[w] = -INFINITY; return; }
    double s = 0.0;
    for (int i = 0; i < nblk; ++i) s += partial[(size_t)w * nblk + i];
    const double ll = -0.5 * s;
    const bool fin = ll == ll && fabs(ll) != INFINITY;
    lnprob[w] = fin ? ll + lnprior[w] : -INFINITY;
}

// ---------------------------------------------------------------------------
// stretch move (Goodman & Weare 2010; emcee 2.2.1 `_propose_stretch`)
// ---------------------------------------------------------------------------
// q[i] = c[j_i] - z_i (c[j_i] - s_i), s = half `h` of pos, c = the other half.
// No FMA contraction: the same three roundings as the numpy expression emcee uses.
// The iteration number is read from device memory (*d_iter) so that one captured
// hipGraph of an iteration can be replayed for every iteration.
__global__ void k_stretch_propose(const double* __restrict__ pos, double* __restrict__ q,
                                  const double* __restrict__ z, const int* __restrict__ partner,
                                  const int* __restrict__ d_iter, int half, int h, int P) {
#pragma clang fp contract(off)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= half * P) return;
    const size_t off = ((size_t)*d_iter * 2 + h) * half;
    const int w = i / P, d = i - w * P;
    const double s = pos[(size_t)(h * half + w) * P + d];
    const double c = pos[(size_t)((1 - h) * half + partner[off + w]) * P + d];
    const double diff = c - s;
    const double step = z[off + w] * diff;
    q[i] = c - step;
}

// accept where lz + newlnp - lnp > ln u; move the walker, count it
__global__ void k_stretch_accept(double* __restrict__ pos, double* __restrict__ lnprob,
                                 const double* __restrict__ q, const double* __restrict__ newlnp,
                                 const double* __restrict__ lz, const double* __restrict__ log_u,
                                 long long* __restrict__ nacc, const int* __restrict__ d_iter, int half,
                                 int h, int P) {
#pragma clang fp contract(off)
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= half) return;
    const size_t off = ((size_t)*d_iter * 2 + h) * half;
    const int g = h * half + w;
    const double diff = (lz[off + w] + newlnp[w]) - lnprob[g];
    if (diff > log_u[off + w]) {
        for (int d = 0; d < P; ++d) pos[(size_t)g * P + d] = q[(size_t)w * P + d];
        lnprob[g] = newlnp[w];
        nacc[g] += 1;
    }
}

// chain[w][it][:] = pos[w][:], lnchain[w][it] = lnprob[w]; the last thread advances *d_iter
__global__ void k_stretch_store(const double* __restrict__ pos, const double* __restrict__ lnprob,
                                double* __restrict__ chain, double* __restrict__ lnchain, int W, int P,
                                int* __restrict__ d_iter, int n_iter) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int it = *d_iter;
    if (i < W * P && chain) {
        const int w = i / P, d = i - w * P;
        chain[((size_t)w * n_iter + it) * P + d] = pos[i];
        if (d == 0) lnchain[(size_t)w * n_iter + it] = lnprob[w];
    }
}

__global__ void k_stretch_next(int* __restrict__ d_iter) { *d_iter += 1; }

}  // namespace psfmc
