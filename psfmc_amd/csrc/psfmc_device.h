// psfmc_device.h -- device-side building blocks shared by every kernel of
// libpsfmc_hip: the per-walker "prep" record, the rasteriser pixel function
// (Sky + PointSource + Sersic) and the Gaussian chi^2 term.
//
// Written for gfx950 (CDNA4, wave64) only.  All arithmetic is fp64.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "psfmc_log_table.h"

namespace psfmc {

// LDS hand-off between lanes of ONE wave.  A wave's DS instructions execute in
// program order, so a ds_read issued after a ds_write of the same wave observes
// it; all that is needed is that the compiler keeps that order (the fences) and
// that the wave is converged here.  No s_barrier: waves never wait for each other.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int kRowSky = 1;      // doubles in a caller row (include/psfmc_hip.h)
constexpr int kRowPs = 4;
constexpr int kRowSersic = 9;

constexpr int kTaps = 7;        // widest point-source window (lanczos3)
// prep record (doubles):
//   [0] sky  [1] psf index  [2] mu  [3] 1/lambda   (channel scales of the fused path)
//   per PS: ylo, yn, xlo, xn, wy[7], wx[7] (flux folded into wx)
//   per Sersic: x0 y0 m00 m01 m10 m11 kappa p sb_eff
constexpr int kPrepHead = 4;
constexpr int kPrepPsfIdx = 1, kPrepMu = 2, kPrepInvLambda = 3;
constexpr int kPrepPs = 4 + 2 * kTaps;
constexpr int kPrepSersic = 9;

__host__ __device__ inline int row_len(int n_ps, int n_sersic) {
    return kRowSky + kRowPs * n_ps + kRowSersic * n_sersic + 1;
}
// Behind the record proper, per Sersic component: the POWER TABLES of its radial exponent p = 1/(2n)
// (k_pow_tables; what raster_row reads instead of evaluating log2 + exp2 per pixel):
//   [0, 256)    PB[j] = 2^(p b_j), b_j = -log2(a_j) of the rasteriser's mantissa table (psfmc_log_table.h)
//   [256, 512)  PE[i] = 2^(p (i - 128)): the exponent e = i - 128 of rho^2 = 2^e m, clamped to [-128, 127]
constexpr int kPowTabB = 256, kPowTabE = 256, kPowTabEBias = 128;
constexpr int kPowTab = kPowTabB + kPowTabE;
// Which form a rasterising kernel uses is a property of its row length (raster_row's TABS, chosen by
// pow_tabs_side): power tables for transforms with more than 256 pixels per row, log2 + exp2 per pixel (round
// 2's form, unchanged) up to 256 -- there the step is memory-bound and the tables measured as a loss with one
// component (256^2: -1.3 % whole step, same box; 128^2 -1.4 %, 64^2 -3 %: 4 KB of tables per row wave through
// the L2 against 16 instructions per pixel), as a small gain with two (+0.7 ... 2.8 %).  Both forms in one
// kernel behind a run-time switch cost 84 spilled registers at 256 and half the rate.  Where the tables are used,
// a batch either has them behind its prep records (kPowTabsBuilt: k_pow_tables ran) or is small and lets every
// row wave form the entries it reads (kPowTabsInWave; same function, same bits), so a walker's result never
// depends on what it is batched with.  The forward row kernels only: k_raster_sums (posterior-image sums, 7 ... 16
// pixels per lane and call) keeps the log2 + exp2 form at every size.
__host__ __device__ constexpr bool pow_tabs_side(int nx) { return nx > 256; }
constexpr int kPowTabsBuilt = 0, kPowTabsInWave = 1;
__host__ __device__ inline int prep_rec_len(int n_ps, int n_sersic) {
    return kPrepHead + kPrepPs * n_ps + kPrepSersic * n_sersic;
}
__host__ __device__ inline int prep_len(int n_ps, int n_sersic) {
    return prep_rec_len(n_ps, n_sersic) + kPowTab * n_sersic;
}

// ---------------------------------------------------------------------------
// Point-source window and 1-D weights.
// psfMC/ModelComponents/PointSource.py:60-81 (minimal_slice: clip, then numpy
// round-half-even), :84-97 (sinc, lanczos), :40-51 (separable product; the
// differences use the UNCLIPPED position; weights are not normalised).
// ---------------------------------------------------------------------------
__device__ inline double sinc_pi(double x) {
    const double pix = M_PI * x;
    return x != 0.0 ? sin(pix) / pix : 1.0;
}

__device__ inline double ps_weight(double d, int method) {
    if (method == 1)                       // bilinear
        return 1.0 - fabs(d);
    return fabs(d) < 3.0 ? sinc_pi(d) * sinc_pi(d / 3.0) : 0.0;
}

// one axis: centre c, axis length n -> first tap lo, tap count cnt (0..7)
__device__ inline void ps_window(double c, int n, int method, int* lo, int* cnt) {
    const double r = method == 1 ? 0.5 : 3.0;
    double cc = fmin(fmax(c, r - 0.5), (double)n - (r + 0.5));
    int a = (int)rint(cc - r);
    int b = (int)rint(cc + r);
    if (a < 0) a = 0;                      // numpy slicing truncates at the edges
    if (b > n - 1) b = n - 1;
    *lo = a;
    *cnt = b >= a ? b - a + 1 : 0;
    if (*cnt > kTaps) *cnt = kTaps;
}

// Expand one caller row into a prep record.  One thread per walker.
//
// mu / lambda: the fused path transforms raw + i*mu*raw^2 as ONE complex signal
// and gets conv + i*lambda*var back from one inverse transform.  A complex FFT's
// rounding error in either part is ~eps times the norm of the WHOLE signal, so the
// two channels are kept at comparable magnitude with exact power-of-two scales:
// mu ~ 1/peak(raw) (then mu*raw^2 ~ raw) and lambda = mu*rho with rho ~ 1/sum(psf
// variance map) a per-PSF constant already folded into the kernel spectrum
// (then lambda*var ~ conv).  `peak` only needs the right order of magnitude.
// The three pieces of a prep record.  Each component block is independent of the
// others (k_theta_prep builds them on different waves); the head needs the largest
// of the components' peak estimates.
// one axis of a point source's window: axis 0 = y (weights p[4..]), axis 1 = x (weights
// p[4 + kTaps..], flux folded in)
__device__ inline void prep_ps_axis(const double* __restrict__ r, double* __restrict__ p, int axis, int n) {
    const double flux = r[0], c0 = axis ? r[1] : r[2];
    const int method = (int)r[3];
    int lo, cnt;
    ps_window(c0, n, method, &lo, &cnt);
    p[2 * axis] = lo;
    p[2 * axis + 1] = cnt;
    double* wgt = p + 4 + kTaps * axis;
    for (int t = 0; t < kTaps; ++t) {
        const double wv = t < cnt ? ps_weight((double)(lo + t) - c0, method) : 0.0;
        wgt[t] = axis ? wv * flux : wv;
    }
}

__device__ inline double prep_ps_block(const double* __restrict__ r, double* __restrict__ p, int ny, int nx) {
    prep_ps_axis(r, p, 0, ny);
    prep_ps_axis(r, p, 1, nx);
    return fabs(r[0]);
}

__device__ inline double prep_sersic_block(const double* __restrict__ r, double* __restrict__ p) {
    for (int j = 0; j < kRowSersic; ++j) p[j] = r[j];
    // brightness half a pixel from the centre along the minor axis
    const double rho2 = 0.25 * fmax(r[2] * r[2] + r[4] * r[4], r[3] * r[3] + r[5] * r[5]);
    const double core = r[8] * exp(-r[6] * expm1(log(rho2) * r[7]));
    return core == core ? fabs(core) : 0.0;
}

__device__ inline void prep_head(double* __restrict__ prep, double sky, int psf, double peak,
                                 const double* __restrict__ rho) {
    prep[0] = sky;
    prep[kPrepPsfIdx] = (double)psf;
    double mu = 1.0;
    if (peak > 0.0 && peak < 1e300) {
        int e;
        (void)frexp(peak, &e);                 // peak = m * 2^e, m in [0.5, 1)
        mu = ldexp(1.0, -e);
    }
    const double lambda = mu * (rho ? rho[psf] : 1.0);
    prep[kPrepMu] = mu;
    prep[kPrepInvLambda] = 1.0 / lambda;
}

// The PSF index of a caller row (public C ABI: anything may arrive) is rounded and clamped
// to [0, n_psf): it indexes rho[] here and the kernel spectra in the column kernels.
__device__ inline int clamp_psf_index(double v, int n_psf) {
    const double r = rint(v);
    if (!(r >= 0.0)) return 0;                          // negative or NaN
    return r > (double)(n_psf - 1) ? n_psf - 1 : (int)r;
}

// psf_base: first kernel spectrum of the walker's field (contexts that hold several fields; else 0);
// n_psf: PSFs of that field
__device__ inline void build_prep(const double* __restrict__ row, double* __restrict__ prep,
                                  int n_ps, int n_sersic, int ny, int nx,
                                  const double* __restrict__ rho, int n_psf, int psf_base = 0) {
    double peak = fabs(row[0]);
    const double* r = row + kRowSky;
    double* p = prep + kPrepHead;
    for (int k = 0; k < n_ps; ++k, r += kRowPs, p += kPrepPs) peak = fmax(peak, prep_ps_block(r, p, ny, nx));
    for (int k = 0; k < n_sersic; ++k, r += kRowSersic, p += kPrepSersic)
        peak = fmax(peak, prep_sersic_block(r, p));
    prep_head(prep, row[0], psf_base + clamp_psf_index(r[0], n_psf), peak, rho);
}

// ---------------------------------------------------------------------------
// Sersic surface brightness of one pixel, with the reference's 1-D
// pixel-centroid correction.  psfMC/ModelComponents/Sersic.py:73-96
// (coordinate_sq_radii), :98-134 (add_to_array), :136-153 (_normed_grad).
//   rho2 = |M (dx,dy)|^2 ; q = rho2 / (dx^2+dy^2) ; L = ln rho2
//   sb = exp(-kappa expm1(L p)) ; g = -2 kappa p exp(L (p - 1/2))
//   value = sb_eff * sb * (1 + g * (q/12 * g))
// At dx = dy = 0 the reference evaluates 0/0 -> NaN; so does this code.
// ---------------------------------------------------------------------------
struct SersicPar {
    double x0, y0, m00, m01, m10, m11, kappa, p, sbeff;
};

__device__ inline SersicPar load_sersic(const double* __restrict__ s) {
    return SersicPar{s[0], s[1], s[2], s[3], s[4], s[5], s[6], s[7], s[8]};
}

__device__ inline double sersic_pixel(const SersicPar& s, double x, double y) {
    const double dx = x - s.x0, dy = y - s.y0;
    const double u = s.m00 * dx + s.m01 * dy;
    const double v = s.m10 * dx + s.m11 * dy;
    const double rho2 = u * u + v * v;
    const double q = rho2 / (dx * dx + dy * dy);
    const double L = log(rho2);
    const double sb = exp(-s.kappa * expm1(L * s.p));
    const double g = -s.kappa * 2.0 * s.p * exp(L * (s.p - 0.5));
    return s.sbeff * sb * (1.0 + g * (q / 12.0 * g));
}

// point-source contribution at integer pixel (ix, iy) from one prep PS block
__device__ inline double ps_pixel(const double* __restrict__ p, int ix, int iy) {
    const int ty = iy - (int)p[0];
    const int tx = ix - (int)p[2];
    if (ty < 0 || ty >= (int)p[1] || tx < 0 || tx >= (int)p[3]) return 0.0;
    return p[4 + ty] * p[4 + kTaps + tx];
}

// raw model value of one pixel: models.py:245-253 (Sky, PointSource..., Sersic...)
__device__ inline double raster_pixel(const double* __restrict__ prep, int n_ps, int n_sersic,
                                      int ix, int iy, bool ps_only) {
    double val = ps_only ? 0.0 : prep[0];
    const double* p = prep + kPrepHead;
    for (int k = 0; k < n_ps; ++k, p += kPrepPs) val += ps_pixel(p, ix, iy);
    if (!ps_only) {
        const double x = (double)ix, y = (double)iy;
        for (int k = 0; k < n_sersic; ++k, p += kPrepSersic)
            val += sersic_pixel(load_sersic(p), x, y);
    }
    return val;
}

// ---------------------------------------------------------------------------
// fp64 elementary functions tuned for the rasteriser (gfx950 has no fp64
// transcendental hardware beyond v_rcp/v_rsq seeds; the OCML routines cost
// ~25-40 instructions each because they cover the full IEEE domain).  These
// versions are accurate to a few ulp on the domain the rasteriser feeds them and
// keep NaN as NaN.  tests/test_gpu_parity.py::test_device_math checks them
// against numpy through psfmc_debug_math().
// ---------------------------------------------------------------------------
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-x, r, 1.0);
    return __builtin_fma(r, e, r);
}

// One Newton step on v_rcp_f64: ~2^-50 relative, for terms that are themselves small
// corrections (the rasteriser's centroid term).
__device__ __forceinline__ double fast_rcp1(double x) {
    const double r = __builtin_amdgcn_rcp(x);
    return __builtin_fma(r, __builtin_fma(-x, r, 1.0), r);
}

// log2(x), x > 0 finite normal.  x = 2^e m, m in [sqrt(1/2), sqrt(2));
// log2(m) = (2/ln2) atanh(s), s = (m-1)/(m+1), |s| <= 0.1716: s times a degree-7
// near-minimax polynomial in z = s^2 (Chebyshev interpolant of (2/ln2) atanh(sqrt z)/sqrt z
// on [0, 0.02944] computed at 50 digits; truncation 1.2e-18 relative).  Full relative
// accuracy near x = 1.  (A 128-entry table + degree-5 polynomial form needs 13 VALU
// instructions instead of 27 but one divergent 16-byte load per call: measured, the
// rasteriser got SLOWER with it, 41.2 vs 37.6 us, and the step did not change.)
__device__ __forceinline__ double fast_log2(double x) {
    int e = __builtin_amdgcn_frexp_exp(x);
    double m = __builtin_amdgcn_frexp_mant(x);            // [0.5, 1)
    const bool lo = m < 0.70710678118654752440;
    m = lo ? m + m : m;
    e = lo ? e - 1 : e;
    const double f = m - 1.0;
    const double s = f * fast_rcp(2.0 + f);
    const double z = s * s;
    double p = 2.13661225027983265e-01;
    p = __builtin_fma(p, z, 2.20912867341801572e-01);
    p = __builtin_fma(p, z, 2.62334360603295458e-01);
    p = __builtin_fma(p, z, 3.20598534765048127e-01);
    p = __builtin_fma(p, z, 4.12198585842294185e-01);
    p = __builtin_fma(p, z, 5.77078016345514033e-01);
    p = __builtin_fma(p, z, 9.61796693925989765e-01);
    p = __builtin_fma(p, z, 2.88539008177792677e+00);
    return __builtin_fma(s, p, (double)e);
}

// log2(x) for the rasteriser, x > 0 finite: x = 2^e m, m in [1/2, 1); the top eight fraction bits
// of m pick {a_j, b_j} from a 256-entry table in LDS with |m a_j - 1| <= 2^-9 and b_j = -log2(a_j):
//   log2(x) = e + b_j + log2(1 + r),  r = fma(m, a_j, -1)  (one rounding)
// 13 instructions and one 16-byte LDS read against the 27 instructions of fast_log2.  ABSOLUTE
// accuracy ~1e-16 + one rounding of the sum, which is what its only consumer, 2^(p log2 rho^2),
// needs (no relative accuracy near x = 1).  (The same table read from global memory made the
// rasteriser slower, see fast_log2.)  psfmc_log_table.h is generated by tools/gen_log_table.py.
constexpr int kLogTabBytes = 256 * 16;
// copy the table into a wave's LDS (lanes 0..63 bring four entries each); converged call
__device__ __forceinline__ void load_log_table(double* __restrict__ lds, int lane) {
    const double2* src = reinterpret_cast<const double2*>(&kLog2Tab[0][0]);
    double2* dst = reinterpret_cast<double2*>(lds);
    const double2 a = src[lane], b = src[lane + 64], c = src[lane + 128], d = src[lane + 192];
    dst[lane] = a;
    dst[lane + 64] = b;
    dst[lane + 128] = c;
    dst[lane + 192] = d;
}
__device__ __forceinline__ double fast_log2_tab(double x, const double* __restrict__ tab) {
    const int e = __builtin_amdgcn_frexp_exp(x);
    const double m = __builtin_amdgcn_frexp_mant(x);            // [0.5, 1)
    const unsigned off = ((unsigned)__double2hiint(m) >> 8) & 0xff0u;   // 16 j
    const double2 ab = *reinterpret_cast<const double2*>(reinterpret_cast<const char*>(tab) + off);
    const double r = __builtin_fma(m, ab.x, -1.0);
    double p = kLog2Poly[4];
    p = __builtin_fma(p, r, kLog2Poly[3]);
    p = __builtin_fma(p, r, kLog2Poly[2]);
    p = __builtin_fma(p, r, kLog2Poly[1]);
    p = __builtin_fma(p, r, kLog2Poly[0]);
    return __builtin_fma(r, p, (double)e + ab.y);
}

// ---------------------------------------------------------------------------
// t = (rho^2)^p without a logarithm or an exponential per pixel (round 3).  With rho^2 = 2^e m, m in
// [1/2, 1), and the mantissa table's a_j (|m a_j - 1| = |r| <= 2^-9, b_j = -log2 a_j):
//     (rho^2)^p = 2^(p e) * 2^(p b_j) * (1 + r)^p = PE[e] * PB[j] * (1 + r (q1 + r (q2 + ... + r q5)))
// q_k = binomial(p, k) (truncation binomial(p, 6) 2^-54: 1e-18 for n >= 1/4, 1e-14 at n = 0.05 where
// kappa ~ 1e-3 scales it away), and the two tables depend on the walker's p only: k_pow_tables writes
// them behind the prep record, a row wave copies them into LDS once per component (3 KB for its
// 1024 ... 2048 pixels).  14 vector instructions where table log2 + multiply + exp2 were 31; the three
// roundings of PE PB (1 + d) replace the rounding of p log2(rho^2), which grew with |log2 rho^2|.
// LDS layout of a wave's rasteriser region (kRasterLdsDoubles): [2j] = a_j, [2j + 1] = PB[j] (one
// 16-byte read per pixel, as the log2 table had), then PE[256].
// ---------------------------------------------------------------------------
constexpr int kRasterLdsDoubles = 2 * kPowTabB + kPowTabE;
// the static half: a_j into the even slots (converged call, once per wave)
__device__ __forceinline__ void load_a_table(double* __restrict__ lds, int lane) {
    const double a0 = kLog2Tab[lane][0], a1 = kLog2Tab[lane + 64][0], a2 = kLog2Tab[lane + 128][0],
                 a3 = kLog2Tab[lane + 192][0];
    lds[2 * lane] = a0;
    lds[2 * (lane + 64)] = a1;
    lds[2 * (lane + 128)] = a2;
    lds[2 * (lane + 192)] = a3;
}
// one component's tables (global, behind the walker's prep record) into the odd slots and the PE part
__device__ __forceinline__ void load_pow_table(double* __restrict__ lds, const double* __restrict__ g, int lane) {
    double b[4], e[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        b[i] = g[lane + 64 * i];
        e[i] = g[kPowTabB + lane + 64 * i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        lds[2 * (lane + 64 * i) + 1] = b[i];
        lds[2 * kPowTabB + lane + 64 * i] = e[i];
    }
}
// 2^(p x) with the product carried in two pieces (hi + lo exactly p x): 2^hi from OCML's exp2 (< 1 ulp; with
// the rasteriser's own 2-ulp exp2 the tables' errors -- shared by every pixel of a mantissa cell -- showed in the
// weight map of a 6e3-count peak, test_general_sides_match_oracle[400x120]) times 1 + lo ln 2.  The same
// function serves k_pow_tables and the row waves that build their own entries (small batches): same bits.
// |p x| beyond the double range of 2^y saturates (entries no pixel reads).
__device__ __forceinline__ double pow2_product(double p, double x) {
    const double hi = p * x;
    const double lo = __builtin_fma(p, x, -hi);
    const double v = exp2(hi);
    const double c = __builtin_fma(v, lo * 0.69314718055994530942, v);
    return (v > 0.0 && v < 1e300 && lo == lo) ? c : v;
}
// entry i of the 8 a lane owns (i < 4: PB[lane + 64 i], else PE[lane + 64 (i - 4)])
__device__ __forceinline__ double pow_tab_entry(double p, int lane, int i) {
    const int j = lane + 64 * (i & 3);
    return pow2_product(p, i < 4 ? kLog2Tab[j][1] : (double)(j - kPowTabEBias));
}
// the power tables of one (walker, component): all 64 lanes of a wave, 8 entries each
__device__ inline void build_pow_table(double p, double* __restrict__ g, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        g[lane + 64 * i] = pow_tab_entry(p, lane, i);
        g[kPowTabB + lane + 64 * i] = pow_tab_entry(p, lane, 4 + i);
    }
}
struct PowPoly { double q1, q2, q3, q4, q5; };
__device__ __forceinline__ PowPoly pow_poly(double p) {
    PowPoly q;
    q.q1 = p;
    q.q2 = q.q1 * (p - 1.0) * 0.5;
    q.q3 = q.q2 * (p - 2.0) * 0.33333333333333333333;
    q.q4 = q.q3 * (p - 3.0) * 0.25;
    q.q5 = q.q4 * (p - 4.0) * 0.2;
    return q;
}
// (rho^2)^p from the wave's LDS tables; x >= 0 finite (x = 0: some finite value -- the pixel is NaN
// through the centroid term's reciprocal, as the reference's 0/0)
__device__ __forceinline__ double fast_pow_tab(double x, const PowPoly& q, const double* __restrict__ tab) {
    const int e = __builtin_amdgcn_frexp_exp(x);
    const double m = __builtin_amdgcn_frexp_mant(x);            // [0.5, 1)
    const unsigned off = ((unsigned)__double2hiint(m) >> 8) & 0xff0u;   // 16 j
    const double2 ab = *reinterpret_cast<const double2*>(reinterpret_cast<const char*>(tab) + off);
    const int ei = min(max(e, -kPowTabEBias), kPowTabE - kPowTabEBias - 1);      // v_med3_i32
    const double pe = tab[2 * kPowTabB + kPowTabEBias + ei];
    const double r = __builtin_fma(m, ab.x, -1.0);
    double d = q.q5;
    d = __builtin_fma(d, r, q.q4);
    d = __builtin_fma(d, r, q.q3);
    d = __builtin_fma(d, r, q.q2);
    d = __builtin_fma(d, r, q.q1);
    d *= r;
    const double eb = pe * ab.y;
    return __builtin_fma(eb, d, eb);
}

#ifndef PSFMC_RASTER_RCP_PAIRS
#define PSFMC_RASTER_RCP_PAIRS 0
#endif
// 2^y for finite |y| (any magnitude: ldexp saturates); no inf handling.
__device__ __forceinline__ double fast_exp2_poly(double r);
__device__ __forceinline__ double fast_exp2_noclamp(double y) {
    const double n = __builtin_rint(y);
    return __builtin_amdgcn_ldexp(fast_exp2_poly(y - n), (int)n);
}

// 2^y with y clamped to [-1100, 1100] (so -inf / +inf give 0 / inf).  The clamp is a
// v_max / v_min pair, which returns the bound for a NaN argument: NaN comes back as 0.  In
// the rasteriser a NaN also reaches the result through the centroid term, so the
// brightness factor need not carry it.
__device__ __forceinline__ double fast_exp2(double y) {
    y = __builtin_fmin(__builtin_fmax(y, -1100.0), 1100.0);
    const double n = __builtin_rint(y);
    return __builtin_amdgcn_ldexp(fast_exp2_poly(y - n), (int)n);
}

// 2^y for the rasteriser's brightness factor: y = kappa log2(e) (1 - t) is bounded above by the walker's
// kappa log2(e) (<= a few hundred: the result may legitimately overflow to inf like the reference's exp),
// so only the lower clamp is needed (-inf and NaN give 0 as in fast_exp2)
__device__ __forceinline__ double fast_exp2_floor(double y) {
    y = __builtin_fmax(y, -1100.0);
    const double n = __builtin_rint(y);
    return __builtin_amdgcn_ldexp(fast_exp2_poly(y - n), (int)n);
}

// 2^r, |r| <= 1/2: degree-11 near-minimax polynomial (Chebyshev interpolant at 50 digits;
// truncation 3e-18, where the degree-12 Taylor series it replaces had 1.7e-16)
constexpr double kExp2Deg11[12] = {
    1.00000000000000000e+00, 6.93147180559945286e-01, 2.40226506959101582e-01, 5.55041086648216248e-02,
    9.61812910758725638e-03, 1.33335581464064708e-03, 1.54035304637243530e-04, 1.52527338415567733e-05,
    1.32154325359123753e-06, 1.01780570877339407e-07, 7.07419429728852106e-09, 4.45581790833606449e-10};
__device__ __forceinline__ double fast_exp2_poly(double r) {
    double p = kExp2Deg11[11];
#pragma unroll
    for (int i = 10; i >= 0; --i) p = __builtin_fma(p, r, kExp2Deg11[i]);
    return p;
}

// ---------------------------------------------------------------------------
// Row rasteriser of the fused path: the P pixels x = T k + t (k < P) of image row
// iy, for one lane.  Same sums as raster_pixel (sky, point sources, Sersics), but
//   * per-walker parameters are read through the wave-uniform `prep` pointer, so
//     they live in scalar registers instead of being re-read per pixel;
//   * point sources are skipped by whole waves whose rows miss the <= 7-row window;
//   * the Sersic profile shares its exponentials.  With t = rho2^p (= exp(L p)):
//         expm1(L p)       = t - 1
//         exp(L (p - 1/2)) = t / sqrt(rho2)
//     and the centroid term g (q/12 g) = (2 kappa p t)^2 / (12 (dx^2 + dy^2)) because the
//     rho2 of g^2 cancels the one of q = rho2 / (dx^2 + dy^2): a pixel costs one log2,
//     two exp2 and one reciprocal instead of log + expm1 + 2 exp + an IEEE division.  |t - 1| loses at most 1 ulp of t
//     absolute, which kappa (<~ 20) scales to <= 4e-15 relative in the brightness.
//     At dx = dy = 0 the reciprocal produces NaN like the reference's 0/0.
// (Sersic.py:98-134 + :136-153, PointSource.py:24-57, Sky.py:14-16.)
// ---------------------------------------------------------------------------
// An image side the transforms are not built for (a prime factor above 13, ...) is EMBEDDED in the next
// built side M >= L + Pk - 1 (L the image side, Pk the PSF's): the rasteriser fills transform pixel x' <
// L + Pk - 1 with model pixel (x' - a) mod L -- the image followed by a wrap-around margin of Pk - 1 pixels,
// zeros after it -- and the circular convolution of length M then equals the circular convolution of
// length L (utils.py:25-32) on the pixels [a, a + L), which are the only ones the likelihood looks at
// (overlap-save: every output there only reaches back over pixels that hold what the periodic image
// holds).  a = Pk - 1 - c, c = the kernel's origin inside the padded PSF (psfmc_hip.hip embed_axis).
// l = 0: no embedding on this axis.
struct WrapDesc {
    int lx, ax, ex;     // image side, margin before the image, extent of the filled pixels (lx + Pk - 1)
    int ly, ay, ey;
};
// the image's own [ly][lx] window at (ay, ax) of a transform-shaped [..][nx] pixel array (the whole of it
// unless the image is embedded)
struct ImgWindow { int nx, ly, lx, ay, ax; };
__device__ __forceinline__ int wrap_coord(int p, int a, int l) {
    int m = p - a;
    m += m < 0 ? l : 0;
    m -= m >= l ? l : 0;
    return m;
}

// K0: the lane's pixels are x = T (K0 + k) + t, k < P (a segment of a longer row; 0 for whole rows)
// WRAP: `iy` and x are transform coordinates of an embedded image (see WrapDesc)
// G: Sersic pixels per lane that go through the profile's stages together (1: pixel after pixel)
// PFETCH: a component's parameters and tables are fetched one component ahead (always with G > 1; the
// one-row-per-wave kernels of psfmc_rows3_path.h ask for it with G = 1 too)
template <int P, int T, int K0 = 0, bool WRAP = false, int G = 1, bool PFETCH = (G > 1)>
__device__ __forceinline__ void raster_row(const double* __restrict__ prep, int n_ps, int n_sersic,
                                           int t, int iy, bool ps_only, double* __restrict__ log_tab,
                                           double (&r)[P], const WrapDesc& wr = WrapDesc{0, 0, 0, 0, 0, 0},
                                           int pow_mode = kPowTabsBuilt) {
    const bool tabs_in_wave = pow_mode == kPowTabsInWave;
    const double sky = ps_only ? 0.0 : prep[0];
#pragma unroll
    for (int k = 0; k < P; ++k) r[k] = sky;
    int xm[WRAP ? P : 1];              // model column of pixel k
    bool live_row = true;
    if constexpr (WRAP) {
        live_row = iy < wr.ey;
        iy = wrap_coord(iy, wr.ay, wr.ly);
#pragma unroll
        for (int k = 0; k < P; ++k) xm[k] = wrap_coord(T * (K0 + k) + t, wr.ax, wr.lx);
    }
    const double* p = prep + kPrepHead;
    const double* const sersic0 = p + kPrepPs * n_ps;              // the Sersic blocks, then their power tables
    const double* const pow_tabs = sersic0 + kPrepSersic * n_sersic;   // (behind the record: prep_len)
    const int lane_id = (int)(threadIdx.x & 63);
    struct Block { double par, b[4], e[4]; };
    auto issue = [&](int c, Block& B) {
        B.par = sersic0[c * kPrepSersic + (lane_id < kPrepSersic ? lane_id : 0)];
        if (tabs_in_wave) {
            // small batches: no k_pow_tables launch, every row wave forms the 8 entries per lane it needs
            // (the same function, the same bits; ~400 instructions per component)
            const double pw = sersic0[c * kPrepSersic + 7];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                B.b[i] = pow_tab_entry(pw, lane_id, i);
                B.e[i] = pow_tab_entry(pw, lane_id, 4 + i);
            }
            return;
        }
        const double* g = pow_tabs + (size_t)c * kPowTab;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            B.b[i] = g[lane_id + 64 * i];
            B.e[i] = g[kPowTabB + lane_id + 64 * i];
        }
    };
    constexpr bool PF = PFETCH;
    Block cur{}, nxt{};
    if (PF && !ps_only && n_sersic > 0) issue(0, cur);            // in flight during the point sources
    for (int c = 0; c < n_ps; ++c, p += kPrepPs) {
        const int ty = iy - (int)p[0];
        const bool row_in = ty >= 0 && ty < (int)p[1];
        if (__any(row_in)) {
            const double wy = row_in ? p[4 + (row_in ? ty : 0)] : 0.0;
            const int xlo = (int)p[2], xn = (int)p[3];
#pragma unroll
            for (int k = 0; k < P; ++k) {
                const int tx = (WRAP ? xm[k] : T * (K0 + k) + t) - xlo;
                const bool in = (unsigned)tx < (unsigned)xn;
                const double wx = p[4 + kTaps + (in ? tx : 0)];
                r[k] += in ? wy * wx : 0.0;
            }
        }
    }
    // beyond the wrap-around margin of an embedded image the transform pixels are zero
    auto blank_margin = [&]() {
        if constexpr (WRAP) {
#pragma unroll
            for (int k = 0; k < P; ++k) r[k] = (live_row && T * (K0 + k) + t < wr.ex) ? r[k] : 0.0;
        }
    };
    if (ps_only) {
        blank_margin();
        return;
    }
    constexpr double kLog2e = 1.44269504088896340736;
    const double y = (double)iy;
    // PF: a component's parameters and power tables are PREFETCHED through vector loads while the previous
    // component's pixels are evaluated (lane i < 9 brings parameter i, every lane eight table entries), and
    // the parameters are then broadcast into scalar registers with v_readlane (scalar loads share lgkmcnt with
    // the LDS reads of the pixel loop and cannot be waited for separately).  Measured (same box, 2 ... 3
    // components): with the grouped pixel stages k_rows_fwd<300> 58.9 -> 52.5 us, <600> 54.3 -> 48.2, <768> 70 ->
    // 63; on the power-of-two kernels nothing (512 / 1024) or a loss (256: -2 % whole step), so they read each
    // component where they use it.
    auto bcast = [](double v, int src) {
        return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src),
                                __builtin_amdgcn_readlane(__double2loint(v), src));
    };
    for (int c = 0; c < n_sersic; ++c) {
        double x0, y0, m00, m01, m10, m11, kappa, pw, sbeff;
        if constexpr (PF) {
            if (c + 1 < n_sersic) issue(c + 1, nxt);       // in flight during this component's pixels
            x0 = bcast(cur.par, 0); y0 = bcast(cur.par, 1); m00 = bcast(cur.par, 2); m01 = bcast(cur.par, 3);
            m10 = bcast(cur.par, 4); m11 = bcast(cur.par, 5);
            kappa = bcast(cur.par, 6); pw = bcast(cur.par, 7); sbeff = bcast(cur.par, 8);
        } else {
            const double* sp = sersic0 + c * kPrepSersic;   // wave-uniform: scalar loads
            x0 = sp[0]; y0 = sp[1]; m00 = sp[2]; m01 = sp[3]; m10 = sp[4]; m11 = sp[5];
            kappa = sp[6]; pw = sp[7]; sbeff = sp[8];
            issue(c, cur);
        }
        const double dy = y - y0;
        const double uy = m01 * dy, vy = m11 * dy, dy2 = dy * dy;
        const double nkl = -kappa * kLog2e;                // sb = 2^(nkl (t - 1))
        // g = gk t / sqrt(rho2); the 1/12 of the centroid term rides on gk
        const double gk = -2.0 * kappa * pw * 0.28867513459481288225;   // sqrt(1/12)
        const PowPoly q = pow_poly(pw);
        wave_lds_sync();                                   // the previous component's reads are done
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            log_tab[2 * (lane_id + 64 * i) + 1] = cur.b[i];
            log_tab[2 * kPowTabB + lane_id + 64 * i] = cur.e[i];
        }
        wave_lds_sync();
        // G > 1 (general shapes above 256 pixels per row with more than 16 complex registers per lane: two waves per
        // SIMD whatever the rasteriser needs): the pixels go through the profile in GROUPS of G, stage by stage (every
        // stage for all pixels of the group before the next stage), and the table reads of the next group are
        // issued before the current group's arithmetic.  Written pixel after pixel (G = 1) the compiler keeps that
        // order: one dependent chain of ~45 instructions per pixel behind an LDS read it waits for at once (s_waitcnt
        // in the loop 73 -> 30 with G = 4).  By measurement (fused_path.h raster_group): -10 % kernel time on the
        // general shapes, nothing on the 512 / 1024 kernels (whose time is their waves' lifetimes).
        constexpr int kG = G, NG = (P + kG - 1) / kG;
        struct Fetched { double m[kG], d2[kG], pe[kG]; double2 ab[kG]; };
        auto fetch = [&](int g, Fetched& F) {
#pragma unroll
            for (int j = 0; j < kG; ++j) {
                const int k = g * kG + j;
                if (k < P) {
                    const double dx = (double)(WRAP ? xm[k] : T * (K0 + k) + t) - x0;   // exact pixel coordinate, one rounding
                    const double u = __builtin_fma(m00, dx, uy);
                    const double v = __builtin_fma(m10, dx, vy);
                    const double rho2 = __builtin_fma(u, u, v * v);
                    F.d2[j] = __builtin_fma(dx, dx, dy2);
                    const int e = __builtin_amdgcn_frexp_exp(rho2);
                    F.m[j] = __builtin_amdgcn_frexp_mant(rho2);                         // [0.5, 1)
                    const unsigned off = ((unsigned)__double2hiint(F.m[j]) >> 8) & 0xff0u;   // 16 j
                    F.ab[j] = *reinterpret_cast<const double2*>(reinterpret_cast<const char*>(log_tab) + off);
                    const int ei = min(max(e, -kPowTabEBias), kPowTabE - kPowTabEBias - 1);  // v_med3_i32
                    F.pe[j] = log_tab[2 * kPowTabB + kPowTabEBias + ei];
                }
            }
        };
        auto finish = [&](int g, const Fetched& F) {
            double rr[kG], d[kG], tt[kG], yy[kG], nn[kG], sb[kG], rc[kG];
            auto each = [&](auto&& fn) {
#pragma unroll
                for (int j = 0; j < kG; ++j)
                    if (g * kG + j < P) fn(j);
            };
            // t = rho2^p = PE PB (1 + r (q1 + r (q2 + ...)))      (fast_pow_tab)
            each([&](int j) { rr[j] = __builtin_fma(F.m[j], F.ab[j].x, -1.0); d[j] = __builtin_fma(q.q5, rr[j], q.q4); });
            each([&](int j) { d[j] = __builtin_fma(d[j], rr[j], q.q3); });
            each([&](int j) { d[j] = __builtin_fma(d[j], rr[j], q.q2); });
            each([&](int j) { d[j] = __builtin_fma(d[j], rr[j], q.q1); });
            each([&](int j) { d[j] *= rr[j]; tt[j] = F.pe[j] * F.ab[j].y; });
            each([&](int j) { tt[j] = __builtin_fma(tt[j], d[j], tt[j]); });
            // sb = 2^(nkl (t - 1))                                 (fast_exp2_floor)
            each([&](int j) { yy[j] = __builtin_fmax(__builtin_fma(nkl, tt[j], -nkl), -1100.0); nn[j] = __builtin_rint(yy[j]); });
            each([&](int j) { yy[j] -= nn[j]; sb[j] = __builtin_fma(kExp2Deg11[11], yy[j], kExp2Deg11[10]); });
#pragma unroll
            for (int i = 9; i >= 0; --i) each([&](int j) { sb[j] = __builtin_fma(sb[j], yy[j], kExp2Deg11[i]); });
#if PSFMC_RASTER_RCP_PAIRS
            // experiment: ONE reciprocal per PAIR of pixels -- 1 / (a b), then 1 / a = b / (a b): v_rcp_f64 issues for
            // four instructions, so a pair costs mul + rcp (4) + Newton (2) + 2 mul = 9 slots instead of 12
            if constexpr (kG % 2 == 0) {
                each([&](int j) { sb[j] = __builtin_amdgcn_ldexp(sb[j], (int)nn[j]); tt[j] *= gk; });
#pragma unroll
                for (int j = 0; j < kG; j += 2)
                    if (g * kG + j + 1 < P) {
                        const double ab = F.d2[j] * F.d2[j + 1];
                        double rr2 = __builtin_amdgcn_rcp(ab);
                        rr2 = __builtin_fma(rr2, __builtin_fma(-ab, rr2, 1.0), rr2);
                        rc[j] = rr2 * F.d2[j + 1];
                        rc[j + 1] = rr2 * F.d2[j];
                    } else if (g * kG + j < P) {
                        rc[j] = __builtin_amdgcn_rcp(F.d2[j]);
                        rc[j] = __builtin_fma(rc[j], __builtin_fma(-F.d2[j], rc[j], 1.0), rc[j]);
                    }
            } else
#endif
            {
            each([&](int j) { sb[j] = __builtin_amdgcn_ldexp(sb[j], (int)nn[j]); rc[j] = __builtin_amdgcn_rcp(F.d2[j]); });
            // 1 / d2 with one Newton step (fast_rcp1); g^2 q / 12 = (gk t)^2 / rho2 * rho2 / d2: the elliptical
            // radius cancels
            each([&](int j) { rc[j] = __builtin_fma(rc[j], __builtin_fma(-F.d2[j], rc[j], 1.0), rc[j]); tt[j] *= gk; });
            }
            each([&](int j) { tt[j] *= tt[j]; sb[j] *= sbeff; });
            each([&](int j) { r[g * kG + j] = __builtin_fma(sb[j], __builtin_fma(tt[j], rc[j], 1.0), r[g * kG + j]); });
        };
        Fetched F[2];
        if constexpr (kG == 1) {
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                fetch(g, F[0]);
                finish(g, F[0]);
            }
        } else {
            fetch(0, F[0]);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g + 1 < NG) fetch(g + 1, F[(g + 1) & 1]);
                finish(g, F[g & 1]);
                __builtin_amdgcn_sched_barrier(0);          // two groups in flight, not more (registers)
            }
        }
        if constexpr (PF) cur = nxt;
    }
    blank_margin();
}

// Round 2's form, unchanged (rows of up to 256 pixels: pow_tabs_side): log2 + exp2 per Sersic pixel, the log2
// through the {a_j, b_j} table in LDS (load_log_table).
// K0: the lane's pixels are x = T (K0 + k) + t, k < P (a segment of a longer row; 0 for whole rows)
// WRAP: `iy` and x are transform coordinates of an embedded image (see WrapDesc)
template <int P, int T, int K0 = 0, bool WRAP = false>
__device__ __forceinline__ void raster_row_logexp(const double* __restrict__ prep, int n_ps, int n_sersic,
                                           int t, int iy, bool ps_only, const double* __restrict__ log_tab,
                                           double (&r)[P], const WrapDesc& wr = WrapDesc{0, 0, 0, 0, 0, 0}) {
    const double sky = ps_only ? 0.0 : prep[0];
#pragma unroll
    for (int k = 0; k < P; ++k) r[k] = sky;
    int xm[WRAP ? P : 1];              // model column of pixel k
    bool live_row = true;
    if constexpr (WRAP) {
        live_row = iy < wr.ey;
        iy = wrap_coord(iy, wr.ay, wr.ly);
#pragma unroll
        for (int k = 0; k < P; ++k) xm[k] = wrap_coord(T * (K0 + k) + t, wr.ax, wr.lx);
    }
    const double* p = prep + kPrepHead;
    for (int c = 0; c < n_ps; ++c, p += kPrepPs) {
        const int ty = iy - (int)p[0];
        const bool row_in = ty >= 0 && ty < (int)p[1];
        if (__any(row_in)) {
            const double wy = row_in ? p[4 + (row_in ? ty : 0)] : 0.0;
            const int xlo = (int)p[2], xn = (int)p[3];
#pragma unroll
            for (int k = 0; k < P; ++k) {
                const int tx = (WRAP ? xm[k] : T * (K0 + k) + t) - xlo;
                const bool in = (unsigned)tx < (unsigned)xn;
                const double wx = p[4 + kTaps + (in ? tx : 0)];
                r[k] += in ? wy * wx : 0.0;
            }
        }
    }
    // beyond the wrap-around margin of an embedded image the transform pixels are zero
    auto blank_margin = [&]() {
        if constexpr (WRAP) {
#pragma unroll
            for (int k = 0; k < P; ++k) r[k] = (live_row && T * (K0 + k) + t < wr.ex) ? r[k] : 0.0;
        }
    };
    if (ps_only) {
        blank_margin();
        return;
    }
    constexpr double kLog2e = 1.44269504088896340736;
    const double y = (double)iy;
    for (int c = 0; c < n_sersic; ++c, p += kPrepSersic) {
        const double x0 = p[0], y0 = p[1], m00 = p[2], m01 = p[3], m10 = p[4], m11 = p[5];
        const double kappa = p[6], pw = p[7], sbeff = p[8];
        const double dy = y - y0;
        const double uy = m01 * dy, vy = m11 * dy, dy2 = dy * dy;
        const double nkl = -kappa * kLog2e;                // sb = 2^(nkl (t - 1))
        // g = gk t / sqrt(rho2); the 1/12 of the centroid term rides on gk
        const double gk = -2.0 * kappa * pw * 0.28867513459481288225;   // sqrt(1/12)
#pragma unroll
        for (int k = 0; k < P; ++k) {
            const double dx = (double)(WRAP ? xm[k] : T * (K0 + k) + t) - x0;   // exact pixel coordinate, one rounding
            const double u = __builtin_fma(m00, dx, uy);
            const double v = __builtin_fma(m10, dx, vy);
            const double rho2 = __builtin_fma(u, u, v * v);
            const double d2 = __builtin_fma(dx, dx, dy2);
            const double tt = fast_exp2_noclamp(pw * fast_log2_tab(rho2, log_tab));
            const double sb = fast_exp2_floor(__builtin_fma(nkl, tt, -nkl));
            // g^2 q / 12 = (gk t)^2 / rho2 * rho2 / d2: the elliptical radius cancels
            const double gt = gk * tt;
            r[k] = __builtin_fma(sbeff * sb, __builtin_fma(gt * gt, fast_rcp1(d2), 1.0), r[k]);
        }
    }
    blank_margin();
}

// ---------------------------------------------------------------------------
// Gaussian chi^2 + log-normalisation term of one good pixel, models.py:233-236:
//   resid^2 * ivm - ln(0.5/pi * ivm),  ivm = 1/(model_var + obs_var) (:278-279)
// ---------------------------------------------------------------------------
__device__ inline double chi2_term(double sci, double obs_var, double conv, double mvar) {
    const double ivm = 1.0 / (mvar + obs_var);
    const double r = sci - conv;
    return r * r * ivm - log(0.5 / M_PI * ivm);
}

// wave64 + LDS block sum; result valid in thread 0.  blockDim.x multiple of 64.
__device__ inline double block_sum(double v, double* lds /* >= blockDim/64 doubles */) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) lds[wave] = v;
    __syncthreads();
    double total = 0.0;
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        for (int i = 0; i < nw; ++i) total += lds[i];
    }
    return total;
}

// Sum of one walker's nblk partial chi^2 sums by the 64 lanes of a wave: lane-strided
// accumulation, then a butterfly, so every lane holds the total.  ONE summation order
// for every finishing kernel (k_finish, k_finish_posterior, k_stretch_finish): the
// host-loop and the device-resident samplers must see the same bits.  (One thread per
// walker walked nblk dependent strided loads: 13 us for a 128-walker half-ensemble.)
__device__ inline double wave_sum_partials(const double* __restrict__ p, int nblk, int lane) {
    double s = 0.0;
    for (int i = lane; i < nblk; i += 64) s += p[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    return s;
}
constexpr int kFinishThreads = 256;                       // 4 waves = 4 walkers per workgroup
__host__ inline int finish_blocks(int W) { return (W + kFinishThreads / 64 - 1) / (kFinishThreads / 64); }

}  // namespace psfmc
