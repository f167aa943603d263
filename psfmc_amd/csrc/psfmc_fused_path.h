// psfmc_fused_path.h -- the PSFMC_BACKEND_FUSED kernels: three launches per
// batch of walkers, no stand-alone FFT passes, no workgroup barriers.
//
//   rows_fwd   each WAVE rasterises its group of RG image rows straight into
//              registers as z = raw + i mu raw^2 (two real images in one complex
//              signal), transforms along x, untangles the two Hermitian spectra
//              and stores the kx <= nx/2 half
//   cols       per (walker, kx): the two columns (c = 0 model, c = 1 variance),
//              transform along y, multiply by the pre-scaled, pre-shifted kernel
//              spectrum Kt[psf][kx][c][ky], inverse transform along y, in place
//   rows_inv   each wave reloads its RG rows (all kx), rebuilds the full complex
//              spectrum Y = G + i H (G, H Hermitian in kx), inverse transform
//              along x: real part = PSF-convolved model, imaginary part =
//              lambda * model variance; fused chi^2 + log term + masked reduction
//              -> one partial sum per wave.
//
// Intermediate layout ("T"): [walker][kx][yg][c][r], y = yg*RG + r, RG = 64/T(nx)
// rows per wave.  The RG rows x {model, variance} of one kx are contiguous
// (128 B at nx = 256), so a row wave touches whole 64-B halves of cache lines
// with the lanes already in transform order (no LDS transposition), and a column
// pair (kx) is one contiguous run of 2*ny complex for the column kernel.
//
// HBM traffic per walker: one write + one read/write + one read of T
// (2*(nx/2+1)*ny complex128 = 1.03 MB at 256^2) against 6.3 MB of "algorithmic"
// bytes for the unfused arrangement.
//
// Reference: psfMC/models.py:213-216, 233-236; utils.py:25-32 (convolve: the
// ifftshift is the (-1)^(kx+ky) sign folded into Kt, as is 1/(nx*ny)).
#pragma once
#ifndef PSFMC_PART
#define PSFMC_PART 0      /* single translation unit (see psfmc_hip.hip) */
#endif
#include "psfmc_device.h"
#include "psfmc_fft.h"

#ifndef PSFMC_COLS3G_MIN
#define PSFMC_COLS3G_MIN 0        /* sides above this that psfmc_fft.h fft3g_pick lists run their columns on the general three-stage engine */
#endif

namespace psfmc {

// Autonomous waves per row-kernel workgroup (they take consecutive row groups).  One is
// enough while a wave's RG rows x {model, variance} fill a 128-B line per kx (RG >= 4:
// nx <= 512; measured no difference between 1, 2 and 4).  At nx = 1024 a wave holds only
// 2 rows (64-B pieces): four waves of a workgroup then share each line in one CU's L1/L2
// (k_rows_inv<1024>: 130 -> 105 us).
#ifndef PSFMC_ROW_WAVES
#define PSFMC_ROW_WAVES 0          /* 0 = choose per shape */
#endif
constexpr int kColThreads = 256;     // column kernel: 4 waves, persistent
#ifndef PSFMC_COLS_PRIO
#define PSFMC_COLS_PRIO 3     /* measured: 512^2 +1.7 %, 1024^2 +0.8 %, 256^2 unchanged */
#endif
#ifndef PSFMC_INV_PRIO
#define PSFMC_INV_PRIO 0
#endif
#ifndef PSFMC_INV_REMAP_MIN
#define PSFMC_INV_REMAP_MIN 720       /* unguarded k_rows_inv of sides from this on (and every guarded one): row groups in XCD-sized batches over all walkers */
#endif
template <int NX, bool FAST> constexpr bool inv_remap() { return !FAST || NX >= PSFMC_INV_REMAP_MIN; }
// walkers per block of that order: all of a launch up to this many, else 8
constexpr int kInvRemapMaxWalkers = 24;
#ifndef PSFMC_INV_CHUNK
#define PSFMC_INV_CHUNK 8             /* field pixels per load chunk of k_rows_inv at nx = 512, 1024 */
#endif
#ifndef PSFMC_DEBUG_FWD
#define PSFMC_DEBUG_FWD 0            /* timing experiments on k_rows_fwd: 1 = no store phase, 2 = no transform */
#endif
#ifndef PSFMC_FWD_STORE_BARRIER
#define PSFMC_FWD_STORE_BARRIER 0
#endif
#ifndef PSFMC_FWD_PRIO_STAGGER
#define PSFMC_FWD_PRIO_STAGGER 0      /* n > 0: forward row workgroups take issue priority (linear id / n) & 3 */
#endif
#ifndef PSFMC_COLS_PREFETCH
#define PSFMC_COLS_PREFETCH 1         /* register double-buffering of the column loads */
#endif

template <int NX> constexpr int row_group() { return FftShape<NX>::TPW; }         // rows per wave
// Sersic pixels per lane that the forward kernel's rasteriser takes through the profile together (raster_row's
// G), by measurement (same box, 2 ... 4 components): 4 for the general shapes that hold more than 16 complex
// registers per lane and run at two waves per SIMD whatever the rasteriser needs (k_rows_fwd<300> 58.9 -> 52.5 us,
// <600> 54.3 -> 48.2, <768> 70 -> 63: whole step +2 ... +5 %); 1 for the power-of-two shapes (512 / 1024: 43.5 vs
// 47.1 us and 66.7 vs 68.9 us -- the grouped form spills there -- and the small ones live on four waves per SIMD)
// and for three sides that the grouped form takes from three waves per SIMD to two (264, 286, 294: +5 % kernel time)
#ifndef PSFMC_RASTER_GROUP
#define PSFMC_RASTER_GROUP 4
#endif
#ifndef PSFMC_RASTER_GROUP_PLAIN
#define PSFMC_RASTER_GROUP_PLAIN 2    /* the 512 / 1024 kernels: two pixels per group (236 registers, no spills).  Round 4, same box, three
                                         alternating runs (profiles/r4_g2_step_ab.txt): kernel alone 66.7 -> 62.7 us at 1024^2, 44.3 -> 42.5 at
                                         512^2; whole step +0.7 % / +1.2 % (51.82 -> 52.17 k, 279.0 -> 282.2 k evals/s): small, consistent, taken
                                         (round 3 read the same numbers as noise) */
#endif
template <int NX> constexpr int raster_group() {
    if (FftShape<NX>::kPlain && NX >= 512) return PSFMC_RASTER_GROUP_PLAIN;
    return (pow_tabs_side(NX) && !FftShape<NX>::kPlain && FftShape<NX>::R > 16 && NX != 264 && NX != 286 && NX != 294) ? PSFMC_RASTER_GROUP : 1;
}
// rows that share a contiguous run of T per kx (the "RG" of the layout comment above): the rows
// of one wave for the power-of-two shapes, 4 otherwise (ny is rounded up to a multiple of it in
// the layout; the spare rows are never read)
// FAST = the unguarded row kernels of the power-of-two shapes (whole workgroups of rows only);
// the same shapes also build with FAST = false -- the guarded general code path -- for images whose
// ny is not a whole number of such workgroups
template <int NX, bool FAST = FftShape<NX>::kPlain> constexpr int layout_rg_log2() {
    return FAST ? __builtin_ctz(FftShape<NX>::TPW) : 2;
}
// General shapes: a wave's RG rows need not be a whole number of the layout's 4-row groups
// (RG = 6 for T = 10, 3 for T = 20, 5 for T = 12): as many consecutive waves as make one (12 or
// 20 rows) share a workgroup, so that the second touch of a straddled cache line comes from the
// same CU at about the same time (k_rows_inv<300> fetched 1.39x its bytes with one wave per
// workgroup, profiles/r2b_pmc_300_*).
#ifndef PSFMC_GEN_ROW_WAVES
#define PSFMC_GEN_ROW_WAVES 1
#endif
template <int NX, bool FAST = FftShape<NX>::kPlain> constexpr int row_waves() {
    if (PSFMC_ROW_WAVES) return PSFMC_ROW_WAVES;
    if (FAST) return row_group<NX>() >= 4 ? 1 : 4;
    constexpr int rg = row_group<NX>();
    return !PSFMC_GEN_ROW_WAVES ? 1 : (rg % 4 == 0 ? 1 : rg % 2 == 0 ? 2 : 4);
}
template <int NX, bool FAST = FftShape<NX>::kPlain> constexpr int row_threads() { return 64 * row_waves<NX, FAST>(); }
// per wave: the exchange regions of its RG transforms, then its twiddle table
// (at least the rasteriser's tables, which borrow the region before the transform starts: psfmc_device.h)
template <int NX> constexpr size_t fused_row_wave_lds_doubles() {
    constexpr size_t fft = (size_t)row_group<NX>() * fft_lds_elems<NX>() + 2 * (size_t)fft_tw_lds_elems<NX>();
    constexpr size_t ras = pow_tabs_side(NX) ? (size_t)kRasterLdsDoubles : (size_t)kLogTabBytes / sizeof(double);
    return fft > ras ? fft : ras;
}
template <int NX, bool FAST = FftShape<NX>::kPlain> constexpr size_t fused_row_lds_bytes() {
    return (size_t)row_waves<NX, FAST>() * fused_row_wave_lds_doubles<NX>() * sizeof(double);
}
template <int NY> constexpr int col_ffts_per_block() { return (kColThreads / 64) * FftShape<NY>::TPW; }
template <int NY> constexpr size_t fused_col_wave_lds_doubles() {
    return (size_t)FftShape<NY>::TPW * fft_lds_elems<NY>() + 2 * (size_t)fft_tw_lds_elems<NY, PSFMC_TW_MODE_COLS>();
}
template <int NY> constexpr size_t fused_col_lds_bytes() {
    return (size_t)(kColThreads / 64) * fused_col_wave_lds_doubles<NY>() * sizeof(double);
}
// waves per SIMD the register allocator must leave room for
#ifndef PSFMC_GEN_R_2WAVES
#define PSFMC_GEN_R_2WAVES 16     /* general shapes with up to this many registers are compiled for 2 waves per SIMD
                                     (20 spills: k_cols<300> 102 us instead of 77, k_cols<320> 88 instead of 53) */
#endif
#ifndef PSFMC_GEN_PF_MAX_R
#define PSFMC_GEN_PF_MAX_R 20        /* general shapes with up to this many registers per lane prefetch the next column */
#endif
#ifndef PSFMC_GEN_STAGED
#define PSFMC_GEN_STAGED 1
#endif
#ifndef PSFMC_PLAIN_COL_WAVES
#define PSFMC_PLAIN_COL_WAVES 2
#endif
template <int N> constexpr int fused_min_waves() {
    return FftShape<N>::kPlain ? (FftShape<N>::R > 16 ? 1 : 2) : (FftShape<N>::R > PSFMC_GEN_R_2WAVES ? 1 : 2);
}
// The row kernels' bound for the general shapes.  Left to itself (a bound of one wave per SIMD) the register
// allocator took 1 ... 50 accumulation registers on top of the 256 vector registers at nine sides, which halves
// the waves per SIMD for the sake of a handful of values.  Bounded to two waves, 650 / 700 / 720 spill 4 ... 10
// registers and run 20 ... 24 % faster (whole step, same box: 94 -> 117 k, 88 -> 105 k, 94 -> 113 k evals/s), 676
// (18 spilled) +9 %, 780 (29) +3 %; 728, 784, 840, 900 (25 ... 51 spilled) measured 2 ... 14 % SLOWER and keep one
// wave.  630 and 660: only the inverse kernel was over, by 2 registers: 57 -> 41 us and 68 -> 47 us, step +14 % / +13 %.  (PSFMC_GEN_ROW_R_2WAVES = 32 bounds every general shape to two waves, for the A/B.)
#ifndef PSFMC_GEN_ROW_R_2WAVES
#define PSFMC_GEN_ROW_R_2WAVES 16
#endif
constexpr bool row_two_waves_side(int n) {
    return n == 630 || n == 650 || n == 660 || n == 676 || n == 700 || n == 720 || n == 780;
}
// Sides whose row kernels sit 2 ... 8 registers above an occupancy step (130 vector registers: three waves per
// SIMD instead of four; 172 ... 176: two instead of three), bounded to the next step where that measured faster
// (same box, kernel time): the forward kernel of 84, 98, 132, 160, 176 (-3 ... -6 %), the inverse kernel of 220, 260,
// 280, 300 (-9 ... -13 %: k_rows_inv<300> 41.9 -> 38.2 us, step +2.9 %).  The other way round -- the rasteriser's
// kernel at 220 ... 300, the inverse at the small sides -- the spills cost more than the extra wave hides; so they
// do for the inverse kernels of 88, 96, 160, 176, 192 (140 registers -> 128: -0.1 ... -2.4 %) and 294, 320, 336
// (174 ... 182 -> 168: -1 ... -15 %).
constexpr int row_fwd_more_waves(int n) { return (n == 84 || n == 98 || n == 132) ? 4 : ((n == 160 || n == 176) ? 3 : 0); }
constexpr int row_inv_more_waves(int n) {
    if (n == 150 || n == 180) return 4;                 // (132 / 131 registers; +1 %)
    return (n == 220 || n == 260 || n == 280 || n == 300) ? 3 : 0;
}
#ifndef PSFMC_FWD512_WAVES
#define PSFMC_FWD512_WAVES 0          /* k_rows_fwd<512> (184 registers) bounded to 3 waves per SIMD: 12 spilled, 46.3 -> 52.2 us, step -2.5 % */
#endif
// the forward kernel's WRAP variant (embedded images) carries a few registers more: at 768 -- a frequent embedding
// target -- that took it to 256 + 32 accumulation registers and one wave per SIMD
constexpr bool row_fwd_wrap_two_waves(int n) { return n == 768; }
template <int N, bool INVERSE, bool WRAP = false> constexpr int fused_row_min_waves() {
    if (WRAP && !INVERSE && row_fwd_wrap_two_waves(N)) return 2;
    if (PSFMC_FWD512_WAVES && N == 512 && !INVERSE) return PSFMC_FWD512_WAVES;
    // (the rasteriser's grouped pixel stages let the allocator of the 512 / 1024 forward kernels drift past 256
    // registers where it used to stop at 185 ... 220 by itself)
    if (FftShape<N>::kPlain && !INVERSE && N >= 512) return 2;
    // ... and so did the general sides from 640 on (one accumulation register over 256 = one wave per SIMD); the
    // four whose forward kernel runs at one wave by measurement (fused_row_min_waves' history above) stay there
    if (!INVERSE && pow_tabs_side(N) && N >= 640 && N != 728 && N != 784 && N != 840 && N != 900) return 2;
    if (FftShape<N>::kPlain) return fused_min_waves<N>();
    if ((INVERSE ? row_inv_more_waves(N) : row_fwd_more_waves(N)) > 0) return INVERSE ? row_inv_more_waves(N) : row_fwd_more_waves(N);
    return (FftShape<N>::R > PSFMC_GEN_ROW_R_2WAVES && !row_two_waves_side(N)) ? 1 : 2;
}
// the column kernel's own bound (experiments: more waves per SIMD instead of the register
// double-buffering)
template <int N> constexpr int fused_col_min_waves() {
    return (FftShape<N>::kPlain && FftShape<N>::R <= 16) ? PSFMC_PLAIN_COL_WAVES : fused_min_waves<N>();
}

// Address = wave-uniform base (scalar registers) + 32-bit byte offset per lane: the form
// global_load/store take directly (saddr + voffset).  64-bit per-lane pointer arithmetic was
// ~100 VALU instructions per wave across the three kernels.  Offsets stay far below 4 GB:
// they index one walker's T (<= 17 MB at 1024^2) or the kernel spectra.
template <typename Tp>
__device__ __forceinline__ Tp* at_bytes(Tp* base, unsigned byte_off) {
    return reinterpret_cast<Tp*>(reinterpret_cast<char*>(base) + byte_off);
}
template <typename Tp>
__device__ __forceinline__ const Tp* at_bytes(const Tp* base, unsigned byte_off) {
    return reinterpret_cast<const Tp*>(reinterpret_cast<const char*>(base) + byte_off);
}
constexpr unsigned kCd = sizeof(double) * 2;          // bytes of a T element

// element offset of (y, c) inside one (walker, kx) run of 2*ny complex.
// Sides are >= 64, so RG <= 8 <= T(ny): for y = T a + t the offset is affine in a,
// t_elem(T a + t) = t_elem(t) + 2 T a  (used by the column kernel).
__device__ __forceinline__ int t_elem(int y, int c, int rg_log2) {
    const int rg = 1 << rg_log2;
    return (((y >> rg_log2) * 2 + c) << rg_log2) + (y & (rg - 1));
}
// column length of the layout: ny rounded up to a whole number of row groups.  (PSFMC_T_PAD=1 adds
// a spare group where the distance between kx columns would be a multiple of 4 KB -- tried against
// the 1.42x fetch of k_rows_inv<1024>, where a row wave touches 32 kx columns 32 KB apart per
// instruction: no change at 256^2, 512^2 or 1024^2, so it is not a set-conflict effect; off.)
#ifndef PSFMC_T_PAD
#define PSFMC_T_PAD 0
#endif
__host__ __device__ __forceinline__ int t_col_len(int ny, int rg_log2) {
    const int len = ((ny + (1 << rg_log2) - 1) >> rg_log2) << rg_log2;
    return (PSFMC_T_PAD && (len & 127) == 0) ? len + (1 << rg_log2) : len;
}

// packed, pre-permuted field arrays for rows_inv: pix[(yg*P + e)*64 + lane] =
// {sci or NaN at excluded pixels, obs_var} of pixel (y = yg*RG + lane/T, x = lane%T + T*e)
struct FieldPx {
    double sci, var;
};

template <int NX> constexpr size_t fused_field_len(int ny) {
    return (size_t)((ny + FftShape<NX>::TPW - 1) / FftShape<NX>::TPW) * FftShape<NX>::R * 64;
}

// out[(yg*R + e)*64 + lane] = pixel (y = yg*RG + lane/T, x = fft_k_of(lane%T, e)); slots that
// hold no pixel (idle tail lanes, registers past P of an inexact shape, rows past ny) are
// marked excluded (NaN sci) with unit variance, so the chi^2 loop needs no other guard
template <int NX>
__global__ void k_pack_field(const double* __restrict__ sci, const double* __restrict__ obs_var,
                             const uint8_t* __restrict__ bad, FieldPx* __restrict__ out, int ny) {
    using S = FftShape<NX>;
    constexpr int T = S::T, R = S::R, RG = S::TPW;
    const int n = (int)fused_field_len<NX>(ny);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int lane = i & 63, e = (i >> 6) % R, yg = i / (64 * R);
        const int f = lane / T, t = lane % T;
        const int y = yg * RG + f, x = fft_k_of<NX>(t, e);
        const bool holds = f < RG && y < ny && fft_slot_valid<NX>(t, e);
        const size_t src = holds ? (size_t)y * NX + x : 0;
        out[i] = holds ? FieldPx{bad[src] ? __builtin_nan("") : sci[src], obs_var[src]}
                       : FieldPx{__builtin_nan(""), 1.0};
    }
}

// ---------------------------------------------------------------------------
// rows_fwd.  grid (ny / RG, n_walkers); one wave per workgroup.
//   FROM_IMAGE = false: rasterise from prep (the hot path)
//   FROM_IMAGE = true : z = img0 + i img_scale[w] img1 from memory (PSF spectra at setup)
// raw_out (optional): [n][ny][nx] copy of the raw model (psfmc_eval_images)
// ---------------------------------------------------------------------------
// WRAP: the image is embedded in a larger transform size (psfmc_device.h WrapDesc); ny, NX are the
// transform's sides
template <int NX, bool FROM_IMAGE, typename TS = cd, bool FAST = FftShape<NX>::kPlain, bool WRAP = false>
__global__ void __launch_bounds__((row_threads<NX, FAST>()), (fused_row_min_waves<NX, false, WRAP>()))
k_rows_fwd(const double* __restrict__ prep, const uint8_t* __restrict__ skip,
           const cd* __restrict__ twx, TS* __restrict__ Tbuf, int n_ps, int n_sersic, int ny,
           int ps_only, const double* __restrict__ img, const double* __restrict__ img_scale,
           double* __restrict__ raw_out, WrapDesc wr, int pow_mode) {
    using S = FftShape<NX>;
    constexpr int P = S::P, T = S::T, R = S::R, RG = row_group<NX>();
    constexpr int NXH = NX / 2 + 1;
    constexpr int RGL2 = layout_rg_log2<NX, FAST>(), RGL = 1 << RGL2;       // rows per layout group
    extern __shared__ __align__(16) double smem[];
    const int w = blockIdx.y;
#if PSFMC_FWD_PRIO_STAGGER
    // experiment: the workgroups that share a CU get different instruction-issue priorities, so that the waves of a
    // SIMD finish -- and store -- at different times instead of all together
    switch (((blockIdx.y * gridDim.x + blockIdx.x) / PSFMC_FWD_PRIO_STAGGER) & 3) {      // (the operand is an immediate)
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        case 3: __builtin_amdgcn_s_setprio(3); break;
        default: break;
    }
#endif
    // the skip flag is a (wave-uniform) byte behind a vector load: tested only after the loads that
    // do not depend on it have been issued, so that its latency is not a serial step of every wave
    const bool skipped = skip && skip[w];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int f = lane / T, t = lane % T;
    const int yg = blockIdx.x * row_waves<NX, FAST>() + wave;
    if constexpr (!FAST)
        if (yg * RG >= ny) return;                                    // wave-uniform: past the last row group
    const bool lane_on = S::kFull || f < RG;                          // not one of the idle tail lanes
    const int fe = S::kFull ? f : (f < RG ? f : RG - 1);              // LDS region (idle lanes alias the last)
    const int iy = yg * RG + f;
    const bool row_on = FAST || (lane_on && iy < ny);            // this lane's row exists
    const int nyp = t_col_len(ny, RGL2);
    const size_t Spx = (size_t)ny * NX;

    cd v[R];
#pragma unroll
    for (int k = P; k < R; ++k) v[k] = cd{0.0, 0.0};
    if constexpr (FROM_IMAGE) {
        if (skipped) return;
        const double* a = img + (size_t)(2 * w) * Spx + (size_t)(row_on ? iy : 0) * NX;
        const double* b = a + Spx;
        const double sc = img_scale[w];
#pragma unroll
        for (int k = 0; k < P; ++k) v[k] = row_on ? cd{a[T * k + t], b[T * k + t] * sc} : cd{0.0, 0.0};
    } else {
        const double* wprep = prep + (size_t)w * prep_len(n_ps, n_sersic);   // wave-uniform
        const double mu = wprep[kPrepMu];
        if (skipped) return;                          // a skipped walker's record may hold anything
        // the rasteriser's tables borrow the start of the wave's transform region (exchange area and
        // twiddle table), which is idle until the transform begins
        static_assert(fused_row_wave_lds_doubles<NX>() >= (pow_tabs_side(NX) ? (size_t)kRasterLdsDoubles : (size_t)kLogTabBytes / 8),
                      "wave region too small");
        double* log_tab = smem + (size_t)wave * fused_row_wave_lds_doubles<NX>();
        if constexpr (!pow_tabs_side(NX)) {
            if (!ps_only) {
                load_log_table(log_tab, lane);
                wave_lds_sync();
            }
        } else {
            if (!ps_only && n_sersic > 0) load_a_table(log_tab, lane);   // (raster_row fences before its reads)
        }
        double r[P];
        if constexpr (!pow_tabs_side(NX)) {
            raster_row_logexp<P, T, 0, WRAP>(wprep, n_ps, n_sersic, t, iy, ps_only != 0, log_tab, r, wr);
        } else {
            // (the WRAP variants keep the serial pixel order: with their P wrapped coordinates on top, the grouped
            // form spilled or lost a wave -- embedded 586^2 -17 %, 698^2 -21 % whole step)
            raster_row<P, T, 0, WRAP, WRAP ? 1 : raster_group<NX>()>(wprep, n_ps, n_sersic, t, iy, ps_only != 0, log_tab, r,
                                                                    wr, pow_mode);
        }
        wave_lds_sync();
#pragma unroll
        for (int k = 0; k < P; ++k) v[k] = cd{r[k], mu * r[k] * r[k]};
        if (raw_out && row_on) {
            double* o = raw_out + (size_t)w * Spx + (size_t)iy * NX;
#pragma unroll
            for (int k = 0; k < P; ++k) o[T * k + t] = v[k].x;
        }
    }
    double* wave_lds = smem + (size_t)wave * fused_row_wave_lds_doubles<NX>();
    cd* twl = reinterpret_cast<cd*>(wave_lds + (size_t)RG * fft_lds_elems<NX>());
    cd tw[fft_tw_regs<NX>()];
    load_twiddles<NX>(tw, twx, t, twl, lane);
    double* xbuf = wave_lds + (size_t)fe * fft_lds_elems<NX>();
#if !(PSFMC_DEBUG_FWD & 2)
    fft_wave<NX, -1>(v, tw, twx, t, xbuf, twl, lane_on);
#endif

    cd* ubuf = reinterpret_cast<cd*>(xbuf);
#if PSFMC_DEBUG_FWD & 1
    {   // timing experiment: no untangle / store phase (one value kept alive)
        double acc = 0.0;
#pragma unroll
        for (int e = 0; e < R; ++e) acc += v[e].x + v[e].y;
        if (acc == 1.2345e300) Tbuf[0] = TS{};
        return;
    }
#endif
    TS* wbase = Tbuf + (size_t)w * 2 * NXH * nyp;                    // wave-uniform
    constexpr unsigned kEl = sizeof(TS);                             // bytes of a T element
    const unsigned kstride = 2u * (unsigned)nyp * kEl;               // bytes between kx columns
#if PSFMC_FWD_STORE_BARRIER
    // experiment: the waves of a workgroup write the halves of the same 128-byte lines (two rows per wave at nx = 1024);
    // started together again, the halves should meet in the L2 instead of going out as two partial writes
    if constexpr (FAST && S::TPW == 2) __syncthreads();
#endif
    if constexpr (FAST) {
        // Untangle.  Lane t holds Z[k], k = t + T e.  Z[NX - k] is held by lane
        // (T - t) % T at e' = P-1-e (t != 0) or P-e (t == 0), i.e. in the upper half of
        // its registers: pass the upper halves through LDS (wave-local; the N/2 complex
        // fit the transform's exchange region of T (P+1) doubles).
#pragma unroll
        for (int e = P / 2; e < P; ++e) ubuf[(e - P / 2) * T + t] = v[e];
        wave_lds_sync();
        const int tm = (T - t) % T;
        const int shift = t ? P - 1 : P;
        const unsigned off0 = (unsigned)t_elem(iy, 0, RGL2) * kEl + (unsigned)t * kstride;
#pragma unroll
        for (int e = 0; e < P / 2; ++e) {
            const cd zk = v[e];
            cd zm = (e == 0 && t == 0) ? zk                       // k = 0 is its own mirror
                                       : ubuf[(shift - e - P / 2) * T + tm];
            TS* o = at_bytes(wbase, off0 + (unsigned)(T * e) * kstride);
            // TWICE the spectra of raw and of mu raw^2: the 1/2 of the untangling is a power of
            // two and rides on the kernel spectra (k_scale_kernel_spectrum), bit for bit the same
            o[0] = cd{zk.x + zm.x, zk.y - zm.y};
            o[RGL] = cd{zk.y + zm.y, zm.x - zk.x};
        }
        if (t == 0) {                                           // Nyquist column, its own mirror
            const cd z = v[P / 2];
            TS* o = at_bytes(wbase, off0 + (unsigned)(NX / 2) * kstride);
            o[0] = cd{z.x + z.x, 0.0};
            o[RGL] = cd{z.y + z.y, 0.0};
        }
    } else {
        // General shapes: register e of lane t holds Z[k], k = fft_k_of(t, e).  Every Z[k] with
        // k > NX/2 goes to LDS slot NX - k; the holder of k <= NX/2 reads its mirror from slot k
        // (k = 0 and k = NX/2 are their own mirrors, for which the same formulas give the
        // doubled real spectrum values).
#pragma unroll
        for (int e = 0; e < R; ++e) {
            const int k = fft_k_of<NX>(t, e);
            if (lane_on && fft_slot_valid<NX>(t, e) && 2 * k > NX) ubuf[NX - k] = v[e];
        }
        wave_lds_sync();
        const unsigned off_row = (unsigned)t_elem(row_on ? iy : 0, 0, RGL2) * kEl;
#pragma unroll
        for (int e = 0; e < R; ++e) {
            const int k = fft_k_of<NX>(t, e);
            if (row_on && fft_slot_valid<NX>(t, e) && 2 * k <= NX) {
                const cd zk = v[e];
                // (a select between the register and the LDS slot itself became a pointer select
                // through scratch and a flat load: read a valid slot always, select the value)
                const bool self = k == 0 || 2 * k == NX;
                cd zm = ubuf[self ? 1 : k];
                if (self) zm = zk;
                TS* o = at_bytes(wbase, off_row + (unsigned)k * kstride);
                o[0] = cd{zk.x + zm.x, zk.y - zm.y};
                o[RGL] = cd{zk.y + zm.y, zm.x - zk.x};
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Work order of the column kernels.  Column items are numbered kx-major, then walker, then
// component: item = (kx * n_w + w) * 2 + c, so that all walkers of a pass meet one kernel-
// spectrum column Kt[psf][kx][c][.] while it is still in an L2 (walker-major order re-read every
// Kt column from the Infinity Cache once per walker: +16 % fetch at 256^2 where Kt is 1 MB, +100 %
// at 1024^2 where it is 17 MB -- profiles/r2_pmc_*).  The two components of a (kx, w) pair stay
// adjacent: they are the interleaved halves of the same cache lines.
// The list of workgroup-sized groups is cut into one contiguous share per XCD (workgroups are
// dealt round-robin over the 8 XCDs, so b % 8 names the XCD a workgroup shares its L2 with;
// MI355X_MICROARCH.md, Workgroup dispatch -- a speed assumption only: any placement covers
// every group exactly once).
// ---------------------------------------------------------------------------
struct GroupRange { int first, end, step; };
// (the divisions of xcd_group_range run on the vector unit and leave wave-uniform values in vector registers, and with
// them every address derived from a group number: this moves them back to scalar registers)
__device__ __forceinline__ GroupRange scalar_range(GroupRange g) {
    return GroupRange{__builtin_amdgcn_readfirstlane(g.first), __builtin_amdgcn_readfirstlane(g.end),
                      __builtin_amdgcn_readfirstlane(g.step)};
}
__device__ __forceinline__ GroupRange xcd_group_range(int n_groups) {
    const int nb = (int)gridDim.x, b = (int)blockIdx.x;
    const int nx = nb >= 8 ? 8 : 1;
    const int x = b % nx, j = b / nx;
    const int blocks_here = (nb - x + nx - 1) / nx;
    const int lo = (int)((long long)x * n_groups / nx), hi = (int)((long long)(x + 1) * n_groups / nx);
    return GroupRange{lo + j, hi, blocks_here};
}

// ---------------------------------------------------------------------------
// cols.  persistent grid of 256-thread workgroups; slot s of a workgroup works
// on column item (kx*n_w + walker)*2 + c.
//   CONVOLVE = true : FFT_y, * Kt[psf][kx][c][.], IFFT_y (the hot path)
//   CONVOLVE = false: FFT_y only (PSF spectra at setup)
// ---------------------------------------------------------------------------
template <int NY, bool CONVOLVE, typename TS = cd>
__global__ void __launch_bounds__(kColThreads, fused_col_min_waves<NY>())
k_cols(TS* __restrict__ Tbuf, const cd* __restrict__ Kt, const double* __restrict__ prep,
       const uint8_t* __restrict__ skip, const cd* __restrict__ twy, int plen, int nxh, int n_w,
       int rg_log2) {
    using S = FftShape<NY>;
    constexpr int P = S::P, T = S::T, R = S::R, TPW = S::TPW;
    constexpr int FPB = col_ffts_per_block<NY>();
    extern __shared__ __align__(16) double smem[];
#if PSFMC_COLS_PRIO
    // the memory-bound kernel's waves go first where they share a SIMD with the VALU-bound row
    // waves of the other in-flight pass: their loads and stores get out sooner
    __builtin_amdgcn_s_setprio(PSFMC_COLS_PRIO);
#endif
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fl = lane / T, t = lane % T;
    const bool slot_on = S::kFull || fl < TPW;                       // not an idle tail lane
    const int fe = S::kFull ? fl : (fl < TPW ? fl : TPW - 1);
    const int s = wave * TPW + fe;                                   // slot within the workgroup
    double* wave_lds = smem + (size_t)wave * fused_col_wave_lds_doubles<NY>();
    double* xbuf = wave_lds + (size_t)fe * fft_lds_elems<NY>();
    cd* twl = reinterpret_cast<cd*>(wave_lds + (size_t)TPW * fft_lds_elems<NY>());
    constexpr int TM = PSFMC_TW_MODE_COLS;
    cd tw[TwRegs<NY, TM>::value];
    load_twiddles<NY, TM>(tw, twy, t, twl, lane);
    const int nyp = t_col_len(NY, rg_log2);                          // column length of the layout
    const int rg_mask = (1 << rg_log2) - 1;
    const int n_cols = n_w * 2 * nxh;
    const int n_groups = (n_cols + FPB - 1) / FPB;
    const GroupRange gr = xcd_group_range(n_groups);
    // slot s of group grp: its column, whether it is live, and the lane's first element
    // Slots past the last column (final group only) and skipped walkers still LOAD -- from
    // the last valid column / the walker's stale data -- and transform; only their stores are
    // masked.  Zero-filling their registers instead cost 32 moves per group in every lane.
    // Returns the column's base: element (y, c) sits at base[2 y - (y & rg_mask)] (t_elem with
    // the component's offset folded into the base).
    auto locate = [&](int grp, int& w, int& kx, int& c, bool& active) -> TS* {
        int col = grp * FPB + s;
        active = slot_on && col < n_cols;
        col = col < n_cols ? col : n_cols - 1;
        c = col & 1;                              // component
        const int pr = col >> 1;                  // kx * n_w + walker
        kx = pr / n_w;
        w = pr - kx * n_w;
        if (active && skip && skip[w]) active = false;
        return Tbuf + ((size_t)w * nxh + kx) * 2 * nyp + (c << rg_log2);
    };
    // element offset of row y inside a column (see locate); for the power-of-two shapes the
    // row groups divide T, so the offset is affine in the register index
    auto row_off = [&](int y) -> int { return 2 * y - (y & rg_mask); };
    // Software pipeline (small shapes): the next group's column is loaded into a second register
    // set while this one is transformed.  Loads return in order, so they are issued AFTER the
    // kernel-spectrum loads were consumed and fly during the inverse transform and the
    // stores (issued before the forward transform they made the multiply wait for them:
    // 56 us instead of 46; here 43 us).  The twiddles live in LDS to make room.
    // (general shapes: only those compiled for one wave per SIMD have the registers for it)
    constexpr bool PF = PSFMC_COLS_PREFETCH && (S::kPlain ? R <= 16 : (R > PSFMC_GEN_R_2WAVES && R <= PSFMC_GEN_PF_MAX_R));
    struct Slot {
        TS* base;
        int w, kx, c;
        bool active;
    };
    // General shapes: row y = t + c with c a compile-time constant per register (T a on entry, T h + P d on
    // exit), and y & rg_mask depends on c only through c & 7 (row groups hold at most 8 rows): eight per-lane
    // byte offsets, formed once, and an immediate per register replace a 64-bit address computation per
    // element (k_cols<200>: 300 of 3900 vector instructions per pair of columns).
    unsigned boff[8];
    if constexpr (!S::kPlain) {
#pragma unroll
        for (int j = 0; j < 8; ++j) boff[j] = (unsigned)(2 * t - ((t + j) & rg_mask)) * (unsigned)sizeof(TS);
    }
    auto gen_off = [&](int c) -> unsigned { return boff[c & 7] + (unsigned)(2 * c) * (unsigned)sizeof(TS); };
    auto load_group = [&](int grp, cd (&dst)[R], Slot& sl) {
        sl.base = locate(grp, sl.w, sl.kx, sl.c, sl.active);
        if constexpr (S::kPlain) {
            const TS* b0 = sl.base + row_off(t);
#pragma unroll
            for (int a = 0; a < P; ++a) dst[a] = load_stream(b0 + 2 * T * a);
        } else {
#pragma unroll
            for (int a = 0; a < P; ++a) dst[a] = load_stream(at_bytes(sl.base, gen_off(T * a)));
#pragma unroll
            for (int a = P; a < R; ++a) dst[a] = cd{0.0, 0.0};
        }
    };
    // one column per slot: forward, * kernel spectrum, `between()`, inverse, store in place
    auto transform = [&](cd (&v)[R], const Slot& sl, auto&& between) {
        fft_wave<NY, -1, TM>(v, tw, twy, t, xbuf, twl, slot_on);
        if constexpr (CONVOLVE) {
            // a skipped walker's record may hold anything: its (masked) lanes use PSF 0
            const int psf = sl.active ? (int)prep[(size_t)sl.w * plen + kPrepPsfIdx] : 0;
            const cd* k = Kt + (((size_t)psf * nxh + sl.kx) * 2 + sl.c) * NY;
            if constexpr (S::kPlain) {
#pragma unroll
                for (int e = 0; e < R; ++e) v[e] = cmul(v[e], k[t + T * e]);
            } else {
                // register e holds X[fft_k_of(t, e)]
#pragma unroll
                for (int e = 0; e < R; ++e) {
                    const int ky = fft_slot_valid<NY>(t, e) ? fft_k_of<NY>(t, e) : 0;
                    v[e] = cmul(v[e], k[ky]);
                }
            }
        }
        between();
        if constexpr (CONVOLVE) {
            // an inexact shape leaves X[(t + T h) + P d] in register h + H d, and a transform
            // wants x[T a + t] in register a: regroup through LDS
            if constexpr (!S::kExact) fft_regroup<NY>(v, t, xbuf, slot_on);
            fft_wave<NY, +1, TM>(v, tw, twy, t, xbuf, twl, slot_on);
        }
        if (sl.active) {
            if constexpr (S::kPlain) {
                TS* b0 = sl.base + row_off(t);
#pragma unroll
                for (int e = 0; e < R; ++e) b0[2 * T * e] = v[e];
            } else {
#pragma unroll
                for (int e = 0; e < R; ++e)
                    if (fft_slot_valid<NY>(t, e)) *at_bytes(sl.base, gen_off(fft_k_of<NY>(0, e))) = v[e];
            }
        }
    };
    if constexpr (PF && sizeof(TS) != sizeof(cd)) {
        // single-precision storage: the next group waits in its raw complex64 form (half the
        // registers of a second fp64 set, which spilled: 108 bytes of scratch per lane at 256),
        // and is widened into the one fp64 working set when its turn comes
        static_assert(S::kPlain, "single-precision storage is built for the power-of-two shapes");
        TS raw[R];
        cd v[R];
        Slot cur{Tbuf, 0, 0, 0, false}, nxt{Tbuf, 0, 0, 0, false};
        auto load_raw = [&](int grp, Slot& sl) {
            sl.base = locate(grp, sl.w, sl.kx, sl.c, sl.active);
            const TS* b0 = sl.base + row_off(t);
#pragma unroll
            for (int a = 0; a < P; ++a) raw[a] = b0[2 * T * a];
        };
        if (gr.first < gr.end) load_raw(gr.first, nxt);
        for (int grp = gr.first; grp < gr.end; grp += gr.step) {
            cur = nxt;
#pragma unroll
            for (int a = 0; a < R; ++a) v[a] = cd{(double)raw[a].x, (double)raw[a].y};
            transform(v, cur, [&] {
                __builtin_amdgcn_sched_barrier(0);
                if (grp + gr.step < gr.end) load_raw(grp + gr.step, nxt);
                __builtin_amdgcn_sched_barrier(0);
            });
        }
    } else if constexpr (PF && !S::kPlain && PSFMC_GEN_STAGED) {
        // general shapes (compiled for one wave per SIMD): ONE working set; the next group waits in a staging
        // set that is only ever a load's destination and a copy's source, which the register allocator can
        // keep in accumulation registers at one move per value.  (Two working sets taking turns, as below,
        // had it shuffle 390 values per column between the two register files at 200: a fifth of the
        // kernel's vector instructions.)
        cd v[R], nxt[R];
        Slot cur{Tbuf, 0, 0, 0, false}, nx{Tbuf, 0, 0, 0, false};
        if (gr.first < gr.end) load_group(gr.first, nxt, nx);
        for (int grp = gr.first; grp < gr.end; grp += gr.step) {
            cur = nx;
#pragma unroll
            for (int a = 0; a < R; ++a) v[a] = nxt[a];
            transform(v, cur, [&] {
                __builtin_amdgcn_sched_barrier(0);
                if (grp + gr.step < gr.end) load_group(grp + gr.step, nxt, nx);
                __builtin_amdgcn_sched_barrier(0);
            });
        }
    } else if constexpr (PF) {
        // two register sets take turns (no copies): while one is transformed the other
        // receives the next group
        cd A[R], B[R];
        Slot sa{Tbuf, 0, 0, 0, false}, sb{Tbuf, 0, 0, 0, false};
        const int step = gr.step;
        if (gr.first < gr.end) load_group(gr.first, A, sa);
        for (int grp = gr.first; grp < gr.end; grp += 2 * step) {
            const int g1 = grp + step, g2 = g1 + step;
            transform(A, sa, [&] {
                __builtin_amdgcn_sched_barrier(0);
                if (g1 < gr.end) load_group(g1, B, sb);
                __builtin_amdgcn_sched_barrier(0);
            });
            if (g1 < gr.end)
                transform(B, sb, [&] {
                    __builtin_amdgcn_sched_barrier(0);
                    if (g2 < gr.end) load_group(g2, A, sa);
                    __builtin_amdgcn_sched_barrier(0);
                });
        }
    } else {
        for (int grp = gr.first; grp < gr.end; grp += gr.step) {
            cd v[R];
            Slot sl{Tbuf, 0, 0, 0, false};
            load_group(grp, v, sl);
            transform(v, sl, [] {});
        }
    }
}

// ---------------------------------------------------------------------------
// cols3: the column kernel for ny = 512, 1024 on the wave-wide three-stage engine
// (psfmc_fft.h fft_wave3): one wave per column, 4 waves per workgroup, persistent.
// ---------------------------------------------------------------------------
// the waves' exchange regions, then (R1 = 16) the shared stage-1 twiddle table [16][64]
#ifndef PSFMC_COLS3_W2_LDS
#define PSFMC_COLS3_W2_LDS 0        /* R1 = 16: the stage-2 twiddles from a 64-entry LDS table (32 registers back) */
#endif
template <int NY> constexpr bool cols3_w2_lds() { return PSFMC_COLS3_W2_LDS && Fft3Shape<NY>::R1 > 8; }
template <int NY> constexpr size_t fused_col3_lds_bytes() {
    return (size_t)(kColThreads / 64) * fft3_lds_doubles<NY>() * sizeof(double) +
           (Fft3Shape<NY>::R1 > 8 ? (size_t)Fft3Shape<NY>::R1 * 64 * sizeof(cd) : 0) +
           (cols3_w2_lds<NY>() ? 64 * sizeof(cd) : 0);
}

#ifndef PSFMC_COLS3_BAR
#define PSFMC_COLS3_BAR 7           /* which of the three scheduling barriers of k_cols3 are in */
#endif
#ifndef PSFMC_COLS3_WAVES
#define PSFMC_COLS3_WAVES 2     /* measured at 1024 (before the shared LDS twiddle table: 150 us now): 1 or 2 -> 178 us, 3 -> 273 us, 4 -> 374 us (spills) */
#endif
template <int NY, bool CONVOLVE, typename TS = cd>
__global__ void __launch_bounds__(kColThreads, Fft3Shape<NY>::R1 > 8 ? PSFMC_COLS3_WAVES : 2)
k_cols3(TS* __restrict__ Tbuf, const cd* __restrict__ Kt, const double* __restrict__ prep,
        const uint8_t* __restrict__ skip, const cd* __restrict__ twy, int plen, int nxh, int n_w,
        int rg_log2) {
    constexpr int R1 = Fft3Shape<NY>::R1;
    constexpr int WPB = kColThreads / 64;
    extern __shared__ __align__(16) double smem[];
#if PSFMC_COLS_PRIO
    __builtin_amdgcn_s_setprio(PSFMC_COLS_PRIO);
#endif
    const int t = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double* lds = smem + (size_t)wave * fft3_lds_doubles<NY>();
    cd w1[fft3_w1_regs<NY>()], w2[8];
    const cd* w1s = nullptr;
    const cd* w2s = nullptr;
    if constexpr (cols3_w2_lds<NY>()) {
#pragma unroll
        for (int k = 0; k < 8; ++k) w2[k] = cd{1.0, 0.0};
        w1[0] = cd{1.0, 0.0};
    } else {
        load_twiddles3<NY>(w1, w2, twy, t);
    }
    if constexpr (R1 > 8) {
        cd* tab = reinterpret_cast<cd*>(smem + (size_t)WPB * fft3_lds_doubles<NY>());
        for (int i = threadIdx.x; i < R1 * 64; i += kColThreads) tab[i] = twy[(i & 63) * (i >> 6)];
        if constexpr (cols3_w2_lds<NY>()) {
            cd* tab2 = tab + R1 * 64;                      // [k2][n3] = W_N^(R1 n3 k2)
            if (threadIdx.x < 64) tab2[threadIdx.x] = twy[R1 * (threadIdx.x & 7) * (threadIdx.x >> 3)];
            w2s = tab2;
        }
        __syncthreads();                                   // once, before any wave can leave
        w1s = tab;
    }
    const int rg = 1 << rg_log2;
    const int e0 = t_elem(t, 0, rg_log2);              // offset of y = t; y = 64 a + t adds 128 a
    const int n_cols = n_w * 2 * nxh;
    const int nyp = t_col_len(NY, rg_log2);
    const GroupRange gr = xcd_group_range((n_cols + WPB - 1) / WPB);
    for (int grp = gr.first; grp < gr.end; grp += gr.step) {
        const int col = grp * WPB + wave;
        if (col >= n_cols) continue;                     // wave-uniform
        const int pr = col >> 1, c = col & 1;           // kx * n_w + walker, component
        const int kx = pr / n_w, w = pr - kx * n_w;
        const bool skipped = skip && skip[w];            // wave-uniform; tested once the column's loads are issued
        TS* base = Tbuf + ((size_t)w * nxh + kx) * 2 * nyp + c * rg + e0;
        cd v[R1];
#pragma unroll
        for (int a = 0; a < R1; ++a) v[a] = load_stream(base + 128 * a);
        if (skipped) continue;
        fft_wave3<NY, -1>(v, w1, w2, twy, t, lds, w1s, w2s);
        if constexpr (CONVOLVE) {
            // keep the kernel-spectrum loads (and the next column's) out of the transform's
            // register budget: occupancy, not load hoisting, hides their latency here
#if PSFMC_COLS3_BAR & 1
            __builtin_amdgcn_sched_barrier(0);
#endif
            const int psf = (int)prep[(size_t)w * plen + kPrepPsfIdx];
            const cd* k = Kt + (((size_t)psf * nxh + kx) * 2 + c) * NY;
#pragma unroll
            for (int e = 0; e < R1; ++e) v[e] = cmul(v[e], k[t + 64 * e]);
#if PSFMC_COLS3_BAR & 2
            __builtin_amdgcn_sched_barrier(0);
#endif
            fft_wave3<NY, +1>(v, w1, w2, twy, t, lds, w1s, w2s);
#if PSFMC_COLS3_BAR & 4
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
#pragma unroll
        for (int e = 0; e < R1; ++e) base[128 * e] = v[e];
    }
}

// ---------------------------------------------------------------------------
// cols3g: the column kernel of the sides above 512 on the general wave-wide three-stage engine (psfmc_fft.h
// fft_wave3g; ny = R1 L with L = R2 R3 <= 64 lanes, R1 <= 16): one wave per column, 4 waves per workgroup,
// persistent.  (Round 2 ran these on the two-stage engine: 20 ... 32 complex registers per lane, one wave per
// SIMD, 1.7 ... 2.9 TB/s.)  The forward transform leaves X[(t + 64 q) + R1 R2 k3] in register (q, k3): the
// kernel spectrum is indexed by that k, and the inverse transform is the forward one's mirror image
// (fft_wave3g_inv), which takes exactly that layout and returns the column in the layout it was loaded in.
// ---------------------------------------------------------------------------
// Waves per workgroup and per SIMD above 1024 (R1 = 18, 20: 1152, 1280).  Compiled for ONE wave per SIMD the kernel
// holds 288 registers; bounded to two it spills 84 bytes per lane and is still the faster one where the LDS lets two
// waves share a SIMD: 1152 (78 KB per four-wave workgroup, two per CU) 92 -> 68 us per 4-walker pass, step +16 %; at
// 1280 four waves take 86 KB -- one workgroup per CU whatever the registers (80.5 us, no gain) -- so EIGHT waves share
// the two tables there (132 KB).  profiles/r4_cols3g_2waves_1152.txt
#ifndef PSFMC_COLS3G_R1_2WAVES
#define PSFMC_COLS3G_R1_2WAVES 20     /* k_cols3g is compiled for two waves per SIMD up to this many registers per lane */
#endif
#ifndef PSFMC_COLS3G_WAVES_1280
#define PSFMC_COLS3G_WAVES_1280 8
#endif
template <class S> constexpr int cols3g_waves() { return S::R1 == 20 && S::L == 64 ? PSFMC_COLS3G_WAVES_1280 : kColThreads / 64; }
template <class S> constexpr int cols3g_threads() { return 64 * cols3g_waves<S>(); }
template <class S> constexpr size_t fused_col3g_lds_bytes() {
    return ((size_t)cols3g_waves<S>() * fft3g_lds_doubles<S>() + (size_t)S::R1 * 64 * 2 + (size_t)S::R1 * S::L * 2) *
           sizeof(double);
}
// the row-group size the kernel's affine addressing allows for a side (see in_off)
#ifndef PSFMC_COLS3G_KT_CHUNKS
#define PSFMC_COLS3G_KT_CHUNKS 1      /* R1 > 16: the kernel-spectrum values of one output block in flight at a time (96 -> 84 bytes of scratch) */
#endif
#ifndef PSFMC_COLS3G_SCALAR_BASE
#define PSFMC_COLS3G_SCALAR_BASE 1
#endif
// The scalar base where 4 does not divide L (250, 294, 330, 350, 440, 500, 630: eight per-lane offsets instead of one),
// alone (profiles/r4_cols3g_gen_offsets_probe.txt): 500 67.0 -> 52.7 us, 294 52.4 -> 49.8; 250 / 350 / 630 even; 330 and
// 440 (L = 55: all eight offsets in use) 67.7 -> 82, 60.4 -> 70.6.  It also takes the 5 x 10 splits of 650 / 700 / 800
// from 108 ... 115 to 65 ... 69 us -- level with, not ahead of, their two-stage kernels (70 / 63 / 54).  Whole step
// (r4_cols3g_gen_offsets_step.txt): 500 +7.1 %, 294 inside the noise (-1.8 ... +1.1 %).  -1: 500.
#ifndef PSFMC_COLS3G_GEN_OFFSETS
#define PSFMC_COLS3G_GEN_OFFSETS -1
#endif
template <class S> constexpr bool cols3g_gen_offsets() {
    return PSFMC_COLS3G_GEN_OFFSETS < 0 ? S::kN == 500 : PSFMC_COLS3G_GEN_OFFSETS != 0;
}
// k_cols3f's load pipeline in this kernel (short columns).  Alone -- tools/cols3g_shapes.hip,
// profiles/r4_cols3g_prefetch_probe.txt: R1 = 5 ... 6 elements per lane (250, 288, 300, 336) 3 ... 10 % SLOWER, R1 = 8 (352,
// 384, 416, 480) 3 ... 5 % faster; in the step (r4_cols3g_prefetch_step.txt, the R1 = 8 sides): 416 +3.5 %, 384 +1.2 %,
// 352 / 440 / 480 +0.0 ... 0.2 %; with the re-surveyed shapes (r4_cols3g_prefetch_step2.txt) 384 still +2 %, 504 +2.5 %,
// 392 / 448 / 480 -1 ... -4 %; at 9 ... 13 elements per lane (r4_cols3g_prefetch_step3.txt: eleven sides 500 ... 832) 660
// +2.2 %, the others -2.3 ... +1.3 % = the noise.  On at 384, 416, 504 and 660 (PSFMC_COLS3G_PREFETCH: -1 those, 0 nowhere,
// 1 every R1 <= 8).
#ifndef PSFMC_COLS3G_PREFETCH
#define PSFMC_COLS3G_PREFETCH -1
#endif
template <class S> constexpr bool cols3g_prefetch() {
    return PSFMC_COLS3G_PREFETCH < 0 ? (S::kN == 384 || S::kN == 416 || S::kN == 504 || S::kN == 660) : (PSFMC_COLS3G_PREFETCH != 0 && S::R1 <= 8);
}
template <class S> constexpr bool cols3g_layout_ok(int rg_log2) { return S::L % 4 != 0 || S::L % (1 << rg_log2) == 0; }
template <int NY> constexpr bool cols3g_side() { return Fft3gShape<NY>::kBuilt && NY > PSFMC_COLS3G_MIN; }

template <int NY, bool CONVOLVE, class S = Fft3gShape<NY>>
__global__ void __launch_bounds__((cols3g_threads<S>()), (S::R1 > PSFMC_COLS3G_R1_2WAVES ? 1 : 2))
k_cols3g(cd* __restrict__ Tbuf, const cd* __restrict__ Kt, const double* __restrict__ prep,
         const uint8_t* __restrict__ skip, const cd* __restrict__ twy, int plen, int nxh, int n_w, int rg_log2) {
    constexpr int R1 = S::R1, R2 = S::R2, R3 = S::R3, L = S::L, NB3 = S::NB3, WPB = cols3g_waves<S>();
    extern __shared__ __align__(16) double smem[];
#if PSFMC_COLS_PRIO
    __builtin_amdgcn_s_setprio(PSFMC_COLS_PRIO);
#endif
    const int t = threadIdx.x & 63;
    const int wave = PSFMC_COLS3G_SCALAR_BASE ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : (int)(threadIdx.x >> 6);
    const bool lane_in = L == 64 || t < L;
    const int tl = lane_in ? t : 0;
    double* lds = smem + (size_t)wave * fft3g_lds_doubles<S>();
    cd* tab = reinterpret_cast<cd*>(smem + (size_t)WPB * fft3g_lds_doubles<S>());
    cd* tab_b = tab + R1 * 64;                         // the inverse transform's first twiddles [n3][c] = W_N^(n3 c)
    for (int i = threadIdx.x; i < R1 * 64; i += cols3g_threads<S>())
        tab[i] = (i & 63) < L ? twy[(i & 63) * (i >> 6)] : cd{0.0, 0.0};
    if constexpr (CONVOLVE)
        for (int i = threadIdx.x; i < NY; i += cols3g_threads<S>()) tab_b[i] = twy[(i / (R1 * R2)) * (i % (R1 * R2))];
    __syncthreads();                                   // once, before any wave can leave
    cd w2[R2];
#pragma unroll
    for (int k = 0; k < R2; ++k) w2[k] = twy[R1 * (tl % R3) * k];
    const int rg_mask = (1 << rg_log2) - 1;
    auto row_off = [&](int y) -> int { return 2 * y - (y & rg_mask); };      // element offset of row y in a column
    constexpr bool kAffine = L % 4 == 0;               // (row groups of 4 or, with 8 | L, 8 rows)
    const int off_t = row_off(tl);
    unsigned boff[8];
#pragma unroll
    for (int j8 = 0; j8 < 8; ++j8) boff[j8] = (unsigned)(2 * tl - ((tl + j8) & rg_mask)) * kCd;
    const int n_cols = n_w * 2 * nxh;
    const int nyp = t_col_len(NY, rg_log2);
    // PSFMC_COLS3G_SCALAR_BASE: a column's addresses as a scalar base + a 32-bit lane offset (see k_cols3f)
    const GroupRange gr0 = xcd_group_range((n_cols + WPB - 1) / WPB);
    const GroupRange gr = PSFMC_COLS3G_SCALAR_BASE ? scalar_range(gr0) : gr0;
    if constexpr (cols3g_prefetch<S>() && CONVOLVE) {
        // load pipeline (short columns: R1 <= 8 elements per lane): the wave's next column is loaded into a second
        // register set between the kernel-spectrum multiply and the inverse transform of the current one
        auto locate = [&](int grp, int& w, int& kx, int& c, bool& live) -> cd* {
            int col = grp * WPB + wave;
            live = col < n_cols;
            col = live ? col : n_cols - 1;
            c = col & 1;
            const int pr = col >> 1;
            kx = __builtin_amdgcn_readfirstlane(pr / n_w);
            w = pr - kx * n_w;
            if (live && skip && skip[w]) live = false;
            return Tbuf + ((size_t)w * nxh + kx) * 2 * nyp + (c << rg_log2);     // (wave-uniform, in scalar registers)
        };
        auto el = [&](cd* base, unsigned ob, int a) -> cd* {
            if constexpr (kAffine) return at_bytes(base, ob + (unsigned)(2 * L * a) * kCd);
            else return base + row_off(L * a + tl);
        };
        cd nxt[R1];
        int w_n = 0, kx_n = 0, c_n = 0;
        bool live_n = false;
        cd* base_n = Tbuf;
        if (gr.first < gr.end) {
            base_n = locate(gr.first, w_n, kx_n, c_n, live_n);
            const unsigned ob = (unsigned)off_t * kCd;
#pragma unroll
            for (int a = 0; a < R1; ++a) nxt[a] = load_stream(el(base_n, ob, a));
        }
        for (int grp = gr.first; grp < gr.end; grp += gr.step) {
            unsigned ob = (unsigned)off_t * kCd;
            if constexpr (kAffine) asm volatile("" : "+v"(ob));
            cd v[R1];
#pragma unroll
            for (int a = 0; a < R1; ++a) v[a] = nxt[a];
            cd* base = base_n;
            const int w = w_n, kx = kx_n, c = c_n;
            const bool live = live_n;
            cd o[NB3][R3];
            fft_wave3g<S, -1>(v, o, w2, t, lds, tab);
            __builtin_amdgcn_sched_barrier(0);
            const int psf = __builtin_amdgcn_readfirstlane(live ? (int)prep[(size_t)w * plen + kPrepPsfIdx] : 0);
            const cd* k = Kt + (((size_t)psf * nxh + kx) * 2 + c) * NY;
#pragma unroll
            for (int q = 0; q < NB3; ++q) {
                const bool ok = fft3g_valid<S>(t, q);
#pragma unroll
                for (int k3 = 0; k3 < R3; ++k3) o[q][k3] = cmul(o[q][k3], k[ok ? fft3g_index<S>(t, q, k3) : 0]);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (grp + gr.step < gr.end) {
                base_n = locate(grp + gr.step, w_n, kx_n, c_n, live_n);
#pragma unroll
                for (int a = 0; a < R1; ++a) nxt[a] = load_stream(el(base_n, ob, a));
            }
            __builtin_amdgcn_sched_barrier(0);
            fft_wave3g_inv<S>(o, v, t, lds, tab, tab_b);
            __builtin_amdgcn_sched_barrier(0);
            if (live && lane_in) {
                unsigned os = (unsigned)off_t * kCd;
                if constexpr (kAffine) asm volatile("" : "+v"(os));
#pragma unroll
                for (int a = 0; a < R1; ++a) *el(base, os, a) = v[a];
            }
        }
        return;
    }
    for (int grp = gr.first; grp < gr.end; grp += gr.step) {
        const int col = grp * WPB + wave;
        if (col >= n_cols) continue;                     // wave-uniform
        const int pr = col >> 1, c = col & 1;           // kx * n_w + walker, component
        const int kx = PSFMC_COLS3G_SCALAR_BASE ? __builtin_amdgcn_readfirstlane(pr / n_w) : pr / n_w, w = pr - kx * n_w;
        const bool skipped = skip && skip[w];            // wave-uniform; tested once the column's loads are issued
        cd* base = Tbuf + ((size_t)w * nxh + kx) * 2 * nyp + (c << rg_log2);
        // L a + t: with L a multiple of the row group (host-checked, cols3g_layout_ok) the offset is affine in a
        auto in_off = [&](int a) -> int { return kAffine ? off_t + 2 * L * a : row_off(L * a + tl); };
        unsigned ob = (unsigned)off_t * kCd;
        if (PSFMC_COLS3G_SCALAR_BASE && kAffine) asm volatile("" : "+v"(ob));
        cd v[R1];
        if constexpr (PSFMC_COLS3G_SCALAR_BASE && kAffine) {
#pragma unroll
            for (int a = 0; a < R1; ++a) v[a] = load_stream(at_bytes(base, ob + (unsigned)(2 * L * a) * kCd));
        } else if constexpr (PSFMC_COLS3G_SCALAR_BASE && cols3g_gen_offsets<S>()) {
            // 4 does not divide L: row L a + t = t + c with c a constant per register, and (row & rg_mask) depends on c
            // only through c mod 8 -- eight per-lane byte offsets formed once (k_cols' scheme), one of them + an
            // immediate per element on the scalar base
            unsigned bo[8];
#pragma unroll
            for (int j8 = 0; j8 < 8; ++j8) { bo[j8] = boff[j8]; asm volatile("" : "+v"(bo[j8])); }
#pragma unroll
            for (int a = 0; a < R1; ++a) v[a] = load_stream(at_bytes(base, bo[(L * a) & 7] + (unsigned)(2 * L * a) * kCd));
        } else {
#pragma unroll
            for (int a = 0; a < R1; ++a) v[a] = load_stream(base + in_off(a));
        }
        if (skipped) continue;
        cd o[NB3][R3];
        fft_wave3g<S, -1>(v, o, w2, t, lds, tab);
        if constexpr (CONVOLVE) {
            __builtin_amdgcn_sched_barrier(0);
            const int psf = (int)prep[(size_t)w * plen + kPrepPsfIdx];
            const cd* k = Kt + (((size_t)psf * nxh + kx) * 2 + c) * NY;
#pragma unroll
            for (int q = 0; q < NB3; ++q) {
                const bool ok = fft3g_valid<S>(t, q);
#pragma unroll
                for (int k3 = 0; k3 < R3; ++k3) o[q][k3] = cmul(o[q][k3], k[ok ? fft3g_index<S>(t, q, k3) : 0]);
                if constexpr (PSFMC_COLS3G_KT_CHUNKS && R1 > 16) __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_sched_barrier(0);
            fft_wave3g_inv<S>(o, v, t, lds, tab, tab_b);
            __builtin_amdgcn_sched_barrier(0);
            if (lane_in) {
                if constexpr (PSFMC_COLS3G_SCALAR_BASE && kAffine) {
                    unsigned os = (unsigned)off_t * kCd;
                    asm volatile("" : "+v"(os));
#pragma unroll
                    for (int a = 0; a < R1; ++a) *at_bytes(base, os + (unsigned)(2 * L * a) * kCd) = v[a];
                } else if constexpr (PSFMC_COLS3G_SCALAR_BASE && cols3g_gen_offsets<S>()) {
                    unsigned bo[8];
#pragma unroll
                    for (int j8 = 0; j8 < 8; ++j8) { bo[j8] = boff[j8]; asm volatile("" : "+v"(bo[j8])); }
#pragma unroll
                    for (int a = 0; a < R1; ++a) *at_bytes(base, bo[(L * a) & 7] + (unsigned)(2 * L * a) * kCd) = v[a];
                } else {
#pragma unroll
                    for (int a = 0; a < R1; ++a) base[in_off(a)] = v[a];
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < NB3; ++q)
                if (fft3g_valid<S>(t, q)) {
#pragma unroll
                    for (int k3 = 0; k3 < R3; ++k3) base[row_off(fft3g_index<S>(t, q, k3))] = o[q][k3];
                }
        }
    }
}

// ---------------------------------------------------------------------------
// cols3f (round 4): the column kernel of ny = 8 m x 64 (512, 1024, 1536, 2048) on the general three-stage engine
// run FORWARD both ways.  With R2 = R3 = 8 and R1 a multiple of 8 the forward transform's output register (q, k3)
// holds X[t + 64 (q + NB3 k3)] -- the engine's own INPUT layout with a = q + NB3 k3 -- so the inverse transform is
// the same engine with conjugated twiddles (no mirrored engine, no second twiddle table: R1 KB + N doubles per wave
// of LDS, which is what lets two workgroups share a CU at 1536), and the kernel is written for a third wave per
// SIMD at R1 = 16.
// ---------------------------------------------------------------------------
template <class S> constexpr bool cols3f_shape() { return S::kBuilt && S::R2 == 8 && S::R3 == 8 && S::R1 % 8 == 0; }
// ny = 2048: a wave's exchange region is 18 KB and the stage-1 table 32 KB -- four waves are ONE workgroup and one wave
// per SIMD on a CU.  Eight waves around HALF the table (fft_wave3g HALF1) are exactly the CU's 160 KB: two waves per
// SIMD at <= 256 registers, without the load pipeline (PSFMC_COLS3F_WAVES_2048 = 4: the old form, with it).
#ifndef PSFMC_COLS3F_WAVES_2048
#define PSFMC_COLS3F_WAVES_2048 8
#endif
template <class S> constexpr int cols3f_waves() { return S::R1 == 32 ? PSFMC_COLS3F_WAVES_2048 : kColThreads / 64; }
template <class S> constexpr int cols3f_threads() { return 64 * cols3f_waves<S>(); }
template <class S> constexpr bool cols3f_half_table() { return cols3f_waves<S>() == 8; }
template <class S> constexpr int cols3f_table_rows() { return cols3f_half_table<S>() ? S::R1 / 2 : S::R1; }
template <class S> constexpr size_t fused_col3f_lds_used() {
    return ((size_t)cols3f_waves<S>() * fft3g_lds_doubles<S>() + (size_t)cols3f_table_rows<S>() * 64 * 2) * sizeof(double);
}
template <class S> constexpr size_t fused_col3f_lds_bytes() { return fused_col3f_lds_used<S>(); }
// The load pipeline (the next column of a wave loaded into a second register set while the current one goes through its
// inverse transform and its stores, as in k_cols) is decided per side by the WHOLE STEP, not by the kernel alone, which it
// makes faster everywhere (512: 38.5 -> 36.5 us, 1024: 49.7 -> 47.4, 1536: 62 -> 56, 2048: 70 -> 55): step 512^2 -1.0 %,
// 1024^2 +0.5 % (and then +0.5 % over k_cols3<1024>), 1536^2 -7 % (the second set costs the second workgroup of a CU),
// 2048^2 +5.4 % (one wave per SIMD either way).  profiles/r4_cols3f_prefetch.txt, r4_cols3f_prefetch_step_ab.txt.
// PSFMC_COLS3F_PREFETCH: -1 per side (1024, 2048), 0 nowhere, 1 everywhere.
#ifndef PSFMC_COLS3F_PREFETCH
#define PSFMC_COLS3F_PREFETCH -1
#endif
#ifndef PSFMC_COLS3F_WAVES16
#define PSFMC_COLS3F_WAVES16 3
#endif
template <class S> constexpr bool cols3f_prefetch() {
    return PSFMC_COLS3F_PREFETCH < 0 ? (S::R1 == 16 || (S::R1 == 32 && !cols3f_half_table<S>())) : PSFMC_COLS3F_PREFETCH != 0;
}
template <class S> constexpr int cols3f_min_waves() {
    if (cols3f_prefetch<S>()) return S::R1 <= 8 ? 3 : S::R1 <= 16 ? 2 : 1;       // (a second register set for the next column)
    return S::R1 <= 8 ? 4 : S::R1 <= 16 ? PSFMC_COLS3F_WAVES16 : (S::R1 <= 24 || cols3f_half_table<S>()) ? 2 : 1;
}

template <int NY, bool CONVOLVE, class S = Fft3gShape<NY>>
__global__ void __launch_bounds__((cols3f_threads<S>()), (cols3f_min_waves<S>()))
k_cols3f(cd* __restrict__ Tbuf, const cd* __restrict__ Kt, const double* __restrict__ prep,
         const uint8_t* __restrict__ skip, const cd* __restrict__ twy, int plen, int nxh, int n_w, int rg_log2) {
    static_assert(cols3f_shape<S>(), "ny = 8 m x 8 x 8");
    constexpr int R1 = S::R1, NB3 = S::NB3, WPB = cols3f_waves<S>();
    constexpr bool HALF = cols3f_half_table<S>();
    static_assert(NB3 * 8 == R1, "registers");
    static_assert(fused_col3f_lds_bytes<S>() <= 160 * 1024, "a workgroup's LDS");
    extern __shared__ __align__(16) double smem[];
#if PSFMC_COLS_PRIO
    __builtin_amdgcn_s_setprio(PSFMC_COLS_PRIO);
#endif
    const int t = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    double* lds = smem + (size_t)wave * fft3g_lds_doubles<S>();
    cd* tab = reinterpret_cast<cd*>(smem + (size_t)WPB * fft3g_lds_doubles<S>());
    for (int i = threadIdx.x; i < cols3f_table_rows<S>() * 64; i += cols3f_threads<S>()) tab[i] = twy[(i & 63) * (i >> 6)];
    __syncthreads();                                   // once, before any wave can leave
    cd w2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) w2[k] = twy[R1 * (t & 7) * k];
    const cd w1h = twy[(R1 / 2) * t];                  // (HALF: the lane's factor for the table's other half)
    const int rg_mask = (1 << rg_log2) - 1;
    // A column's addresses are a wave-uniform base (scalar registers; row 64 a adds 128 a elements: 64 a is a multiple
    // of the row group) plus ONE per-lane byte offset -- written as base + lane they cost a 64-bit vector address
    // pair per element (k_cols3f<2048> at two waves per SIMD: 124 bytes of scratch; none this way)
    // (ny = 512 keeps base + lane pointers: eight elements per column, and the scalar form measured 42 -> 45 us there)
    constexpr bool kScalar = R1 > 8;
    const int off_t = 2 * t - (t & rg_mask);
    const unsigned off_b = (unsigned)off_t * kCd, kt_b = (unsigned)t * kCd;
    auto t_at = [&](cd* base, unsigned ob, int a) -> cd* {       // element of row 64 a + t of the column at `base`
        if constexpr (kScalar) return at_bytes(base, ob + 128u * kCd * a);
        else return base + off_t + 128 * a;
    };
    auto k_at = [&](const cd* k, unsigned kb, int a) -> const cd* {
        if constexpr (kScalar) return at_bytes(k, kb + 64u * kCd * a);
        else return k + t + 64 * a;
    };
    // (`ob` / `kb` below: the lane offsets made opaque INSIDE the loop body -- hoisted out of the loop as 64-bit values
    // the instruction selector no longer sees "scalar base + 32-bit lane offset" and falls back to a vector address pair
    // per element)
    const int n_cols = n_w * 2 * nxh;
    const int nyp = t_col_len(NY, rg_log2);
    const GroupRange gr = scalar_range(xcd_group_range((n_cols + WPB - 1) / WPB));
    // cols3f_prefetch: the next column of this wave is loaded into a second register set while the current one
    // goes through its inverse transform and its stores (k_cols' load pipeline)
    auto locate = [&](int grp, int& w, int& kx, int& c, bool& live) -> cd* {
        int col = grp * WPB + wave;
        live = col < n_cols;
        col = live ? col : n_cols - 1;
        c = col & 1;
        const int pr = col >> 1;
        kx = __builtin_amdgcn_readfirstlane(pr / n_w);   // (the division runs on the vector unit: back to a scalar register)
        w = pr - kx * n_w;
        if (live && skip && skip[w]) live = false;
        return Tbuf + ((size_t)w * nxh + kx) * 2 * nyp + (c << rg_log2);        // (wave-uniform, in scalar registers)
    };
    if constexpr (cols3f_prefetch<S>() && CONVOLVE) {
        cd nxt[R1];
        int w_n, kx_n, c_n;
        bool live_n = false;
        cd* base_n = Tbuf;
        if (gr.first < gr.end) {
            base_n = locate(gr.first, w_n, kx_n, c_n, live_n);
            const unsigned ob = off_b;
#pragma unroll
            for (int a = 0; a < R1; ++a) nxt[a] = load_stream(t_at(base_n, ob, a));
        }
        for (int grp = gr.first; grp < gr.end; grp += gr.step) {
            unsigned ob = off_b, kb = kt_b;
            if constexpr (kScalar) asm volatile("" : "+v"(ob), "+v"(kb));
            cd v[R1];
#pragma unroll
            for (int a = 0; a < R1; ++a) v[a] = nxt[a];
            cd* base = base_n;
            const int w = w_n, kx = kx_n, c = c_n;
            const bool live = live_n;
            cd o[NB3][8];
            fft_wave3g<S, -1, HALF>(v, o, w2, t, lds, tab, w1h);
            __builtin_amdgcn_sched_barrier(0);
            const int psf = __builtin_amdgcn_readfirstlane(live ? (int)prep[(size_t)w * plen + kPrepPsfIdx] : 0);
            const cd* k = Kt + (((size_t)psf * nxh + kx) * 2 + c) * NY;
#pragma unroll
            for (int q = 0; q < NB3; ++q)
#pragma unroll
                for (int k3 = 0; k3 < 8; ++k3) v[q + NB3 * k3] = cmul(o[q][k3], *k_at(k, kb, q + NB3 * k3));
            __builtin_amdgcn_sched_barrier(0);
            if (grp + gr.step < gr.end) {
                base_n = locate(grp + gr.step, w_n, kx_n, c_n, live_n);
#pragma unroll
                for (int a = 0; a < R1; ++a) nxt[a] = load_stream(t_at(base_n, ob, a));
            }
            __builtin_amdgcn_sched_barrier(0);
            fft_wave3g<S, +1, HALF>(v, o, w2, t, lds, tab, w1h);
            __builtin_amdgcn_sched_barrier(0);
            if (live) {
                unsigned os = off_b;                     // (formed again: 32 offsets kept from the loads would sit in
                if constexpr (kScalar) asm volatile("" : "+v"(os));   //  registers through both transforms)
#pragma unroll
                for (int q = 0; q < NB3; ++q)
#pragma unroll
                    for (int k3 = 0; k3 < 8; ++k3) *t_at(base, os, q + NB3 * k3) = o[q][k3];
            }
        }
        return;
    }
    for (int grp = gr.first; grp < gr.end; grp += gr.step) {
        const int col = grp * WPB + wave;
        if (col >= n_cols) continue;                     // wave-uniform
        const int pr = col >> 1, c = col & 1;           // kx * n_w + walker, component
        const int kx = __builtin_amdgcn_readfirstlane(pr / n_w), w = pr - kx * n_w;    // (the division runs on the vector unit)
        const bool skipped = skip && skip[w];            // wave-uniform; tested once the column's loads are issued
        cd* base = Tbuf + ((size_t)w * nxh + kx) * 2 * nyp + (c << rg_log2);     // (wave-uniform, in scalar registers)
        unsigned ob = off_b, kb = kt_b;
        if constexpr (kScalar) asm volatile("" : "+v"(ob), "+v"(kb));
        cd v[R1];
#pragma unroll
        for (int a = 0; a < R1; ++a) v[a] = load_stream(t_at(base, ob, a));
        if (skipped) continue;
        cd o[NB3][8];
        fft_wave3g<S, -1, HALF>(v, o, w2, t, lds, tab, w1h);        // o[q][k3] = X[t + 64 (q + NB3 k3)]
        if constexpr (CONVOLVE) {
            __builtin_amdgcn_sched_barrier(0);
            const int psf = __builtin_amdgcn_readfirstlane((int)prep[(size_t)w * plen + kPrepPsfIdx]);
            const cd* k = Kt + (((size_t)psf * nxh + kx) * 2 + c) * NY;
#pragma unroll
            for (int q = 0; q < NB3; ++q)
#pragma unroll
                for (int k3 = 0; k3 < 8; ++k3) v[q + NB3 * k3] = cmul(o[q][k3], *k_at(k, kb, q + NB3 * k3));
            __builtin_amdgcn_sched_barrier(0);
            fft_wave3g<S, +1, HALF>(v, o, w2, t, lds, tab, w1h);
            __builtin_amdgcn_sched_barrier(0);
        }
        unsigned os = off_b;                             // (formed again: 32 offsets kept from the loads would sit in
        if constexpr (kScalar) asm volatile("" : "+v"(os));   //  registers through both transforms)
#pragma unroll
        for (int q = 0; q < NB3; ++q)
#pragma unroll
            for (int k3 = 0; k3 < 8; ++k3) *t_at(base, os, q + NB3 * k3) = o[q][k3];
    }
}

// ---------------------------------------------------------------------------
// rows_inv.  grid (ny / RG, n_walkers); one wave per workgroup.
// partial[w][yg] = sum over the wave's good pixels of the chi^2 term.
// conv_out / var_out (optional): [n][ny][nx] images (psfmc_eval_images)
// ---------------------------------------------------------------------------
// MULTI: the context holds several observed fields (psfmc_ctx_create_fields) -- its own instantiation,
// so that the one-field kernel keeps its registers (at nx = 1024 two more kernel arguments pushed the
// scalar registers over their limit and the kernel to one wave per SIMD: 33.7 -> 39.6 us)
template <int NX, typename TS = cd, bool FAST = FftShape<NX>::kPlain, bool MULTI = false>
__global__ void __launch_bounds__((row_threads<NX, FAST>()), (fused_row_min_waves<NX, true>()))
k_rows_inv(const TS* __restrict__ Tbuf, const uint8_t* __restrict__ skip, const cd* __restrict__ twx,
           const FieldPx* __restrict__ field, double* __restrict__ partial, int ny,
           const double* __restrict__ prep, int plen,
           double* __restrict__ conv_out, double* __restrict__ var_out, int n_psf_field, unsigned field_stride) {
    using S = FftShape<NX>;
    constexpr int P = S::P, T = S::T, R = S::R, RG = row_group<NX>();
    constexpr int NXH = NX / 2 + 1;
    constexpr int RGL2 = layout_rg_log2<NX, FAST>(), RGL = 1 << RGL2;
    extern __shared__ __align__(16) double smem[];
#if PSFMC_INV_PRIO
    __builtin_amdgcn_s_setprio(PSFMC_INV_PRIO);
#endif

    int w = blockIdx.y, bx = blockIdx.x;
    if constexpr (inv_remap<NX, FAST>()) {
        // Which workgroup takes which (row group, walker).  Workgroups are dealt to the 8 XCDs round-robin in
        // launch order (x fastest).  With walker = blockIdx.y a row group's field pixels (FieldPx: 16 bytes per
        // pixel, as much as T itself) came through the same XCD's L2 one walker apart -- after 1/8 of (T + field)
        // = 4.2 MB of other lines at nx = 1024: gone from a 4-MiB L2, so every walker of a pass fetched the field
        // again (24.4 MB per walker for a 16.8-MB T: the "1.42x" of rounds 1-2, which was never about half-lines
        // of T) -- or, where the workgroups per walker are not a multiple of 8 (25 at 300^2), through a different
        // XCD for every walker.  Now the walkers go in blocks of WB (all of them up to 24, else 8), and inside a
        // block eight consecutive row groups (one per XCD) of all its walkers run back to back; the gx mod 8 row
        // groups left over take whatever XCD comes (padding the grid to whole rounds instead loaded one XCD with
        // 4 workgroups per walker against 3: k_rows_inv<200> 37 -> 50 us).  Measured: FETCH_SIZE of
        // k_rows_inv<1024> 71.1 -> 60.2 k per launch (1.45x -> 1.23x: the field once per pass), step +1 % (that
        // pass is VALU-bound); 600^2 +2.6 %, 832^2 +4.3 %.  The unguarded kernels of 64 ... 512 keep the plain
        // order: a walker's share of T + field per XCD stays in the L2 (512^2: 1.07x).
        const int gx = (int)gridDim.x, n_w = (int)gridDim.y;
        const int id = w * gx + bx;
        const int WB = n_w <= kInvRemapMaxWalkers ? n_w : 8;
        const int B = id / (WB * gx), rem = id - B * WB * gx;
        const int left = n_w - B * WB, wb = left < WB ? left : WB;
        const int q8 = gx & ~7;                                    // row groups in whole rounds of the XCDs
        int wi;
        if (rem < wb * q8) {
            const int gc = rem / (wb * 8), rr = rem - gc * wb * 8;
            wi = rr >> 3;
            bx = gc * 8 + (rr & 7);
        } else {
            const int rr = rem - wb * q8, j = rr / wb;
            wi = rr - j * wb;
            bx = q8 + j;
        }
        w = B * WB + wi;
    }
    // (testing the flag only after the loads of T were issued, as k_rows_fwd and k_cols3 do, made
    // this kernel slower at 512 and 1024 -- 31.5 -> 37.7 us, 34.1 -> 41.6 us -- and left 256 unchanged)
    if (skip && skip[w]) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int f = lane / T, t = lane % T;
    const int yg = bx * row_waves<NX, FAST>() + wave;
    if constexpr (!FAST)
        if (yg * RG >= ny) return;                                    // wave-uniform
    const bool lane_on = S::kFull || f < RG;
    const int fe = S::kFull ? f : (f < RG ? f : RG - 1);
    const int iy = yg * RG + f;
    const bool row_on = FAST || (lane_on && iy < ny);
    const int nyp = t_col_len(ny, RGL2);
    const int nyg = FAST ? (int)gridDim.x * row_waves<NX, FAST>() : (ny + RG - 1) / RG;
    const TS* wbase = Tbuf + (size_t)w * 2 * NXH * nyp;             // wave-uniform
    constexpr unsigned kEl = sizeof(TS);                             // bytes of a T element
    const unsigned kstride = 2u * (unsigned)nyp * kEl;               // bytes between kx columns

    // Y[k], k = T a + t:  k <= NX/2: G[k] + i H[k];  else conj(G[NX-k]) + i conj(H[NX-k]).
    // Every (G, H) pair is loaded once, by the lane that owns k <= NX/2; that lane also
    // forms the mirrored value and hands it to the owner of NX - k through the transform's
    // LDS region.
    double* wave_lds = smem + (size_t)wave * fused_row_wave_lds_doubles<NX>();
    cd* mbuf = reinterpret_cast<cd*>(wave_lds + (size_t)fe * fft_lds_elems<NX>());
    cd v[R];
#pragma unroll
    for (int a = P; a < R; ++a) v[a] = cd{0.0, 0.0};
    if constexpr (FAST) {
        // the owner of NX - k is lane (T - t) % T at a' = P-1-a (t != 0) or P-a (t == 0);
        // mbuf is [P/2][T] complex
        const unsigned off0 = (unsigned)t_elem(iy, 0, RGL2) * kEl + (unsigned)t * kstride;
        // (nx = 1024: a wave holds 2 rows = 64-byte halves of the 128-byte lines [kx][2 row groups], and the
        // kernel fetches 1.42x its bytes.  Round 3 tried whole lines per wave -- the two waves that share lines
        // each loading every other block of 32 kx columns for both, four consecutive 32-byte pieces per line,
        // and swapping rows through LDS behind one workgroup barrier: the fetch went UP to 1.82x and the kernel
        // from 33.8 to 40.4 us.  The over-fetch is not a matter of which wave touches a line when.)
#pragma unroll
        for (int a = 0; a < P / 2; ++a) {
            const TS* p = at_bytes(wbase, off0 + (unsigned)(T * a) * kstride);
            const cd g = load_stream(p), h = load_stream(p + RGL);
            v[a] = cd{g.x - h.y, g.y + h.x};
            mbuf[a * T + t] = cd{g.x + h.y, h.x - g.y};
        }
        {   // Nyquist column k = NX/2 (lane 0 only); the other lanes take a mirror for a = P/2
            const TS* p = at_bytes(wbase, off0 + (unsigned)(NX / 2) * kstride);   // t == 0: kx = NX/2
            cd g = cd{0.0, 0.0}, h = cd{0.0, 0.0};
            if (t == 0) { g = load_stream(p); h = load_stream(p + RGL); }
            v[P / 2] = cd{g.x - h.y, g.y + h.x};
        }
        wave_lds_sync();
        {
            const int tm = (T - t) % T;
            const int shift = t ? P - 1 : P;
#pragma unroll
            for (int a = P / 2; a < P; ++a) {
                if (a == P / 2 && t == 0) continue;               // lane 0 holds the Nyquist value
                v[a] = mbuf[(shift - a) * T + tm];
            }
        }
    } else {
        // general shapes: the mirror of k goes to LDS slot k (0 < k < NX/2), the owner of
        // k' > NX/2 reads slot NX - k'
        const unsigned off_row = (unsigned)t_elem(row_on ? iy : 0, 0, RGL2) * kEl;
#pragma unroll
        for (int a = 0; a < P; ++a) {
            const int k = T * a + t;
            if (2 * T * a <= NX) {                                // (folds when unrolled) some lane of this register is in the lower half
                const bool low = 2 * k <= NX;
                cd g = cd{0.0, 0.0}, h = cd{0.0, 0.0};
                if (low && row_on) {
                    const TS* p = at_bytes(wbase, off_row + (unsigned)k * kstride);
                    g = load_stream(p);
                    h = load_stream(p + RGL);
                }
                v[a] = cd{g.x - h.y, g.y + h.x};
                if (low && lane_on && k > 0 && 2 * k < NX) mbuf[k] = cd{g.x + h.y, h.x - g.y};
            }
        }
        wave_lds_sync();
#pragma unroll
        for (int a = 0; a < P; ++a) {
            const int k = T * a + t;
            if (2 * (T * a + T - 1) > NX) {                       // (folds) some lane of this register is in the upper half
                if (2 * k > NX) v[a] = mbuf[NX - k];
            }
        }
    }
    wave_lds_sync();
    cd* twl = reinterpret_cast<cd*>(wave_lds + (size_t)RG * fft_lds_elems<NX>());
    cd tw[fft_tw_regs<NX>()];
    load_twiddles<NX>(tw, twx, t, twl, lane);
    fft_wave<NX, +1>(v, tw, twx, t, wave_lds + (size_t)fe * fft_lds_elems<NX>(), twl, lane_on);
    // imaginary part is lambda * model variance (see build_prep)
    const double inv_lambda = prep[(size_t)w * plen + kPrepInvLambda];

    if (conv_out) {
        const size_t rowoff = (size_t)w * ny * NX + (size_t)(row_on ? iy : 0) * NX;
#pragma unroll
        for (int e = 0; e < R; ++e) {
            if (row_on && fft_slot_valid<NX>(t, e)) {
                const int x = fft_k_of<NX>(t, e);
                conv_out[rowoff + x] = v[e].x;
                var_out[rowoff + x] = v[e].y * inv_lambda;
            }
        }
    }
    // sum over the lane's good pixels of r^2 / d + ln(2 pi d), d = model_var + obs_var
    // (= r^2 ivm - ln(ivm / 2 pi), models.py:233-236).  The logarithms of a lane are
    // taken as ONE: ln prod d = ln2 (log2 prod mant(d) + sum exp(d)), with the mantissas in
    // [1/2, 1) so that the product of <= 32 of them cannot underflow -- a log2 is 27
    // instructions, frexp + multiply + integer add are 4.  d <= 0 or NaN gives NaN like the
    // reference's log of a non-positive weight.
    // Waves whose pixels are all good (NaN sci marks an excluded pixel, and a register slot
    // that holds no pixel at all) skip the selects.
    // contexts that hold several observed fields: the walker's field is the quotient of its
    // (field * PSFs-per-field + PSF) index; field_stride = packed pixels of one field
    const FieldPx* fbase = field + (size_t)yg * R * 64;                                // wave-uniform
    // (below nx = 1024 the one kernel serves both kinds of context: n_psf_field = 0 marks one field)
    if (MULTI || (NX < 1024 && n_psf_field > 0))
        fbase += (size_t)((int)prep[(size_t)w * plen + kPrepPsfIdx] / n_psf_field) * field_stride;
    const unsigned foff = (unsigned)lane * (unsigned)sizeof(FieldPx);
    // The field pixels arrive in chunks of CH registers, the next chunk's loads in flight while this
    // one is summed.  One chunk = all R for the small shapes; at R = 32 (nx = 512, 1024) four chunks of
    // 8: all 32 pixels at once were 128 registers on top of the transform's 128 -- the whole budget of a
    // wave at two per SIMD, with scalar registers spilling into what was left.  The sums do not depend
    // on the chunking: good pixels are added in register order by the same instructions either way.
    constexpr int CH = (FAST && R == 32) ? PSFMC_INV_CHUNK : R, NCH = R / CH;
    static_assert(R % CH == 0, "chunks");
    double acc = 0.0, mant = 1.0;
    int expo = 0, n_good = 0;
    bool invalid = false;
    auto load_chunk = [&](FieldPx (&px)[CH], int c) {
#pragma unroll
        for (int j = 0; j < CH; ++j) px[j] = *at_bytes(fbase, foff + (unsigned)((c * CH + j) * 64 * sizeof(FieldPx)));
    };
    auto sum_chunk = [&](const FieldPx (&px)[CH], int c) {
        bool any_bad = false;
#pragma unroll
        for (int j = 0; j < CH; ++j) any_bad |= px[j].sci != px[j].sci;
        if (!__any(any_bad)) {
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const int e = c * CH + j;
                const double d = __builtin_fma(v[e].y, inv_lambda, px[j].var);
                const double r = px[j].sci - v[e].x;
                acc = __builtin_fma(r * r, fast_rcp(d), acc);
                invalid |= !(d > 0.0);
                mant *= __builtin_amdgcn_frexp_mant(d);
                expo += __builtin_amdgcn_frexp_exp(d);
            }
            n_good += CH;
        } else {
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const int e = c * CH + j;
                const bool good = px[j].sci == px[j].sci;
                const double d = __builtin_fma(v[e].y, inv_lambda, px[j].var);
                const double r = px[j].sci - v[e].x;
                const double a1 = __builtin_fma(r * r, fast_rcp(d), acc);
                acc = good ? a1 : acc;
                const double dd = good ? d : 1.0;                // neutral factor
                invalid |= !(dd > 0.0);
                mant *= __builtin_amdgcn_frexp_mant(dd);
                expo += __builtin_amdgcn_frexp_exp(dd);
                n_good += good ? 1 : 0;
            }
        }
    };
    if constexpr (NCH == 1) {
        FieldPx px[CH];
        load_chunk(px, 0);
        sum_chunk(px, 0);
    } else {
        static_assert(NCH % 2 == 0, "chunk pairs");
        FieldPx pa[CH], pb[CH];
        load_chunk(pa, 0);
#pragma unroll
        for (int c = 0; c < NCH; c += 2) {
            load_chunk(pb, c + 1);
            sum_chunk(pa, c);
            if (c + 2 < NCH) load_chunk(pa, c + 2);
            sum_chunk(pb, c + 1);
        }
    }
    acc += 0.69314718055994530942 * (fast_log2(mant) + (double)expo) +
           1.83787706640934548356 * (double)n_good;           // ln(2 pi) per good pixel
    acc = invalid ? __builtin_nan("") : acc;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if (lane == 0) partial[(size_t)w * nyg + yg] = acc;
}

// ---------------------------------------------------------------------------
// Posterior-image sums (models.py:74-97) without transforms.  Every posterior image is linear in
// three per-sample images -- raw, raw^2 and the point-source-only raw model: mean convolved model =
// conv(mean raw), mean model variance = conv_var(mean raw^2) (the reference averages the weight map
// as a variance), mean PS-only convolved = conv(mean raw_PS) -- so a sample only has to be
// rasterised and added to three sums per PSF; the convolutions happen once, when the images are
// asked for (psfmc_hip.hip flush_linear_sums).
// grid (row groups, walker groups), one wave per workgroup: the wave rasterises its RG rows for the
// walkers of its group one after the other with the row kernels' own rasteriser (same bits as the
// likelihood's raw model), keeps the three sums of a 16-pixel segment in registers and stores them
// as the group's partial: part[group][psf][3][ny][NX].  k_sum_partials adds the groups up in order:
// no atomics, the sums do not depend on how the walkers were grouped in time.
// ---------------------------------------------------------------------------
// which lane rasterises which pixel in k_raster_sums: the side's two-stage row shape, or -- sides above 1024,
// which have none -- one row per wave, x = 64 k + lane
constexpr bool two_stage_side(int n) { return n <= 1024; }
template <int NX, bool TWO = two_stage_side(NX)> struct RasterShape {
    static constexpr int T = FftShape<NX>::T, P = FftShape<NX>::P, TPW = FftShape<NX>::TPW;
};
template <int NX> struct RasterShape<NX, false> {
    static_assert(NX % 64 == 0, "side");
    static constexpr int T = 64, P = NX / 64, TPW = 1;
};

template <int P> constexpr int raster_seg() { return P <= 16 ? P : (P % 16 == 0 ? 16 : P % 15 == 0 ? 15 : P % 12 == 0 ? 12 : P % 10 == 0 ? 10 : P % 9 == 0 ? 9 : P % 7 == 0 ? 7 : P % 5 == 0 ? 5 : P); }

template <int NX, int K0, int SEG, bool WRAP>
__device__ __forceinline__ void raster_sums_segment(const double* __restrict__ prep, int plen, int w0, int w1, int psf,
                                                    int n_ps, int n_sersic, int t, int iy, bool row_on,
                                                    double* __restrict__ log_tab, double* __restrict__ out,
                                                    size_t S, const WrapDesc& wr) {
    constexpr int T = RasterShape<NX>::T;
    double a[SEG], b[SEG], cps[SEG];
#pragma unroll
    for (int k = 0; k < SEG; ++k) a[k] = b[k] = cps[k] = 0.0;
    for (int w = w0; w < w1; ++w) {
        const double* wprep = prep + (size_t)w * plen;                   // wave-uniform
        if ((int)wprep[kPrepPsfIdx] != psf) continue;
        double r[SEG];
        // (log2 + exp2 form at every size: a segment is 7 ... 16 pixels per lane, and the power tables would be
        // fetched again for each -- 512^2, 256 walkers: 0.59 ms per iteration of image sums with them, 0.46 without)
        raster_row_logexp<SEG, T, K0, WRAP>(wprep, n_ps, n_sersic, t, iy, false, log_tab, r, wr);
#pragma unroll
        for (int k = 0; k < SEG; ++k) {
            a[k] += r[k];
            b[k] = __builtin_fma(r[k], r[k], b[k]);
        }
        if (n_ps) {
            raster_row_logexp<SEG, T, K0, WRAP>(wprep, n_ps, n_sersic, t, iy, true, log_tab, r, wr);
#pragma unroll
            for (int k = 0; k < SEG; ++k) cps[k] += r[k];
        }
    }
    if (row_on) {
        double* o = out + (size_t)iy * NX + t;
#pragma unroll
        for (int k = 0; k < SEG; ++k) {
            o[T * (K0 + k)] = a[k];
            o[S + T * (K0 + k)] = b[k];
            o[2 * S + T * (K0 + k)] = cps[k];
        }
    }
}

template <int NX, int K0, bool WRAP>
__device__ __forceinline__ void raster_sums_all(const double* __restrict__ prep, int plen, int w0, int w1, int psf,
                                                int n_ps, int n_sersic, int t, int iy, bool row_on,
                                                double* __restrict__ log_tab, double* __restrict__ out, size_t S,
                                                const WrapDesc& wr) {
    constexpr int P = RasterShape<NX>::P, SEG = raster_seg<P>();
    if constexpr (K0 < P) {
        raster_sums_segment<NX, K0, SEG, WRAP>(prep, plen, w0, w1, psf, n_ps, n_sersic, t, iy, row_on, log_tab, out, S,
                                               wr);
        raster_sums_all<NX, K0 + SEG, WRAP>(prep, plen, w0, w1, psf, n_ps, n_sersic, t, iy, row_on, log_tab, out, S, wr);
    }
}

// per_field > 0: the walkers are field-contiguous (per_field of them per field, the first belonging to
// field f0) and a group never straddles two fields: the group then only looks at its own field's npf
// kernel spectra and writes part[group][npf][3][ny][NX] (k_sum_partials_fields adds a field's groups up).
// per_field == 0: any mixture of kernel spectra, part[group][n_psf][3][ny][NX].
template <int NX, bool WRAP = false>
__global__ void __launch_bounds__(64) k_raster_sums(const double* __restrict__ prep, int plen, int n_w, int group_size,
                                                    int n_ps, int n_sersic, int ny, int n_psf,
                                                    double* __restrict__ part, int per_field, int f0, int npf,
                                                    WrapDesc wr) {
    using S = RasterShape<NX>;
    constexpr int T = S::T, RG = S::TPW;
    static_assert(S::P % raster_seg<S::P>() == 0, "segment");
    __shared__ __align__(16) double log_tab[kLogTabBytes / sizeof(double)];
    const int lane = threadIdx.x;
    const int f = lane / T, t = lane % T;
    const int iy = blockIdx.x * RG + f;
    const bool row_on = f < RG && iy < ny;
    const int g = blockIdx.y;
    const int w0 = g * group_size, w1 = w0 + group_size < n_w ? w0 + group_size : n_w;
    load_log_table(log_tab, lane);
    wave_lds_sync();
    const size_t Spx = (size_t)ny * NX;
    const int psf0 = per_field > 0 ? (f0 + w0 / per_field) * npf : 0;      // (wave-uniform)
    const int n_here = per_field > 0 ? npf : n_psf;
    for (int p = 0; p < n_here; ++p)
        raster_sums_all<NX, 0, WRAP>(prep, plen, w0, w1, psf0 + p, n_ps, n_sersic, t, row_on ? iy : 0, row_on, log_tab,
                                     part + ((size_t)g * n_here + p) * 3 * Spx, Spx, wr);
}

// lin[i] += part[0][i] + part[1][i] + ... (fixed order)
#if PSFMC_PART == 0          /* not a template: defined in the API part only */
__global__ void k_sum_partials(const double* __restrict__ part, int n_groups, double* __restrict__ lin, size_t n_el) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_el; i += (size_t)gridDim.x * blockDim.x) {
        double s = 0.0;
        for (int g = 0; g < n_groups; ++g) s += part[(size_t)g * n_el + i];
        lin[i] += s;
    }
}
// field-contiguous walkers (k_raster_sums with per_field > 0): field j of the call owns the groups
// [j gpf, (j + 1) gpf); lin is the block of the call's first field, n_el = elements of ONE field (npf 3 S)
__global__ void k_sum_partials_fields(const double* __restrict__ part, int gpf, int n_fields_here,
                                      double* __restrict__ lin, size_t n_el) {
    const size_t total = n_el * n_fields_here;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t j = i / n_el, r = i - j * n_el;
        double s = 0.0;
        for (int g = 0; g < gpf; ++g) s += part[((size_t)j * gpf + g) * n_el + r];
        lin[i] += s;
    }
}
#endif

// Kt[psf][kx][c][ky] = spec_c[psf][ky][kx] * (-1)^(kx+ky) / S from the
// column-transformed PSF buffer (T layout with ky in place of y; its c = 1 half
// already carries the channel scale rho[psf], which stays in Kt).  k_rows_fwd leaves
// every spectrum doubled: `inv_s` carries 1/2 for the PSF's own doubling and 1/2 for
// the doubling of the model spectra it will multiply.
#if PSFMC_PART == 0          /* not a template: defined in the API part only */
__global__ void k_scale_kernel_spectrum(const cd* __restrict__ raw, cd* __restrict__ Kt, int n_total,
                                        int ny, int nxh, int rg_log2, double inv_s) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_total; i += gridDim.x * blockDim.x) {
        const int ky = i % ny;
        const int c = (i / ny) & 1;
        const int pk = i / (2 * ny);                    // psf*nxh + kx
        const int kx = pk % nxh;
        const double sc = ((kx + ky) & 1) ? -inv_s : inv_s;
        const cd v = raw[(size_t)pk * 2 * t_col_len(ny, rg_log2) + t_elem(ky, c, rg_log2)];
        Kt[i] = cd{v.x * sc, v.y * sc};
    }
}
#endif

// natural-layout copy for psfmc_get_spectra: out[psf][ky][kx] of component c
#if PSFMC_PART == 0          /* not a template: defined in the API part only */
__global__ void k_untranspose_spectrum(const cd* __restrict__ raw, cd* __restrict__ out, int n_psf,
                                       int c, int ny, int nxh, int rg_log2,
                                       const double* __restrict__ rho) {
    const int n = n_psf * ny * nxh;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int kx = i % nxh;
        const int ky = (i / nxh) % ny;
        const int p = i / (nxh * ny);
        const double sc = 0.5 * (c ? 1.0 / rho[p] : 1.0);      // k_rows_fwd doubles
        const cd v = raw[((size_t)p * nxh + kx) * 2 * t_col_len(ny, rg_log2) + t_elem(ky, c, rg_log2)];
        out[i] = cd{v.x * sc, v.y * sc};
    }
}
#endif

}  // namespace psfmc
