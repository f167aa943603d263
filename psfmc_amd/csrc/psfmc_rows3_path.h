// psfmc_rows3_path.h -- row kernels on the wave-wide three-stage engine (psfmc_fft.h fft_wave3g): ONE row per
// wave, nx = R1 (R2 R3) with all L = R2 R3 <= 64 lanes of the wave on that one transform and R1 complex
// registers per lane (16 at nx = 1024 where the two-stage engine of psfmc_fused_path.h holds 32 and two rows).
//
// Why (round 4).  A two-stage row wave at nx = 512 / 1024 holds 128 registers of transform data and 16.9 KB of
// exchange LDS: two waves per SIMD, every wave of a launch in the same phase (rasterise, transform, store), a
// launch of 1.5 rounds of such waves, and nothing of another kernel beside them.  Here a wave needs R1 complex
// registers and N doubles of LDS: three or four waves per SIMD in different phases, whole rounds.  The same
// kernels carry the sides the two-stage shapes P, T <= 32 cannot (above 1024: R1 up to 32) and the sides whose
// two-stage row shape has P >= 25.
//
// Layout of T as in psfmc_fused_path.h with 4-row groups: [walker][kx][yg][c][r], y = 4 yg + r.  The four waves
// of a workgroup own the four rows of one group, so the 16-byte pieces they store into (load from) a 128-byte
// line [kx][yg] meet in one CU's L2 slice at about the same time; there is no LDS tile and no barrier for it
// (round 3's experiment at nx = 512 exchanged whole lines through LDS behind four barriers per group and had
// 41 % of its wave-cycles parked).  ONE workgroup barrier per kernel: the stage-1 twiddle table [k1][lane] the
// four waves share.
//
// Forward: rasterise the row into registers as z = raw + i mu raw^2 (x = L a + lane), transform, untangle the two
// Hermitian spectra through the wave's LDS region, store kx <= nx / 2.  Inverse: load G, H for kx = L a + lane <=
// nx / 2, rebuild Y = G + i H and its mirrored half, transform with the FORWARD engine and conjugated twiddles
// (decimation in frequency both ways: a row kernel has no transform to mirror), fused chi^2 over the lane's
// pixels x = (lane + 64 q) + R1 R2 k3.
// Reference: psfMC/models.py:213-216, 233-236; utils.py:25-32.
#pragma once
#include "psfmc_fused_path.h"

namespace psfmc {

// Rows (= waves) of a workgroup: 4 = one layout group of the general sides.  At nx = 2048 a wave's exchange region is
// 18 KB and the stage-1 table the workgroup shares 32 KB: four waves = 104 KB = ONE workgroup and one wave per SIMD on
// a CU.  EIGHT waves around HALF the table (psfmc_fft.h fft_wave3g HALF1: the other half is the lane's one factor
// times the first) are exactly the CU's 160 KB: two waves per SIMD, and the 2048 rows of a walker are one whole round
// of the chip's 256 CUs (PSFMC_ROWS3_WAVES_2048: 4 / 6 / 7 with the whole table for comparison; the layout above 1024
// has no row groups to respect).
#ifndef PSFMC_ROWS3_WAVES_2048
#define PSFMC_ROWS3_WAVES_2048 8
#endif
// (1152 / 1280 with six / five waves per workgroup and the kernels bounded to three waves per SIMD: the bound costs
// scratch, inverse kernel 41 -> 73 us, step -32 %: profiles/r4_rows3_waves_1152.txt)
#ifndef PSFMC_ROWS3_WAVES_1280
#define PSFMC_ROWS3_WAVES_1280 4
#endif
constexpr int rows3_waves(int n) { return n == 2048 ? PSFMC_ROWS3_WAVES_2048 : n == 1280 ? PSFMC_ROWS3_WAVES_1280 : 4; }
constexpr int rows3_threads(int n) { return 64 * rows3_waves(n); }
// nx = 1152: half the table takes a four-wave workgroup from 59.9 to 50.7 KB -- THREE workgroups per CU; the forward
// kernel (166 registers) then runs three waves per SIMD: 41.6 -> 37.7 us, step +4 % (profiles/r4_rows3_half_1152.txt)
#ifndef PSFMC_ROWS3_HALF_1152
#define PSFMC_ROWS3_HALF_1152 1
#endif
// (nx = 1280: 56 KB with half the table is still two workgroups per CU; FIVE waves around it -- 67.6 KB, ten waves per
// CU -- measured 37.5 -> 50 us forward, 36 -> 48 inverse, step -20 %: profiles/r4_rows3_half_1280.txt.  Left alone.)
#ifndef PSFMC_ROWS3_HALF_1280
#define PSFMC_ROWS3_HALF_1280 0
#endif
constexpr bool rows3_half_table(int n) {
    return (n == 2048 && rows3_waves(n) == 8) || (PSFMC_ROWS3_HALF_1152 && n == 1152) || (PSFMC_ROWS3_HALF_1280 && n == 1280);
}
template <class S> constexpr int rows3_table_rows() { return rows3_half_table(S::kN) ? S::R1 / 2 : S::R1; }
// Layout groups.  The sides that also have two-stage row kernels share their guarded layout: groups of 4 rows,
// [kx][yg][c][r].  The sides above 1024 have no other row kernels and take groups of ONE row, [kx][y][c]: the two
// components of a (kx, y) are then 32 adjacent bytes, and a lane PAIR (kx even / odd) trades one value so that every
// store (load) instruction moves whole 32-byte sectors -- one lane the c = 0 half, its neighbour the c = 1 half --
// instead of 16-byte pieces of two sectors 64 bytes apart: the memory system charges per sector touched, 2.6 TB/s for
// 16-byte pieces against 5.1 TB/s for 32-byte ones (tools/store_pattern_probe.hip, profiles/r4_store_pattern_probe.txt;
// the column kernels then see 16 bytes every 32: 5.4 against 6.4 TB/s there).  PSFMC_ROWS3_BIG_RG_LOG2 = 2 builds the
// old form for comparison.
#ifndef PSFMC_ROWS3_BIG_RG_LOG2
#define PSFMC_ROWS3_BIG_RG_LOG2 0
#endif
constexpr int rows3_rg_log2(int n) { return n > 1024 ? PSFMC_ROWS3_BIG_RG_LOG2 : 2; }

// the row shape of a side: {R2, R3}, R1 = nx / (R2 R3); {0, 0} = the side has no three-stage row kernels
#ifndef PSFMC_ROWS3_EXTRA
#define PSFMC_ROWS3_EXTRA 0         /* 1: also build the candidates below (tools/rows3_probe.hip; PSFMC_ROWS3=1 in the environment selects them) */
#endif
// Which sides have these kernels, and which take them by default, is by measurement (tools/rows3_probe.hip on an
// MI355X, profiles/r4_rows3_probe_*.txt, same prep records, pass-sized batches, us per launch two-stage ->
// three-stage):
//   inverse   676: 48.2 -> 40.9   728: 53.9 -> 38.4   780: 52.7 -> 43.0   784: 57.6 -> 41.6   840: 55.8 -> 37.5
//             900: 63.3 -> 42.3   (their two-stage kernels hold 26 ... 30 complex registers per lane: one wave per
//             SIMD, or two with 72 ... 128 bytes of scratch); 630 / 650 / 700: 35.7 -> 40.7, 38.9 -> 44.1, 50.4 -> 48.8
//             (two waves without much scratch: no gain); 512 / 1024: 32.1 -> 40.2, 33.8 -> 42.4 (SLOWER)
//             second survey (profiles/r4_rows3_probe_survey.txt: every side with a three-stage column shape, two Sersic
//             components): the inverse kernel alone gains 8 ... 28 % at 308, 330, 364, 384, 392, 440, 484, 500, 520, 560,
//             572, 600, 660, 720, 832 -- but IN THE STEP (one component, tools/side_costs.py, default against PSFMC_ROWS3=0:
//             profiles/r4_inv3_more_sides_*.jsonl) only 720 gains (+3.6 %); the others move -4.5 ... +1.0 %: a kernel that
//             is faster alone but leaner in registers changes how the two lanes' kernels share the chip.  720 taken.
//   forward   728: 62.9 -> 61.6   840: 64.8 -> 63.0   900: 71.2 -> 72.6   (even) and SLOWER everywhere else
//             (630: 42.4 -> 58.7 ... 1024: 72.4 -> 94.5, 512: 46.5 -> 64.9): a wave's stores are 16-byte pieces of
//             lines where the two-stage wave's are 32 ... 64 bytes, its store phase takes twice as long (26 vs 13.7 us
//             at 1024^2 / 6 walkers, measured by compiling the phases out), and the table traffic per pixel doubles.
// So: the inverse kernel for the six sides above and 720, both kernels for the sides above 1024 (nothing else
// reaches them).
constexpr Fft3gPick rows3_pick(int n) {
    switch (n) {
        // sides above 1024: the only row kernels there are (the two-stage shapes end at P = T = 32)
        case 1152: case 1280: case 1536: case 2048: return {8, 8};
        // sides whose two-stage row shape holds 26 ... 30 complex registers per lane.  (The shapes the column kernel's
        // re-survey preferred -- (7, 8), (6, 10), (13, 4) -- were tried here too, profiles/r4_rows3_inv_shapes.txt: the
        // inverse kernel 0 ... +15 % SLOWER except 728 (-2 %), the step -3 ... +1 %.  These stay.)
        case 676: case 780: return {4, 13};
        case 728: case 784: case 840: return {4, 14};
        case 900: return {4, 15};
        // second survey (every side with a three-stage column shape): the one more side whose whole step gains
        case 720: return fft3g_pick(n);
#if PSFMC_ROWS3_EXTRA
        case 512: case 1024: return {8, 8};
        case 650: case 700: return {5, 10};
        case 630: return {7, 9};
#endif
#if PSFMC_ROWS3_EXTRA > 1
        default: return n > 256 ? fft3g_pick(n) : Fft3gPick{0, 0};      // (survey builds: every side with a three-stage shape)
#else
        default: return {0, 0};
#endif
    }
}
constexpr bool rows3_only_side(int n) { return n > 1024; }
// the inverse kernel is the default where it measured faster; the forward kernel only where there is no other
// (the sides whose two-stage row shapes round 4 re-surveyed for lanes now hold 24 ... 32 complex registers per lane:
// the inverse kernel there, profiles/r4_inv3_reshaped_sides.txt -- faster alone at 330 and 350 only (9.5 -> 8.0, 8.9 -> 7.3
// ps per pixel), slower at 264, 312, 352, 416, and the whole step -1 ... -14 % at all six.  Not taken.)
#ifndef PSFMC_ROWS3_INV_MORE
#define PSFMC_ROWS3_INV_MORE 0      /* survey builds (with PSFMC_ROWS3_EXTRA=2): the inverse kernel also at the sides listed below */
#endif
constexpr bool rows3_inv_default(int n) {
    if (PSFMC_ROWS3_INV_MORE && (n == 250 || n == 264 || n == 312 || n == 330 || n == 350 || n == 352 || n == 416)) return true;
    return rows3_only_side(n) || n == 676 || n == 720 || n == 728 || n == 780 || n == 784 || n == 840 || n == 900;
}
constexpr bool rows3_fwd_built(int n) { return rows3_pick(n).r2 > 0 && (rows3_only_side(n) || PSFMC_ROWS3_EXTRA); }
template <int NX> struct Rows3 {
    using S = Fft3gShape<NX, rows3_pick(NX).r2, rows3_pick(NX).r3>;
    static constexpr bool kBuilt = S::kBuilt;
    static constexpr int NSLOT = S::NB3 * S::R3;      // output registers per lane: pixel slots of the inverse kernel
};
template <int NX> constexpr bool rows3_side() { return Rows3<NX>::kBuilt; }

template <class S> constexpr size_t rows3_wave_lds_doubles() {
    constexpr size_t fft = fft3g_lds_doubles<S>();          // >= N doubles: also holds the N / 2 mirror values
    constexpr size_t ras = (size_t)kRasterLdsDoubles;
    return fft > ras ? fft : ras;
}
template <class S> constexpr size_t rows3_lds_bytes() {
    return ((size_t)rows3_waves(S::kN) * rows3_wave_lds_doubles<S>() + (size_t)rows3_table_rows<S>() * 64 * 2) * sizeof(double);
}
#ifndef PSFMC_ROWS3_WAVES16
#define PSFMC_ROWS3_WAVES16 3       /* waves per SIMD the kernels with 9 ... 16 complex registers per lane are compiled for */
#endif
template <class S> constexpr bool rows3_lds_fits() { return rows3_lds_bytes<S>() <= 160 * 1024; }
template <class S, bool INVERSE> constexpr int rows3_min_waves() {
    return S::R1 <= 8 ? 4 : S::R1 <= 16 ? PSFMC_ROWS3_WAVES16 : (S::R1 <= 24 || rows3_waves(S::kN) > 4) ? 2 : 1;
}
#ifndef PSFMC_ROWS3_RASTER_GROUP
#define PSFMC_ROWS3_RASTER_GROUP 1
#endif
#ifndef PSFMC_ROWS3_PREFETCH
#define PSFMC_ROWS3_PREFETCH 0      /* the next component's parameters and tables in flight during this one's pixels
                                       (measured: 840 -6 %, 900 0, 1024 +3 % kernel time; 15 registers) */
#endif
#ifndef PSFMC_DEBUG_ROWS3
#define PSFMC_DEBUG_ROWS3 0          /* timing experiments: 1 = no store phase, 2 = no transform, 4 = no rasteriser */
#endif

template <int NX> constexpr size_t rows3_field_len(int ny) { return (size_t)ny * Rows3<NX>::NSLOT * 64; }

// out[(y NSLOT + e) 64 + lane] = pixel (y, x = (lane + 64 q) + R1 R2 k3), e = q R3 + k3; slots that hold no pixel
// are marked excluded (NaN sci, unit variance) like k_pack_field's
template <int NX>
__global__ void k_pack_field3(const double* __restrict__ sci, const double* __restrict__ obs_var,
                              const uint8_t* __restrict__ bad, FieldPx* __restrict__ out, int ny) {
    using S = typename Rows3<NX>::S;
    constexpr int NSLOT = Rows3<NX>::NSLOT;
    const size_t n = rows3_field_len<NX>(ny);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63), e = (int)((i >> 6) % NSLOT), y = (int)(i / (64 * NSLOT));
        const int q = e / S::R3, k3 = e % S::R3;
        const bool holds = fft3g_valid<S>(lane, q);
        const size_t src = holds ? (size_t)y * NX + fft3g_index<S>(lane, q, k3) : 0;
        out[i] = holds ? FieldPx{bad[src] ? __builtin_nan("") : sci[src], obs_var[src]}
                       : FieldPx{__builtin_nan(""), 1.0};
    }
}

// the value of the neighbouring lane (lane ^ 1): DPP quad_perm [1, 0, 3, 2] on the two halves of a double
__device__ __forceinline__ double swap_adjacent_lanes(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    int lo = (int)b, hi = (int)(b >> 32);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0xB1, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xF, 0xF, false);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned)lo);
}
// a lane pair's trade: the even lane gives `from_even` and the odd lane `from_odd`; each gets the other's
__device__ __forceinline__ cd pair_trade(bool odd, cd from_even, cd from_odd) {
    const double sx = odd ? from_odd.x : from_even.x, sy = odd ? from_odd.y : from_even.y;
    return cd{swap_adjacent_lanes(sx), swap_adjacent_lanes(sy)};
}
__device__ __forceinline__ cd pick(bool second, cd a, cd b) { return cd{second ? b.x : a.x, second ? b.y : a.y}; }

// the workgroup's stage-1 twiddle table [k1][lane] = W_N^(lane k1) (zero for the lanes past L)
template <class S>
__device__ __forceinline__ void rows3_fill_table(cd* __restrict__ tab, const cd* __restrict__ twx) {
    for (int i = threadIdx.x; i < rows3_table_rows<S>() * 64; i += rows3_threads(S::kN))
        tab[i] = (i & 63) < S::L ? twx[(i & 63) * (i >> 6)] : cd{0.0, 0.0};
}

// ---------------------------------------------------------------------------
// rows3_fwd.  grid (ceil(ny / 4), n_walkers), 4 waves per workgroup, wave = row of the layout group.
// Arguments as k_rows_fwd.
// ---------------------------------------------------------------------------
template <int NX, bool FROM_IMAGE, bool WRAP = false, class S = typename Rows3<NX>::S>
__global__ void __launch_bounds__((rows3_threads(NX)), (rows3_min_waves<S, false>()))
k_rows3_fwd(const double* __restrict__ prep, const uint8_t* __restrict__ skip, const cd* __restrict__ twx,
            cd* __restrict__ Tbuf, int n_ps, int n_sersic, int ny, int ps_only, const double* __restrict__ img,
            const double* __restrict__ img_scale, double* __restrict__ raw_out, WrapDesc wr, int pow_mode) {
    static_assert(pow_tabs_side(NX), "the three-stage row kernels rasterise with the power tables");
    static_assert(rows3_lds_fits<S>(), "a workgroup's LDS");
    constexpr int R1 = S::R1, R2 = S::R2, R3 = S::R3, L = S::L, NB3 = S::NB3;
    constexpr int RGL2 = rows3_rg_log2(NX), NXH = NX / 2 + 1, RGL = 1 << RGL2;
    constexpr int WAVES = rows3_waves(NX);
    constexpr bool kPair = RGL2 == 0 && L == 64;                      // lane pairs store whole sectors (see rows3_rg_log2)
    static_assert(!kPair || (R1 * R2) % 2 == 0, "lane pairs hold outputs together");
    extern __shared__ __align__(16) double smem[];
    const int w = blockIdx.y;
    if (skip && skip[w]) return;                                      // (workgroup-uniform)
#if PSFMC_FWD_PRIO_STAGGER
    switch (((blockIdx.y * gridDim.x + blockIdx.x) / PSFMC_FWD_PRIO_STAGGER) & 3) {      // (the operand is an immediate)
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        case 3: __builtin_amdgcn_s_setprio(3); break;
        default: break;
    }
#endif
    const int t = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    cd* tab = reinterpret_cast<cd*>(smem + (size_t)WAVES * rows3_wave_lds_doubles<S>());
    rows3_fill_table<S>(tab, twx);
    double* wave_lds = smem + (size_t)wave * rows3_wave_lds_doubles<S>();
    const int iy = blockIdx.x * WAVES + wave;
    const bool row_in = iy < ny;                                      // wave-uniform
    const bool lane_in = L == 64 || t < L;
    const int tl = lane_in ? t : 0;
    const size_t Spx = (size_t)ny * NX;
    cd v[R1];
    if constexpr (FROM_IMAGE) {
        const double* a = img + (size_t)(2 * w) * Spx + (size_t)(row_in ? iy : 0) * NX;
        const double* b = a + Spx;
        const double sc = img_scale[w];
#pragma unroll
        for (int k = 0; k < R1; ++k) v[k] = (row_in && lane_in) ? cd{a[L * k + tl], b[L * k + tl] * sc} : cd{0.0, 0.0};
    } else {
        const double* wprep = prep + (size_t)w * prep_len(n_ps, n_sersic);   // wave-uniform
        const double mu = wprep[kPrepMu];
#pragma unroll
        for (int k = 0; k < R1; ++k) v[k] = cd{0.0, 0.0};
        if (row_in) {
            // the rasteriser's tables borrow the wave's transform region, idle until the transform begins
            if (!ps_only && n_sersic > 0) load_a_table(wave_lds, t);
            double r[R1];
#if PSFMC_DEBUG_ROWS3 & 4
#pragma unroll
            for (int k = 0; k < R1; ++k) r[k] = wprep[0] * (double)(t + k);
#else
            raster_row<R1, L, 0, WRAP, WRAP ? 1 : PSFMC_ROWS3_RASTER_GROUP, PSFMC_ROWS3_PREFETCH != 0>(
                wprep, n_ps, n_sersic, tl, iy, ps_only != 0, wave_lds, r, wr, pow_mode);
#endif
            wave_lds_sync();
#pragma unroll
            for (int k = 0; k < R1; ++k) v[k] = cd{r[k], mu * r[k] * r[k]};
            if (raw_out && lane_in) {
                double* o = raw_out + (size_t)w * Spx + (size_t)iy * NX;
#pragma unroll
                for (int k = 0; k < R1; ++k) o[L * k + t] = v[k].x;
            }
        }
    }
    __syncthreads();                                                  // the twiddle table is complete
    if (!row_in) return;
    cd w2[R2];
#pragma unroll
    for (int k = 0; k < R2; ++k) w2[k] = twx[R1 * (tl % R3) * k];
    cd o[NB3][R3];
#if PSFMC_DEBUG_ROWS3 & 2
#pragma unroll
    for (int q = 0; q < NB3; ++q)
#pragma unroll
        for (int k3 = 0; k3 < R3; ++k3) o[q][k3] = v[(q * R3 + k3) % R1];
#else
    fft_wave3g<S, -1, rows3_half_table(NX)>(v, o, w2, t, wave_lds, tab, twx[(R1 / 2) * tl]);   // o[q][k3] = Z[(t + 64 q) + R1 R2 k3]
#endif
#if PSFMC_DEBUG_ROWS3 & 1
    {
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < NB3; ++q)
#pragma unroll
            for (int k3 = 0; k3 < R3; ++k3) acc += o[q][k3].x + o[q][k3].y;
        if (acc == 1.2345e300) Tbuf[0] = cd{};
        return;
    }
#endif
    // Untangle.  Every Z[k] with k > NX/2 goes to LDS slot NX - k; the holder of k <= NX/2 reads its mirror from
    // slot k (k = 0 and k = NX/2 are their own mirrors).  TWICE the spectra of raw and mu raw^2, as k_rows_fwd.
    cd* ubuf = reinterpret_cast<cd*>(wave_lds);
#pragma unroll
    for (int q = 0; q < NB3; ++q)
#pragma unroll
        for (int k3 = 0; k3 < R3; ++k3) {
            const int k = fft3g_index<S>(t, q, k3);
            if (fft3g_valid<S>(t, q) && 2 * k > NX) ubuf[NX - k] = o[q][k3];
        }
    wave_lds_sync();
    const int nyp = t_col_len(ny, RGL2);
    cd* wbase = Tbuf + (size_t)w * 2 * NXH * nyp;                     // wave-uniform
    const unsigned kstride = 2u * (unsigned)nyp * kCd;                // bytes between kx columns
    const unsigned off_row = (unsigned)t_elem(iy, 0, RGL2) * kCd;
    if constexpr (kPair) {
        // lanes 2j / 2j + 1 hold kx = k_e / k_e + 1.  The even lane hands H[k_e] over and takes G[k_e + 1]: the first
        // instruction then writes (k_e; c = 0, c = 1) from the pair, the second (k_e + 1; c = 0, c = 1) -- 32
        // adjacent bytes each, the component = the lane's parity
        const bool odd = (t & 1) != 0;
        const unsigned off_pair = off_row + (odd ? kCd : 0u);
#pragma unroll
        for (int q = 0; q < NB3; ++q)
#pragma unroll
            for (int k3 = 0; k3 < R3; ++k3) {
                const int k = fft3g_index<S>(t, q, k3);
                if (2 * (k - t) <= NX) {                              // (folds) lane 0 of this register is in the lower half
                    // (R1 R2 is even: the two lanes of a pair hold an output or not together)
                    const bool held = fft3g_valid<S>(t, q), low = held && 2 * k <= NX;
                    const cd zk = o[q][k3];
                    const bool self = k == 0 || 2 * k == NX;
                    cd zm = ubuf[(self || !low) ? 1 : k];
                    if (self) zm = zk;
                    const cd g = cd{zk.x + zm.x, zk.y - zm.y}, h = cd{zk.y + zm.y, zm.x - zk.x};
                    const cd got = pair_trade(odd, h, g);
                    const cd first = pick(odd, g, got), second = pick(odd, got, h);
                    const int ke = k & ~1, ko = k | 1;
                    if (held && 2 * ke <= NX) *at_bytes(wbase, off_pair + (unsigned)ke * kstride) = first;
                    if (held && 2 * ko <= NX) *at_bytes(wbase, off_pair + (unsigned)ko * kstride) = second;
                }
            }
        return;
    }
#pragma unroll
    for (int q = 0; q < NB3; ++q)
#pragma unroll
        for (int k3 = 0; k3 < R3; ++k3) {
            const int k = fft3g_index<S>(t, q, k3);
            if (fft3g_valid<S>(t, q) && 2 * k <= NX) {
                const cd zk = o[q][k3];
                const bool self = k == 0 || 2 * k == NX;
                cd zm = ubuf[self ? 1 : k];
                if (self) zm = zk;
                cd* dst = at_bytes(wbase, off_row + (unsigned)k * kstride);
                dst[0] = cd{zk.x + zm.x, zk.y - zm.y};
                dst[RGL] = cd{zk.y + zm.y, zm.x - zk.x};
            }
        }
}

// ---------------------------------------------------------------------------
// rows3_inv.  grid (ceil(ny / 4), n_walkers); partial[w][y] = the row's chi^2 sum (ny partials per walker).
// Arguments as k_rows_inv.
// ---------------------------------------------------------------------------
template <int NX, bool MULTI = false, class S = typename Rows3<NX>::S>
__global__ void __launch_bounds__((rows3_threads(NX)), (rows3_min_waves<S, true>()))
k_rows3_inv(const cd* __restrict__ Tbuf, const uint8_t* __restrict__ skip, const cd* __restrict__ twx,
            const FieldPx* __restrict__ field, double* __restrict__ partial, int ny,
            const double* __restrict__ prep, int plen, double* __restrict__ conv_out, double* __restrict__ var_out,
            int n_psf_field, unsigned field_stride) {
    constexpr int R1 = S::R1, R2 = S::R2, R3 = S::R3, L = S::L, NB3 = S::NB3, NSLOT = NB3 * R3;
    constexpr int RGL2 = rows3_rg_log2(NX), NXH = NX / 2 + 1, RGL = 1 << RGL2;
    constexpr int WAVES = rows3_waves(NX);
    constexpr bool kPair = RGL2 == 0 && L == 64;                      // lane pairs load whole sectors (see rows3_rg_log2)
    extern __shared__ __align__(16) double smem[];
    int w = blockIdx.y, bx = blockIdx.x;
    {   // which workgroup takes which (row group, walker): k_rows_inv's order (eight consecutive row groups, one
        // per XCD, of all walkers of a block back to back: the field pixels of a row group meet in one L2)
        const int gx = (int)gridDim.x, n_w = (int)gridDim.y;
        const int id = w * gx + bx;
        const int WB = n_w <= kInvRemapMaxWalkers ? n_w : 8;
        const int B = id / (WB * gx), rem = id - B * WB * gx;
        const int left = n_w - B * WB, wb = left < WB ? left : WB;
        const int q8 = gx & ~7;
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
    if (skip && skip[w]) return;                                      // (workgroup-uniform)
    const int t = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    cd* tab = reinterpret_cast<cd*>(smem + (size_t)WAVES * rows3_wave_lds_doubles<S>());
    rows3_fill_table<S>(tab, twx);
    double* wave_lds = smem + (size_t)wave * rows3_wave_lds_doubles<S>();
    const int iy = bx * WAVES + wave;
    const bool row_in = iy < ny;                                      // wave-uniform
    const bool lane_in = L == 64 || t < L;
    const int tl = lane_in ? t : 0;
    const int nyp = t_col_len(ny, RGL2);
    const cd* wbase = Tbuf + (size_t)w * 2 * NXH * nyp;               // wave-uniform
    const unsigned kstride = 2u * (unsigned)nyp * kCd;
    // Y[k], k = L a + t:  k <= NX/2: G[k] + i H[k];  else conj(G[NX-k]) + i conj(H[NX-k]).  Every (G, H) pair is
    // loaded once, by the lane that owns k <= NX/2; that lane also forms the mirrored value and hands it to the
    // owner of NX - k through the wave's LDS region (slot k, 0 < k < NX/2).
    cd* mbuf = reinterpret_cast<cd*>(wave_lds);
    cd v[R1];
    if (row_in) {
        const unsigned off_row = (unsigned)t_elem(iy, 0, RGL2) * kCd;
        const bool odd = (t & 1) != 0;
        const unsigned off_pair = off_row + (odd ? kCd : 0u);
#pragma unroll
        for (int a = 0; a < R1; ++a) {
            const int k = L * a + tl;
            v[a] = cd{0.0, 0.0};
            if constexpr (kPair) {
                // the pair's two instructions fetch (k_e; c = 0, c = 1) and (k_e + 1; c = 0, c = 1), the component =
                // the lane's parity; the even lane then takes H[k_e] and hands G[k_e + 1] over
                if (2 * L * a <= NX) {
                    const int ke = k & ~1, ko = k | 1;
                    cd first = cd{0.0, 0.0}, second = cd{0.0, 0.0};
                    if (2 * ke <= NX) first = load_stream(at_bytes(wbase, off_pair + (unsigned)ke * kstride));
                    if (2 * ko <= NX) second = load_stream(at_bytes(wbase, off_pair + (unsigned)ko * kstride));
                    const cd got = pair_trade(odd, second, first);
                    const cd g = pick(odd, first, got), h = pick(odd, got, second);
                    const bool low = 2 * k <= NX;
                    if (low) v[a] = cd{g.x - h.y, g.y + h.x};
                    if (low && k > 0 && 2 * k < NX) mbuf[k] = cd{g.x + h.y, h.x - g.y};
                }
            } else
            if (2 * L * a <= NX) {                                    // (folds) some lane of this register is in the lower half
                const bool low = lane_in && 2 * k <= NX;
                cd g = cd{0.0, 0.0}, h = cd{0.0, 0.0};
                if (low) {
                    const cd* p = at_bytes(wbase, off_row + (unsigned)k * kstride);
                    g = load_stream(p);
                    h = load_stream(p + RGL);
                }
                v[a] = cd{g.x - h.y, g.y + h.x};
                if (low && k > 0 && 2 * k < NX) mbuf[k] = cd{g.x + h.y, h.x - g.y};
            }
        }
        wave_lds_sync();
#pragma unroll
        for (int a = 0; a < R1; ++a) {
            const int k = L * a + tl;
            if (2 * (L * a + L - 1) > NX) {                           // (folds) some lane of this register is in the upper half
                if (lane_in && 2 * k > NX) v[a] = mbuf[NX - k];
            }
        }
        wave_lds_sync();
    }
    __syncthreads();                                                  // the twiddle table is complete
    if (!row_in) return;
    cd w2[R2];
#pragma unroll
    for (int k = 0; k < R2; ++k) w2[k] = twx[R1 * (tl % R3) * k];
    cd o[NB3][R3];
    fft_wave3g<S, +1, rows3_half_table(NX)>(v, o, w2, t, wave_lds, tab, twx[(R1 / 2) * tl]);   // o[q][k3] = y[(t + 64 q) + R1 R2 k3]
    // imaginary part is lambda * model variance (see build_prep)
    const double inv_lambda = prep[(size_t)w * plen + kPrepInvLambda];
    if (conv_out) {
        const size_t rowoff = (size_t)w * ny * NX + (size_t)iy * NX;
#pragma unroll
        for (int q = 0; q < NB3; ++q)
#pragma unroll
            for (int k3 = 0; k3 < R3; ++k3)
                if (fft3g_valid<S>(t, q)) {
                    const int x = fft3g_index<S>(t, q, k3);
                    conv_out[rowoff + x] = o[q][k3].x;
                    var_out[rowoff + x] = o[q][k3].y * inv_lambda;
                }
    }
    // chi^2 + log term of the lane's good pixels, the logarithms taken as ONE per lane: k_rows_inv's arithmetic
    const FieldPx* fbase = field + (size_t)iy * NSLOT * 64;           // wave-uniform
    if (MULTI || n_psf_field > 0)
        fbase += (size_t)((int)prep[(size_t)w * plen + kPrepPsfIdx] / (n_psf_field > 0 ? n_psf_field : 1)) * field_stride;
    const unsigned foff = (unsigned)t * (unsigned)sizeof(FieldPx);
    constexpr int CH = NSLOT % 16 == 0 ? 8 : NSLOT, NCH = NSLOT / CH;
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
                const cd z = o[e / R3][e % R3];
                const double d = __builtin_fma(z.y, inv_lambda, px[j].var);
                const double r = px[j].sci - z.x;
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
                const cd z = o[e / R3][e % R3];
                const bool good = px[j].sci == px[j].sci;
                const double d = __builtin_fma(z.y, inv_lambda, px[j].var);
                const double r = px[j].sci - z.x;
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
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (t == 0) partial[(size_t)w * ny + iy] = acc;
}

}  // namespace psfmc
