// psfmc_pair_path.h -- the PAIRED pipeline of the large power-of-two squares (512^2, 1024^2).
//
// At these sizes a pass of the three-kernel flow (psfmc_fused_path.h) carries as much arithmetic as
// memory traffic: with two and four Sersic components the forward row kernel is VALU-bound (the
// rasteriser runs at the issue rate), the column kernel memory-bound, and run as separate launches
// on two streams they overlap badly -- a SIMD holds two 185-register row waves OR one of them and a
// 240-register column wave, whichever the dispatcher happens to place, and a launch is only 1.5
// rounds of row waves (round 2: 1024^2 ran at 1.9x, 512^2 at 1.3x their no-arithmetic sweep floor).
//
// Here the overlap is by construction.  ONE launch per pass, one 512-thread workgroup per CU:
//   waves 0..3  ROW role    for their row groups of buffer B: the inverse row transform + chi^2 of the
//                           walker that occupies the slot (pass k - 2), then -- in place, same wave,
//                           same addresses -- the rasteriser + forward row transform of the walker
//                           that takes the slot over (pass k)
//   waves 4..7  COLUMN role the column pass (FFT_y, x kernel spectrum, IFFT_y, in place) of pass k - 1
//                           in buffer A
// so every SIMD holds one VALU-bound wave and one memory-bound wave for the whole launch.  The two
// roles never touch the same buffer inside a launch: every dependency (rows -> columns -> rows)
// crosses a kernel boundary on ONE stream, so there are no flags, no agent-scope fences, no events.
// Two buffers (A and B swap every launch), both resident in the Infinity Cache as before.
//
// A pass of n walkers therefore moves through three consecutive launches:
//   launch k      row role:    rasterise + FFT_x                -> buffer k % 2
//   launch k + 1  column role: in place                            buffer k % 2
//   launch k + 2  row role:    IFFT_x + chi^2 partial sums      <- buffer k % 2   (then pass k + 2 moves in)
//
// Reference: psfMC/models.py:213-216, 233-236 (the same arithmetic as the three kernels: this file
// only schedules their per-wave bodies).
#pragma once
#include "psfmc_fused_path.h"

namespace psfmc {

constexpr int kPairRowWaves = 4, kPairColWaves = kColThreads / 64;
constexpr int kPairThreads = 64 * (kPairRowWaves + kPairColWaves);

// LDS: the row waves' regions, the column waves' exchange regions, the shared stage-1 twiddle table
template <int N> constexpr size_t pair_lds_doubles() {
    return (size_t)kPairRowWaves * fused_row_wave_lds_doubles<N>() + (size_t)kPairColWaves * fft3_lds_doubles<N>() +
           (Fft3Shape<N>::R1 > 8 ? (size_t)Fft3Shape<N>::R1 * 64 * 2 : 0);
}
template <int N> constexpr size_t pair_lds_bytes() { return pair_lds_doubles<N>() * sizeof(double); }

struct PairRows {                    // the row role's work: slots [0, n) of buffer T
    cd* T;
    int n_inv, n_fwd;                // walkers leaving (inverse + chi^2) / entering (rasterise + forward)
    const double *prep_inv, *prep_fwd;
    const uint8_t *skip_inv, *skip_fwd;
    double* partial;                 // [n_inv][nyg]
};
struct PairCols {                    // the column role's work: the n walkers of buffer T
    cd* T;
    int n;
    const double* prep;
    const uint8_t* skip;
};

#ifndef PSFMC_PAIR_ROW_PRIO
#define PSFMC_PAIR_ROW_PRIO 0
#endif
#ifndef PSFMC_PAIR_COL_PRIO
#define PSFMC_PAIR_COL_PRIO 3        /* the memory-bound partner goes first, as in the separate kernels */
#endif

template <int N, bool MULTI>
__global__ void __launch_bounds__(kPairThreads, 2)
k_pair(PairRows rows, PairCols cols, const cd* __restrict__ tw, const cd* __restrict__ Kt,
       const FieldPx* __restrict__ field, int n_ps, int n_sersic, int plen, int n_psf_field, unsigned field_stride) {
    static_assert(FftShape<N>::kPlain && (N == 512 || N == 1024), "paired pipeline: 512^2 and 1024^2");
    constexpr int R1 = Fft3Shape<N>::R1;
    constexpr int RG = row_group<N>();
    constexpr int NYG = N / RG;                         // row groups (waves' worth of rows) per walker
    constexpr int NG = NYG / kPairRowWaves;             // items per walker slot: 4 consecutive row groups
    static_assert(NYG % kPairRowWaves == 0, "rows");
    extern __shared__ __align__(16) double smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double* col_lds = smem + (size_t)kPairRowWaves * fused_row_wave_lds_doubles<N>();
    const cd* w1s = nullptr;
    if constexpr (R1 > 8) {
        cd* tab = reinterpret_cast<cd*>(col_lds + (size_t)kPairColWaves * fft3_lds_doubles<N>());
        if (cols.n > 0) {
            for (int i = threadIdx.x; i < R1 * 64; i += kPairThreads) tab[i] = tw[(i & 63) * (i >> 6)];
        }
        __syncthreads();                                   // once, before the roles part
        w1s = tab;
    }
    if (wave < kPairRowWaves) {
#if PSFMC_PAIR_ROW_PRIO
        __builtin_amdgcn_s_setprio(PSFMC_PAIR_ROW_PRIO);
#endif
        double* wave_lds = smem + (size_t)wave * fused_row_wave_lds_doubles<N>();
        const int n_slots = rows.n_inv > rows.n_fwd ? rows.n_inv : rows.n_fwd;
        const int n_items = n_slots * NG;
        for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
            const int slot = item / NG, yg = (item - slot * NG) * kPairRowWaves + wave;
            if (slot < rows.n_inv)
                rows_inv_wave<N, cd, true, MULTI>(slot, yg, NYG, lane, wave_lds, rows.T, rows.skip_inv, tw, field,
                                                  rows.partial, N, rows.prep_inv, plen, nullptr, nullptr, n_psf_field,
                                                  field_stride);
            if (slot < rows.n_fwd) {
                // the transform of the walker that leaves has used the wave's LDS region; the rasteriser's
                // table goes into the same region next
                wave_lds_sync();
                rows_fwd_wave<N, false, cd, true>(slot, yg, lane, wave_lds, rows.prep_fwd, rows.skip_fwd, tw, rows.T,
                                                  n_ps, n_sersic, N, 0, nullptr, nullptr, nullptr);
                wave_lds_sync();
            }
        }
    } else if (cols.n > 0) {
#if PSFMC_PAIR_COL_PRIO
        __builtin_amdgcn_s_setprio(PSFMC_PAIR_COL_PRIO);
#endif
        const int cw = wave - kPairRowWaves;
        cols3_wave<N, true, cd>(cw, lane, col_lds + (size_t)cw * fft3_lds_doubles<N>(), w1s, cols.T, Kt, cols.prep,
                                cols.skip, tw, plen, N / 2 + 1, cols.n, __builtin_ctz(RG));
    }
}

}  // namespace psfmc
