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

#ifndef PSFMC_PAIR_ROLE_ATTR
#define PSFMC_PAIR_ROLE_ATTR __forceinline__
#endif
#ifndef PSFMC_PAIR_ROW_PRIO
#define PSFMC_PAIR_ROW_PRIO 0
#endif
#ifndef PSFMC_PAIR_COL_PRIO
#define PSFMC_PAIR_COL_PRIO 3        /* the memory-bound partner goes first, as in the separate kernels */
#endif

// Arguments of a non-kernel function arrive in VECTOR registers and count as divergent: without these
// the roles' wave-uniform pointers (walker records read with scalar loads into scalar registers,
// saddr-form global addressing) would all turn into per-lane values.
template <typename Tp> __device__ __forceinline__ Tp* uniform(Tp* p) {
    const unsigned long long u = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return reinterpret_cast<Tp*>(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ unsigned uniform(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }

// A value the compiler must take as new at this point: inside the roles' item loops it keeps the loop-
// invariant parts of a body (31 inter-stage twiddle loads per transform at P = 32, per-lane address
// offsets, ...) from being hoisted out of the loop, where they would hold ~130 registers for the whole
// launch and push the body itself into scratch.
template <typename Tp> __device__ __forceinline__ Tp* opaque(Tp* p) {
    asm volatile("" : "+s"(p));
    return p;
}
__device__ __forceinline__ int opaque_lane(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

// The roles are separate, NOT inlined functions: each gets a register allocation of its own (inlined
// into one kernel body the allocator carried one role's wave-uniform pointers and constants through the
// other role's code and spilled ~200 registers per lane; the kernel's budget is the larger role's).
template <int N, bool MULTI>
__device__ PSFMC_PAIR_ROLE_ATTR void pair_row_role(const PairRows rows_, const cd* __restrict__ tw_,
                                                        const FieldPx* __restrict__ field_, int n_ps_, int n_sersic_,
                                                        int plen_, int n_psf_field_, unsigned field_stride_,
                                                        double* __restrict__ wave_lds_, int wave_, int lane) {
    PairRows rows;
    rows.T = uniform(rows_.T);
    rows.n_inv = uniform(rows_.n_inv);
    rows.n_fwd = uniform(rows_.n_fwd);
    rows.prep_inv = uniform(rows_.prep_inv);
    rows.prep_fwd = uniform(rows_.prep_fwd);
    rows.skip_inv = uniform(rows_.skip_inv);
    rows.skip_fwd = uniform(rows_.skip_fwd);
    rows.partial = uniform(rows_.partial);
    const cd* __restrict__ tw = uniform(tw_);
    const FieldPx* __restrict__ field = uniform(field_);
    const int n_ps = uniform(n_ps_), n_sersic = uniform(n_sersic_), plen = uniform(plen_);
    const int n_psf_field = uniform(n_psf_field_), wave = uniform(wave_);
    const unsigned field_stride = uniform(field_stride_);
    double* __restrict__ wave_lds = uniform(wave_lds_);
    constexpr int RG = row_group<N>();
    constexpr int NYG = N / RG;                         // row groups (waves' worth of rows) per walker
    constexpr int NG = NYG / kPairRowWaves;             // items per walker slot: 4 consecutive row groups
    static_assert(NYG % kPairRowWaves == 0, "rows");
    // Two loops over the same items, not one loop doing both: every wave first takes all its leaving
    // walkers' rows out, then moves the entering walkers' rows in -- in place, the same wave reading
    // and later writing the same addresses, so program order is all the ordering there is to keep.
    // (One loop with both bodies made the register allocator carry each body's hoisted invariants
    // through the other: ~340 scratch accesses per item.)
#ifndef PSFMC_PAIR_ROWS
#define PSFMC_PAIR_ROWS 3
#endif
#if PSFMC_PAIR_ROWS & 1
    for (int item = blockIdx.x; item < rows.n_inv * NG; item += gridDim.x) {
        const int slot = item / NG, yg = (item - slot * NG) * kPairRowWaves + wave;
        rows_inv_wave<N, cd, true, MULTI>(slot, yg, NYG, opaque_lane(lane), opaque(wave_lds), opaque(rows.T),
                                          rows.skip_inv, opaque(tw), opaque(field), rows.partial, N,
                                          opaque(rows.prep_inv), plen, nullptr, nullptr, n_psf_field, field_stride);
        wave_lds_sync();
    }
#endif
#if PSFMC_PAIR_ROWS & 2
    for (int item = blockIdx.x; item < rows.n_fwd * NG; item += gridDim.x) {
        const int slot = item / NG, yg = (item - slot * NG) * kPairRowWaves + wave;
        rows_fwd_wave<N, false, cd, true>(slot, yg, opaque_lane(lane), opaque(wave_lds), opaque(rows.prep_fwd),
                                          rows.skip_fwd, opaque(tw), opaque(rows.T), n_ps, n_sersic, N, 0, nullptr,
                                          nullptr, nullptr);
        wave_lds_sync();
    }
#endif
}

template <int N>
__device__ PSFMC_PAIR_ROLE_ATTR void pair_col_role(const PairCols cols, const cd* __restrict__ tw,
                                                        const cd* __restrict__ Kt, int plen,
                                                        double* __restrict__ lds, const cd* __restrict__ w1s, int cw,
                                                        int lane) {
    cols3_wave<N, true, cd>(uniform(cw), lane, uniform(lds), uniform(w1s), uniform(cols.T), uniform(Kt),
                            uniform(cols.prep), uniform(cols.skip), uniform(tw), uniform(plen), N / 2 + 1,
                            uniform(cols.n), __builtin_ctz(row_group<N>()));
}

template <int N, bool MULTI>
__global__ void __launch_bounds__(kPairThreads, 2)
k_pair(PairRows rows, PairCols cols, const cd* __restrict__ tw, const cd* __restrict__ Kt,
       const FieldPx* __restrict__ field, int n_ps, int n_sersic, int plen, int n_psf_field, unsigned field_stride) {
    static_assert(FftShape<N>::kPlain && (N == 512 || N == 1024), "paired pipeline: 512^2 and 1024^2");
    constexpr int R1 = Fft3Shape<N>::R1;
    extern __shared__ __align__(16) double smem[];
    // the wave number decides the role: readfirstlane makes it a scalar, so the roles part with a
    // scalar branch (as a per-lane value the two role bodies became one divergent region)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
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
#ifndef PSFMC_PAIR_ROLES
#define PSFMC_PAIR_ROLES 3
#endif
    if (wave < kPairRowWaves) {
#if !(PSFMC_PAIR_ROLES & 1)
        return;
#endif
#if PSFMC_PAIR_ROW_PRIO
        __builtin_amdgcn_s_setprio(PSFMC_PAIR_ROW_PRIO);
#endif
        pair_row_role<N, MULTI>(rows, tw, field, n_ps, n_sersic, plen, n_psf_field, field_stride,
                                smem + (size_t)wave * fused_row_wave_lds_doubles<N>(), wave, lane);
    } else if (cols.n > 0) {
#if !(PSFMC_PAIR_ROLES & 2)
        return;
#endif
#if PSFMC_PAIR_COL_PRIO
        __builtin_amdgcn_s_setprio(PSFMC_PAIR_COL_PRIO);
#endif
        const int cw = wave - kPairRowWaves;
        pair_col_role<N>(cols, tw, Kt, plen, col_lds + (size_t)cw * fft3_lds_doubles<N>(), w1s, cw, lane);
    }
}

}  // namespace psfmc
