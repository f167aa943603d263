// psfmc_rows3_path.h -- row kernels of the long power-of-two rows (nx = 512) on the wave-wide three-stage
// engine (psfmc_fft.h fft_wave3): ONE row per wave at a time, R1 = nx / 64 complex registers per lane.
//
// Why.  The two-stage row kernels hold a whole layout group in registers -- 4 rows x 512 points = 32 complex
// per lane, 185...255 VGPRs, and 16.9 KB of LDS per wave for the exchange -- so a SIMD holds two of them and
// nothing else: the VALU-bound forward rows, the memory-bound columns and the inverse rows of the two passes in
// flight cannot share SIMDs, and each kernel alone is phase-locked (all waves load, then all compute).  Here a
// wave needs ~100 registers and 4.6 KB: four waves per SIMD.
//
// The price is the transposition a layout group needs: the intermediate T keeps [walker][kx][row group][c][r]
// with 4 rows per 128-byte line, and a wave now owns ONE of those rows.  The four waves of a workgroup own the
// four rows of a group; they exchange through a tile in LDS ([kx][c][r], padded) so that global memory still
// sees whole lines: forward, every wave writes its row's two spectra into the tile and the workgroup stores
// the lines; inverse, the workgroup loads the lines into the tile and every wave picks its row.
// The tile aliases the waves' transform exchange regions (a workgroup barrier on either side of its use) and
// is filled in two halves of kx.
//
// Same arithmetic as k_rows_fwd / k_rows_inv (psfmc_fused_path.h): rasteriser, z = raw + i mu raw^2, untangling
// of the two Hermitian spectra (the mirror values come from lane (64 - t) % 64 by a wave shuffle instead of
// LDS), chi^2 with the log terms taken as one per lane.  The chi^2 partial sums are per ROW (ny per walker).
// Reference: psfMC/models.py:213-216, 233-236; utils.py:25-32.
#pragma once
#include "psfmc_fused_path.h"

namespace psfmc {

constexpr int kRows3Waves = 4;                         // = rows of a layout group
constexpr int kRows3Threads = 64 * kRows3Waves;
constexpr int kTileStride = 9;                         // complex per kx in the tile: 2 components x 4 rows + 1 of padding

template <int NX> constexpr int rows3_half_kx() { return NX / 4; }            // kx columns per tile half
// LDS doubles: max(the waves' exchange regions, one tile half (+ the Nyquist column)) + the rasteriser's table
template <int NX> constexpr size_t rows3_lds_doubles(bool with_table) {
    const size_t ex = (size_t)kRows3Waves * fft3_lds_doubles<NX>();
    const size_t tile = (size_t)(rows3_half_kx<NX>() + 1) * kTileStride * 2;
    return (ex > tile ? ex : tile) + (with_table ? kLogTabBytes / sizeof(double) : 0);
}

__device__ __forceinline__ cd shfl_cd(cd v, int src_lane) {
    return cd{__shfl(v.x, src_lane, 64), __shfl(v.y, src_lane, 64)};
}

// ---------------------------------------------------------------------------
// rows3_fwd.  grid (ny / 4, n_walkers), 4 waves per workgroup, wave = row of the layout group.
// ---------------------------------------------------------------------------
template <int NX, bool WRAP = false>
__global__ void __launch_bounds__(kRows3Threads, 4)
k_rows3_fwd(const double* __restrict__ prep, const uint8_t* __restrict__ skip, const cd* __restrict__ twx,
            cd* __restrict__ Tbuf, int n_ps, int n_sersic, int ny, int ps_only, double* __restrict__ raw_out,
            WrapDesc wr) {
    constexpr int R1 = Fft3Shape<NX>::R1, NXH = NX / 2 + 1, HK = rows3_half_kx<NX>();
    static_assert(R1 == 8, "built for nx = 512");
    extern __shared__ __align__(16) double smem[];
    const int w = blockIdx.y;
    if (skip && skip[w]) return;                                     // (workgroup-uniform)
    const int t = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int yg = blockIdx.x, iy = yg * 4 + wv;
    double* ex = smem + (size_t)wv * fft3_lds_doubles<NX>();          // this wave's transform exchange region
    cd* tile = reinterpret_cast<cd*>(smem);                           // aliases all four (barriers around its use)
    double* log_tab = smem + rows3_lds_doubles<NX>(false);            // shared by the workgroup
    const double* wprep = prep + (size_t)w * prep_len(n_ps, n_sersic);   // wave-uniform
    const double mu = wprep[kPrepMu];
    if (!ps_only) {
        if (threadIdx.x < 64) load_log_table(log_tab, t);
        __syncthreads();
    }
    double r[R1];
    raster_row<R1, 64, 0, WRAP>(wprep, n_ps, n_sersic, t, iy, ps_only != 0, log_tab, r, wr);
    cd v[R1];
#pragma unroll
    for (int k = 0; k < R1; ++k) v[k] = cd{r[k], mu * r[k] * r[k]};
    if (raw_out) {
        double* o = raw_out + (size_t)w * ny * NX + (size_t)iy * NX;
#pragma unroll
        for (int k = 0; k < R1; ++k) o[64 * k + t] = v[k].x;
    }
    cd w1[fft3_w1_regs<NX>()], w2[8];
    load_twiddles3<NX>(w1, w2, twx, t);
    fft_wave3<NX, -1>(v, w1, w2, twx, t, ex);                         // v[e] = Z[t + 64 e]
    // Untangle: Z[NX - k] for k = t + 64 e sits in lane (64 - t) % 64, register R1 - 1 - e (t != 0) or R1 - e
    // (t == 0, e >= 1); k = 0 and k = NX / 2 are their own mirrors.  TWICE the spectra of raw and mu raw^2
    // (the 1/2 rides on the kernel spectra, like k_rows_fwd).
    const int src = (64 - t) & 63;
    cd A[R1 / 2], B[R1 / 2];
#pragma unroll
    for (int e = 0; e < R1 / 2; ++e) {
        const cd zk = v[e];
        const cd from_other = shfl_cd(v[R1 - 1 - e], src);
        cd zm = from_other;
        if (t == 0) zm = e == 0 ? zk : v[R1 - e];
        A[e] = cd{zk.x + zm.x, zk.y - zm.y};
        B[e] = cd{zk.y + zm.y, zm.x - zk.x};
    }
    const cd zn = v[R1 / 2];                                          // lane 0: the Nyquist column
    const int nyp = ny;                                               // (ny a multiple of 4)
    cd* wbase = Tbuf + (size_t)w * 2 * NXH * nyp + (size_t)yg * 8;   // + kx * 2 nyp: the line of (kx, yg)
    __syncthreads();                                                  // every wave is done with its exchange region
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        // this half's kx = t + 64 e, e = 2 half .. 2 half + 1, tile slot kx - half HK; the Nyquist column rides in
        // the second half's spare slot HK
#pragma unroll
        for (int j = 0; j < R1 / 4; ++j) {
            const int e = half * (R1 / 4) + j, slot = t + 64 * j;
            tile[slot * kTileStride + wv] = A[e];
            tile[slot * kTileStride + 4 + wv] = B[e];
        }
        if (half == 1 && t == 0) {
            tile[HK * kTileStride + wv] = cd{zn.x + zn.x, 0.0};
            tile[HK * kTileStride + 4 + wv] = cd{zn.y + zn.y, 0.0};
        }
        __syncthreads();
        // the workgroup stores whole 128-byte lines: element i of the half = (slot, 0..7)
        const int n_el = (HK + (half == 1 ? 1 : 0)) * 8;
        for (int i = threadIdx.x; i < n_el; i += kRows3Threads) {
            const int slot = i >> 3, q = i & 7;
            const int kx = half * HK + slot;
            wbase[(size_t)kx * 2 * nyp + q] = tile[slot * kTileStride + q];
        }
        __syncthreads();
    }
}

}  // namespace psfmc
