// psfmc_fused_path.h -- the PSFMC_BACKEND_FUSED kernels: three launches per
// batch of walkers, no stand-alone FFT passes.
//
//   rows_fwd   rasterise a tile of image rows straight into registers as
//              z = raw + i raw^2 (two real images in one complex signal), FFT
//              along x, untangle the two Hermitian spectra and store them
//              transposed:   T[w][c][kx][y]   c = 0 (raw), 1 (raw^2), kx <= nx/2
//   cols       per (w, c, kx) column, contiguous in y: FFT along y, multiply by
//              the pre-scaled, pre-shifted kernel spectrum Kt[psf][c][kx][ky],
//              inverse FFT along y, store in place
//   rows_inv   reload a tile of rows (all kx), rebuild the full complex spectrum
//              Y = G + i H (G, H Hermitian in kx), inverse FFT along x: real part
//              = PSF-convolved model, imaginary part = model variance; fused
//              chi^2 + log term + masked reduction -> one partial per tile.
//
// HBM traffic per walker: one write + one read/write + one read of T
// (2*(nx/2+1)*ny complex128), i.e. 4 x 1.03 MB at 256^2, against 6.3 MB of
// "algorithmic" bytes for the unfused arrangement.
//
// Reference: psfMC/models.py:213-216, 233-236; utils.py:25-32 (convolve: the
// ifftshift is the (-1)^(kx+ky) sign folded into Kt, as is 1/(nx*ny)).
#pragma once
#include "psfmc_device.h"
#include "psfmc_fft.h"

namespace psfmc {

constexpr int kFusedThreads = 256;

template <int N> constexpr int fused_ffts_per_block() { return kFusedThreads / FftShape<N>::T; }
// waves per SIMD the register allocator must leave room for: 32-point lanes
// (N >= 512) need the whole register file
template <int N> constexpr int fused_min_waves() { return FftShape<N>::P > 16 ? 1 : 2; }

// LDS bytes of a row kernel working on transforms of length NX
template <int NX> constexpr size_t fused_row_lds_bytes() {
    constexpr size_t fpb = fused_ffts_per_block<NX>();
    constexpr size_t xch = fpb * fft_lds_elems<NX>();
    constexpr size_t tile = fpb * (NX + 1);                 // Z tile (rows_fwd)
    constexpr size_t gh = 2 * fpb * (NX / 2 + 1);            // G,H tile (rows_inv)
    constexpr size_t m = xch > tile ? (xch > gh ? xch : gh) : (tile > gh ? tile : gh);
    return m * sizeof(cd);
}
template <int NY> constexpr size_t fused_col_lds_bytes() {
    return (size_t)fused_ffts_per_block<NY>() * fft_lds_elems<NY>() * sizeof(cd);
}

// ---------------------------------------------------------------------------
// rows_fwd.  grid (ny / FPB, n_walkers); block 256.
//   FROM_IMAGE = false: rasterise from prep (the hot path)
//   FROM_IMAGE = true : z = img0 + i img_scale[w] img1 read from memory (PSF spectra at setup)
// raw_out (optional): [n][ny][nx] copy of the raw model (psfmc_eval_images)
// ---------------------------------------------------------------------------
template <int NX, bool FROM_IMAGE>
__global__ void __launch_bounds__(kFusedThreads, fused_min_waves<NX>())
k_rows_fwd(const double* __restrict__ prep, const uint8_t* __restrict__ skip,
           const cd* __restrict__ twx, cd* __restrict__ Tbuf, int n_ps, int n_sersic, int ny,
           int ps_only, const double* __restrict__ img, const double* __restrict__ img_scale,
           double* __restrict__ raw_out) {
    constexpr int P = FftShape<NX>::P, T = FftShape<NX>::T;
    constexpr int FPB = fused_ffts_per_block<NX>();
    constexpr int NXH = NX / 2 + 1;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    cd* smem = reinterpret_cast<cd*>(smem_raw);
    __shared__ double s_prep[kPrepHead + kPrepPs * 16 + kPrepSersic * 16];

    const int w = blockIdx.y;
    if (skip && skip[w]) return;
    const int f = threadIdx.x / T, t = threadIdx.x % T;
    const int y0 = blockIdx.x * FPB;
    const int iy = y0 + f;
    const size_t S = (size_t)ny * NX;

    cd v[P];
    if constexpr (FROM_IMAGE) {
        const double* a = img + (size_t)(2 * w) * S + (size_t)iy * NX;
        const double* b = a + S;
#pragma unroll
        for (int k = 0; k < P; ++k) v[k] = cd{a[T * k + t], b[T * k + t] * img_scale[w]};
    } else {
        const int plen = prep_len(n_ps, n_sersic);
        for (int i = threadIdx.x; i < plen; i += kFusedThreads) s_prep[i] = prep[(size_t)w * plen + i];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < P; ++k) {
            const double r = raster_pixel(s_prep, n_ps, n_sersic, T * k + t, iy, ps_only != 0);
            v[k] = cd{r, s_prep[kPrepMu] * r * r};
        }
        if (raw_out) {
            double* o = raw_out + (size_t)w * S + (size_t)iy * NX;
#pragma unroll
            for (int k = 0; k < P; ++k) o[T * k + t] = v[k].x;
        }
    }
    cd tw[fft_tw_regs<NX>()];
    load_twiddles<NX>(tw, twx, t);
    fft_coop<NX, -1>(v, tw, twx, t, smem + (size_t)f * fft_lds_elems<NX>());

    // Z tile in LDS: Zl[f][k], row stride NX+1 (conflict-free column reads)
    cd* Zl = smem;
#pragma unroll
    for (int e = 0; e < P; ++e) Zl[f * (NX + 1) + t + T * e] = v[e];
    __syncthreads();
    // untangle + transposed store: lanes run over the tile's rows (contiguous y)
    const int fr = threadIdx.x % FPB, g = threadIdx.x / FPB;
    cd* T0 = Tbuf + (size_t)w * 2 * NXH * ny + y0 + fr;
    cd* T1 = T0 + (size_t)NXH * ny;
    for (int kx = g; kx < NXH; kx += kFusedThreads / FPB) {
        const cd zk = Zl[fr * (NX + 1) + kx];
        const cd zm = Zl[fr * (NX + 1) + ((NX - kx) & (NX - 1))];
        T0[(size_t)kx * ny] = cd{0.5 * (zk.x + zm.x), 0.5 * (zk.y - zm.y)};
        T1[(size_t)kx * ny] = cd{0.5 * (zk.y + zm.y), -0.5 * (zk.x - zm.x)};
    }
}

// ---------------------------------------------------------------------------
// cols.  persistent grid; one column = ny contiguous complex.
//   CONVOLVE = true : FFT_y, * Kt[psf][c][kx][.], IFFT_y (the hot path)
//   CONVOLVE = false: FFT_y only (PSF spectra at setup)
// ---------------------------------------------------------------------------
template <int NY, bool CONVOLVE>
__global__ void __launch_bounds__(kFusedThreads, fused_min_waves<NY>())
k_cols(cd* __restrict__ Tbuf, const cd* __restrict__ Kt, const double* __restrict__ prep,
       const uint8_t* __restrict__ skip, const cd* __restrict__ twy, int plen, int nxh, int n_cols) {
    constexpr int P = FftShape<NY>::P, T = FftShape<NY>::T;
    constexpr int FPB = fused_ffts_per_block<NY>();
    extern __shared__ __align__(16) unsigned char smem_raw[];
    cd* smem = reinterpret_cast<cd*>(smem_raw);
    const int s = threadIdx.x / T, t = threadIdx.x % T;
    cd* xbuf = smem + (size_t)s * fft_lds_elems<NY>();
    cd tw[fft_tw_regs<NY>()];
    load_twiddles<NY>(tw, twy, t);
    const int n_groups = (n_cols + FPB - 1) / FPB;
    for (int grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        const int col = grp * FPB + s;
        bool active = col < n_cols;
        int w = 0, q = 0;
        if (active) {
            w = col / (2 * nxh);
            q = col - w * 2 * nxh;
            if (skip && skip[w]) active = false;
        }
        cd* base = Tbuf + (size_t)col * NY;
        cd v[P];
#pragma unroll
        for (int a = 0; a < P; ++a) v[a] = active ? base[T * a + t] : cd{0.0, 0.0};
        fft_coop<NY, -1>(v, tw, twy, t, xbuf);
        if constexpr (CONVOLVE) {
            const int psf = active ? (int)prep[(size_t)w * plen + kPrepPsfIdx] : 0;
            const cd* k = Kt + ((size_t)psf * 2 * nxh + q) * NY;
#pragma unroll
            for (int e = 0; e < P; ++e) v[e] = cmul(v[e], k[t + T * e]);
            fft_coop<NY, +1>(v, tw, twy, t, xbuf);
        }
        if (active) {
#pragma unroll
            for (int e = 0; e < P; ++e) base[T * e + t] = v[e];
        }
    }
}

// ---------------------------------------------------------------------------
// rows_inv.  grid (ny / FPB, n_walkers); block 256.
// partial[w][blockIdx.x] = sum over the tile's good pixels of the chi^2 term.
// conv_out / var_out (optional): [n][ny][nx] images (psfmc_eval_images)
// ---------------------------------------------------------------------------
template <int NX>
__global__ void __launch_bounds__(kFusedThreads, fused_min_waves<NX>())
k_rows_inv(const cd* __restrict__ Tbuf, const uint8_t* __restrict__ skip, const cd* __restrict__ twx,
           const double* __restrict__ sci, const double* __restrict__ obs_var,
           const uint8_t* __restrict__ bad, double* __restrict__ partial, int ny,
           const double* __restrict__ prep, int plen,
           double* __restrict__ conv_out, double* __restrict__ var_out) {
    constexpr int P = FftShape<NX>::P, T = FftShape<NX>::T;
    constexpr int FPB = fused_ffts_per_block<NX>();
    constexpr int NXH = NX / 2 + 1;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    cd* smem = reinterpret_cast<cd*>(smem_raw);
    __shared__ double s_red[kFusedThreads / 64];

    const int w = blockIdx.y;
    if (skip && skip[w]) return;
    const int y0 = blockIdx.x * FPB;
    // transposed load: lanes run over the tile's rows
    {
        const int fr = threadIdx.x % FPB, g = threadIdx.x / FPB;
        const cd* src = Tbuf + (size_t)w * 2 * NXH * ny + y0 + fr;
        for (int q = g; q < 2 * NXH; q += kFusedThreads / FPB) {
            const int c = q >= NXH ? 1 : 0;
            const int kx = q - c * NXH;
            smem[((size_t)c * FPB + fr) * NXH + kx] = src[(size_t)q * ny];
        }
    }
    __syncthreads();
    const int f = threadIdx.x / T, t = threadIdx.x % T;
    cd v[P];
    {
        const cd* G = smem + (size_t)f * NXH;
        const cd* H = smem + ((size_t)FPB + f) * NXH;
#pragma unroll
        for (int a = 0; a < P; ++a) {
            const int k = T * a + t;
            if (k <= NX / 2) {
                const cd gk = G[k], hk = H[k];
                v[a] = cd{gk.x - hk.y, gk.y + hk.x};
            } else {
                const cd gk = G[NX - k], hk = H[NX - k];
                v[a] = cd{gk.x + hk.y, hk.x - gk.y};
            }
        }
    }
    cd tw[fft_tw_regs<NX>()];
    load_twiddles<NX>(tw, twx, t);
    __syncthreads();                       // G/H tile is dead; its LDS becomes the exchange buffer
    fft_coop<NX, +1>(v, tw, twx, t, smem + (size_t)f * fft_lds_elems<NX>());
    // imaginary part is lambda * model variance (see build_prep)
    const double inv_lambda = prep[(size_t)w * plen + kPrepInvLambda];
#pragma unroll
    for (int e = 0; e < P; ++e) v[e].y *= inv_lambda;

    const int iy = y0 + f;
    const size_t rowoff = (size_t)iy * NX;
    if (conv_out) {
        double* oc = conv_out + (size_t)w * ny * NX + rowoff;
        double* ov = var_out + (size_t)w * ny * NX + rowoff;
#pragma unroll
        for (int e = 0; e < P; ++e) {
            oc[T * e + t] = v[e].x;
            ov[T * e + t] = v[e].y;
        }
    }
    double acc = 0.0;
#pragma unroll
    for (int e = 0; e < P; ++e) {
        const size_t i = rowoff + T * e + t;
        if (!bad[i]) acc += chi2_term(sci[i], obs_var[i], v[e].x, v[e].y);
    }
    const double tot = block_sum(acc, s_red);
    if (threadIdx.x == 0) partial[(size_t)w * gridDim.x + blockIdx.x] = tot;
}

// Kt[psf][c][kx][ky] = spec_c[psf][ky][kx] * (-1)^(kx+ky) / S, from the raw
// column-transformed PSF buffer (same [c][kx][ky] layout; its c = 1 half already
// carries the channel scale rho[psf], which stays in Kt)
__global__ void k_scale_kernel_spectrum(const cd* __restrict__ raw, cd* __restrict__ Kt, int n_total,
                                        int ny, int nxh, double inv_s) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_total; i += gridDim.x * blockDim.x) {
        const int ky = i % ny;
        const int kx = (i / ny) % nxh;
        const double sc = ((kx + ky) & 1) ? -inv_s : inv_s;
        Kt[i] = cd{raw[i].x * sc, raw[i].y * sc};
    }
}

// natural-layout copy for psfmc_get_spectra: out[psf][ky][kx] of component c
__global__ void k_untranspose_spectrum(const cd* __restrict__ raw, cd* __restrict__ out, int n_psf,
                                       int c, int ny, int nxh, const double* __restrict__ rho) {
    const int n = n_psf * ny * nxh;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int kx = i % nxh;
        const int ky = (i / nxh) % ny;
        const int p = i / (nxh * ny);
        const double sc = c ? 1.0 / rho[p] : 1.0;
        const cd v = raw[(((size_t)p * 2 + c) * nxh + kx) * ny + ky];
        out[i] = cd{v.x * sc, v.y * sc};
    }
}

}  // namespace psfmc
