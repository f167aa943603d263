// psfmc_hipfft_path.h -- kernels of the PSFMC_BACKEND_HIPFFT path: separate
// rasteriser / spectral multiply / chi^2 kernels around batched hipFFT D2Z/Z2D.
// This is the straightforward arrangement (every intermediate crosses HBM); it
// is kept as the on-device cross-check and the baseline the fused path is
// measured against.  It also serves psfmc_eval_images().
#pragma once
#include "psfmc_device.h"

namespace psfmc {

// rows [W][row_len] -> prep [W][prep_len]
__global__ void k_prep(const double* __restrict__ rows, double* __restrict__ prep, int W,
                       int n_ps, int n_sersic, int ny, int nx, const double* __restrict__ rho, int n_psf,
                       int psf_base) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= W) return;
    build_prep(rows + (size_t)w * row_len(n_ps, n_sersic),
               prep + (size_t)w * prep_len(n_ps, n_sersic), n_ps, n_sersic, ny, nx, rho, n_psf, psf_base);
}

// raw model and its square: real[(2w)][S] = raw, real[(2w+1)][S] = raw^2
// (models.py:213, :277).  grid (ceil(S / (256*4)), W)
__global__ void __launch_bounds__(256)
k_raster(const double* __restrict__ prep, const uint8_t* __restrict__ skip,
         double* __restrict__ real, int n_ps, int n_sersic, int ny, int nx, int ps_only) {
    const int w = blockIdx.y;
    if (skip && skip[w]) return;
    extern __shared__ double s_prep[];
    const int plen = prep_len(n_ps, n_sersic), rec = prep_rec_len(n_ps, n_sersic);   // the power tables stay behind
    for (int i = threadIdx.x; i < rec; i += blockDim.x) s_prep[i] = prep[(size_t)w * plen + i];
    __syncthreads();
    const int S = ny * nx;
    double* raw = real + (size_t)(2 * w) * S;
    double* raw2 = raw + S;
    const int base = blockIdx.x * (256 * 4) + threadIdx.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int idx = base + j * 256;
        if (idx < S) {
            const int iy = idx / nx, ix = idx - iy * nx;
            const double v = raster_pixel(s_prep, n_ps, n_sersic, ix, iy, ps_only != 0);
            raw[idx] = v;
            raw2[idx] = v * v;
        }
    }
}

// spectrum of image (2w+c) *= K_c[psf(w)] * (-1)^(ky+kx) / S
// (utils.py:32: the ifftshift of an even-sized image is that sign in Fourier
// space; 1/S is the inverse-transform normalisation hipFFT leaves out)
__global__ void __launch_bounds__(256)
k_spec_mul(double2* __restrict__ spec, const double2* __restrict__ pspec,
           const double2* __restrict__ vspec, const double* __restrict__ prep,
           const uint8_t* __restrict__ skip, int plen, int ny, int nxh, double inv_s) {
    const int w = blockIdx.y;
    if (skip && skip[w]) return;
    const int F = ny * nxh;
    const int psf = (int)prep[(size_t)w * plen + kPrepPsfIdx];
    const double2* kp = pspec + (size_t)psf * F;
    const double2* kv = vspec + (size_t)psf * F;
    double2* a = spec + (size_t)(2 * w) * F;
    double2* b = a + F;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < F; i += gridDim.x * blockDim.x) {
        const int ky = i / nxh, kx = i - ky * nxh;
        const double sc = ((ky + kx) & 1) ? -inv_s : inv_s;
        const double2 p = kp[i], v = kv[i], za = a[i], zb = b[i];
        a[i] = make_double2((za.x * p.x - za.y * p.y) * sc, (za.x * p.y + za.y * p.x) * sc);
        b[i] = make_double2((zb.x * v.x - zb.y * v.y) * sc, (zb.x * v.y + zb.y * v.x) * sc);
    }
}

// masked chi^2 partial sums: partial[w][blockIdx.x]; grid (nblk, W)
__global__ void __launch_bounds__(256)
k_chi2(const double* __restrict__ real, const double* __restrict__ sci,
       const double* __restrict__ obs_var, const uint8_t* __restrict__ bad,
       const uint8_t* __restrict__ skip, double* __restrict__ partial, int S) {
    const int w = blockIdx.y;
    if (skip && skip[w]) return;
    __shared__ double s_red[4];
    const double* conv = real + (size_t)(2 * w) * S;
    const double* mvar = conv + S;
    double acc = 0.0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < S; i += gridDim.x * blockDim.x)
        if (!bad[i]) acc += chi2_term(sci[i], obs_var[i], conv[i], mvar[i]);
    const double tot = block_sum(acc, s_red);
    if (threadIdx.x == 0) partial[(size_t)w * gridDim.x + blockIdx.x] = tot;
}

// loglike[w] = -0.5 * sum(partials) in a fixed order; skipped walkers -> -inf.
// One wave per walker (launch: finish_blocks(W) x kFinishThreads).
__global__ void k_finish(const double* __restrict__ partial, const uint8_t* __restrict__ skip,
                         double* __restrict__ loglike, int W, int nblk) {
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * (kFinishThreads / 64) + (threadIdx.x >> 6);
    if (w >= W) return;
    if (skip && skip[w]) { if (lane == 0) loglike[w] = -INFINITY; return; }
    const double s = wave_sum_partials(partial + (size_t)w * nblk, nblk, lane);
    if (lane == 0) loglike[w] = -0.5 * s;
}

// centre-pad a small image into an [ny][nx] canvas at offset pad/2 (utils.py:19-21)
__global__ void k_pad(const double* __restrict__ src, double* __restrict__ dst, int n_img,
                      int sy, int sx, int ny, int nx) {
    const int S = ny * nx;
    const int oy = (ny - sy) / 2, ox = (nx - sx) / 2;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_img * S; i += gridDim.x * blockDim.x) {
        const int img = i / S, r = i - img * S;
        const int y = r / nx - oy, x = r % nx - ox;
        dst[i] = (y >= 0 && y < sy && x >= 0 && x < sx) ? src[((size_t)img * sy + y) * sx + x] : 0.0;
    }
}

// image outputs: out = real[(2w+c)] or a pointwise function of it
// `real`, `sci`, `obs_var` are [ny][nx] (the transform's shape); the output is the image's own window of it
// (ImgWindow, psfmc_device.h)
enum ImgOp { IMG_COPY = 0, IMG_RESID = 1, IMG_IVM = 2 };
__global__ void k_image_out(const double* __restrict__ real, const double* __restrict__ sci,
                            const double* __restrict__ obs_var, double* __restrict__ out,
                            int S, int stride, int c, int op, ImgWindow win) {
    const int w = blockIdx.y;
    const double* src = real + (size_t)(stride * w + c) * S;
    const int S_out = win.ly * win.lx;
    double* dst = out + (size_t)w * S_out;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < S_out; i += gridDim.x * blockDim.x) {
        const int y = i / win.lx, x = i - y * win.lx;
        const int j = (y + win.ay) * win.nx + x + win.ax;
        const double v = src[j];
        dst[i] = op == IMG_COPY ? v : op == IMG_RESID ? sci[j] - v : 1.0 / (v + obs_var[j]);
    }
}

// acc[i] += sum_w src[(stride*w + c)][i]   (posterior-image sums)
__global__ void k_accumulate(const double* __restrict__ src, double* __restrict__ acc, int S, int n,
                             int stride, int c) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < S; i += gridDim.x * blockDim.x) {
        double s = 0.0;
        for (int w = 0; w < n; ++w) s += src[(size_t)(stride * w + c) * S + i];
        acc[i] += s;
    }
}

// means from the sums: op 0 mean, 1 sci - mean, 2 1 / (mean + obs_var)
__global__ void k_accumulated_out(const double* __restrict__ acc, const double* __restrict__ sci,
                                  const double* __restrict__ obs_var, double* __restrict__ out,
                                  double inv_n, int op, ImgWindow win) {
    const int S_out = win.ly * win.lx;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < S_out; i += gridDim.x * blockDim.x) {
        const int y = i / win.lx, x = i - y * win.lx;
        const int j = (y + win.ay) * win.nx + x + win.ax;
        const double m = acc[j] * inv_n;
        out[i] = op == 0 ? m : op == 1 ? sci[j] - m : 1.0 / (m + obs_var[j]);
    }
}

}  // namespace psfmc
