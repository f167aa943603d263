// psfmc_hip.hip -- C ABI of libpsfmc_hip.so (see include/psfmc_hip.h).
// gfx950 only.  Build: psfmc_amd/csrc/Makefile (hipcc --offload-arch=gfx950).
#include "../../include/psfmc_hip.h"

#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "psfmc_device.h"
#include "psfmc_hipfft_path.h"

using namespace psfmc;

// ---------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------
static thread_local std::string g_err;

static int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                   \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess)                                                           \
            return fail(e_ == hipErrorOutOfMemory ? PSFMC_ENOMEM : PSFMC_EHIP,          \
                        "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                        __LINE__);                                                      \
    } while (0)

#define FFT_TRY(expr)                                                                  \
    do {                                                                               \
        hipfftResult r_ = (expr);                                                      \
        if (r_ != HIPFFT_SUCCESS)                                                      \
            return fail(PSFMC_EHIP, "%s failed: hipfftResult %d (%s:%d)", #expr,       \
                        (int)r_, __FILE__, __LINE__);                                  \
    } while (0)

// ---------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------
struct psfmc_ctx {
    int device = 0;
    int ny = 0, nx = 0, nxh = 0, S = 0, F = 0;
    int n_psf = 0, n_ps = 0, n_sersic = 0;
    int max_walkers = 0, chunk = 0, backend = 0;
    int rlen = 0, plen = 0;
    hipStream_t stream = nullptr;
    // shared field arrays (SURVEY row Cfg)
    double *d_sci = nullptr, *d_var = nullptr;
    uint8_t* d_bad = nullptr;
    double2 *d_pspec = nullptr, *d_vspec = nullptr;   // [n_psf][ny][nxh], = numpy rfft2
    // per-call staging
    double *d_rows = nullptr, *d_prep = nullptr, *d_like = nullptr, *d_partial = nullptr;
    uint8_t* d_skip = nullptr;
    // hipFFT path work space
    double* d_real = nullptr;     // [2*chunk][S]
    double2* d_spec = nullptr;    // [2*chunk][F]
    std::map<int, std::pair<hipfftHandle, hipfftHandle>> plans;   // batch -> (D2Z, Z2D)
    hipfftHandle plan_fwd = 0, plan_inv = 0;                      // the pair in use
    int chi2_blocks = 0;
};

// batched 2-D plans for `batch` images, cached per batch size (a half-ensemble
// call and a full-ensemble call use different sizes)
static int use_plans(psfmc_ctx* c, int batch) {
    auto it = c->plans.find(batch);
    if (it == c->plans.end()) {
        if (c->plans.size() >= 8) {            // bound the cache
            for (auto& kv : c->plans) {
                hipfftDestroy(kv.second.first);
                hipfftDestroy(kv.second.second);
            }
            c->plans.clear();
        }
        int n[2] = {c->ny, c->nx};
        hipfftHandle f = 0, b = 0;
        FFT_TRY(hipfftPlanMany(&f, 2, n, nullptr, 1, 0, nullptr, 1, 0, HIPFFT_D2Z, batch));
        FFT_TRY(hipfftPlanMany(&b, 2, n, nullptr, 1, 0, nullptr, 1, 0, HIPFFT_Z2D, batch));
        it = c->plans.emplace(batch, std::make_pair(f, b)).first;
    }
    c->plan_fwd = it->second.first;
    c->plan_inv = it->second.second;
    return PSFMC_OK;
}

static int alloc_work(psfmc_ctx* c) {
    const size_t nimg = (size_t)2 * c->chunk;
    if (c->d_real) { (void)hipFree(c->d_real); c->d_real = nullptr; }
    if (c->d_spec) { (void)hipFree(c->d_spec); c->d_spec = nullptr; }
    HIP_TRY(hipMalloc(&c->d_real, nimg * c->S * sizeof(double)));
    HIP_TRY(hipMalloc(&c->d_spec, nimg * c->F * sizeof(double2)));
    return use_plans(c, (int)nimg);
}

static bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

extern "C" int psfmc_abi_version(void) { return 1; }

extern "C" const char* psfmc_last_error(void) { return g_err.c_str(); }

extern "C" int psfmc_ctx_create(psfmc_ctx** out, int device, int ny, int nx, const double* sci,
                                const double* obs_var, const uint8_t* bad_px, int n_psf,
                                int psf_ny, int psf_nx, const double* psf, const double* psf_var,
                                int n_ps, int n_sersic, int max_walkers, int backend) {
    if (!out) return fail(PSFMC_EINVAL, "out is NULL");
    *out = nullptr;
    if (!sci || !obs_var || !bad_px || !psf || !psf_var) return fail(PSFMC_EINVAL, "NULL input array");
    if (ny < 2 || nx < 2 || (ny & 1) || (nx & 1))
        return fail(PSFMC_EINVAL, "image sides must be even (got %d x %d)", ny, nx);
    if (n_psf < 1 || psf_ny < 1 || psf_nx < 1 || psf_ny > ny || psf_nx > nx)
        return fail(PSFMC_EINVAL, "PSF larger than the observation is not supported (%d x %d in %d x %d)",
                    psf_ny, psf_nx, ny, nx);
    if (n_ps < 0 || n_sersic < 0 || n_ps > 16 || n_sersic > 16)
        return fail(PSFMC_EINVAL, "component counts out of range (n_ps=%d n_sersic=%d)", n_ps, n_sersic);
    if (max_walkers < 1) return fail(PSFMC_EINVAL, "max_walkers must be >= 1");
    if (backend != PSFMC_BACKEND_HIPFFT && backend != PSFMC_BACKEND_FUSED)
        return fail(PSFMC_EINVAL, "unknown backend %d", backend);
    if (backend == PSFMC_BACKEND_FUSED && !(is_pow2(ny) && is_pow2(nx)))
        return fail(PSFMC_EINVAL, "fused backend needs power-of-two sides (got %d x %d)", ny, nx);
    if (backend == PSFMC_BACKEND_FUSED) return fail(PSFMC_EINVAL, "fused backend not built yet");

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(PSFMC_ENODEV, "no HIP device");
    if (device < 0 || device >= ndev) return fail(PSFMC_ENODEV, "device %d of %d", device, ndev);
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(PSFMC_ENODEV, "device %d is %s; this library is built for gfx950 only", device,
                    prop.gcnArchName);

    psfmc_ctx* c = new psfmc_ctx;
    c->device = device;
    c->ny = ny; c->nx = nx; c->nxh = nx / 2 + 1; c->S = ny * nx; c->F = ny * c->nxh;
    c->n_psf = n_psf; c->n_ps = n_ps; c->n_sersic = n_sersic;
    c->max_walkers = max_walkers; c->backend = backend;
    c->rlen = row_len(n_ps, n_sersic);
    c->plen = prep_len(n_ps, n_sersic);
    c->chi2_blocks = (c->S + 1023) / 1024;
    if (c->chi2_blocks > 64) c->chi2_blocks = 64;
    // walkers per internal pass: keep the work space of the hipFFT path <= ~6 GiB
    const double per_walker = 2.0 * (c->S * 8.0 + c->F * 16.0);
    int chunk = (int)(6.0 * 1073741824.0 / per_walker);
    if (chunk < 1) chunk = 1;
    if (chunk > max_walkers) chunk = max_walkers;
    c->chunk = chunk;

#define CTX_TRY(expr)                         \
    do {                                      \
        int rc_ = (expr);                     \
        if (rc_ != PSFMC_OK) {                \
            psfmc_ctx_destroy(c);             \
            return rc_;                       \
        }                                     \
    } while (0)
#define CTX_HIP(expr) CTX_TRY([&]() -> int { HIP_TRY(expr); return PSFMC_OK; }())

    CTX_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    CTX_HIP(hipMalloc(&c->d_sci, c->S * sizeof(double)));
    CTX_HIP(hipMalloc(&c->d_var, c->S * sizeof(double)));
    CTX_HIP(hipMalloc(&c->d_bad, c->S));
    CTX_HIP(hipMemcpy(c->d_sci, sci, c->S * sizeof(double), hipMemcpyHostToDevice));
    CTX_HIP(hipMemcpy(c->d_var, obs_var, c->S * sizeof(double), hipMemcpyHostToDevice));
    CTX_HIP(hipMemcpy(c->d_bad, bad_px, c->S, hipMemcpyHostToDevice));
    CTX_HIP(hipMalloc(&c->d_pspec, (size_t)n_psf * c->F * sizeof(double2)));
    CTX_HIP(hipMalloc(&c->d_vspec, (size_t)n_psf * c->F * sizeof(double2)));
    CTX_HIP(hipMalloc(&c->d_rows, (size_t)max_walkers * c->rlen * sizeof(double)));
    CTX_HIP(hipMalloc(&c->d_prep, (size_t)max_walkers * c->plen * sizeof(double)));
    CTX_HIP(hipMalloc(&c->d_like, (size_t)max_walkers * sizeof(double)));
    CTX_HIP(hipMalloc(&c->d_skip, (size_t)max_walkers));
    CTX_HIP(hipMalloc(&c->d_partial, (size_t)max_walkers * 64 * sizeof(double)));

    // F0 on the device: pad + forward transform of every PSF and variance map
    {
        const size_t small = (size_t)n_psf * psf_ny * psf_nx;
        double *d_small = nullptr, *d_canvas = nullptr;
        CTX_HIP(hipMalloc(&d_small, 2 * small * sizeof(double)));
        CTX_HIP(hipMalloc(&d_canvas, (size_t)2 * n_psf * c->S * sizeof(double)));
        CTX_HIP(hipMemcpy(d_small, psf, small * sizeof(double), hipMemcpyHostToDevice));
        CTX_HIP(hipMemcpy(d_small + small, psf_var, small * sizeof(double), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_pad, dim3(256), dim3(256), 0, c->stream, d_small, d_canvas, 2 * n_psf,
                           psf_ny, psf_nx, ny, nx);
        hipfftHandle plan;
        int n[2] = {ny, nx};
        int rc = PSFMC_OK;
        if (hipfftPlanMany(&plan, 2, n, nullptr, 1, 0, nullptr, 1, 0, HIPFFT_D2Z, n_psf) != HIPFFT_SUCCESS)
            rc = fail(PSFMC_EHIP, "hipfftPlanMany (PSF spectra) failed");
        if (rc == PSFMC_OK) {
            hipfftSetStream(plan, c->stream);
            if (hipfftExecD2Z(plan, d_canvas, (hipfftDoubleComplex*)c->d_pspec) != HIPFFT_SUCCESS ||
                hipfftExecD2Z(plan, d_canvas + (size_t)n_psf * c->S, (hipfftDoubleComplex*)c->d_vspec) !=
                    HIPFFT_SUCCESS)
                rc = fail(PSFMC_EHIP, "hipfftExecD2Z (PSF spectra) failed");
            (void)hipStreamSynchronize(c->stream);
            hipfftDestroy(plan);
        }
        (void)hipFree(d_small);
        (void)hipFree(d_canvas);
        CTX_TRY(rc);
    }
    CTX_TRY(alloc_work(c));
    CTX_HIP(hipStreamSynchronize(c->stream));
    *out = c;
    return PSFMC_OK;
}

extern "C" int psfmc_ctx_destroy(psfmc_ctx* c) {
    if (!c) return PSFMC_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto& kv : c->plans) {
        hipfftDestroy(kv.second.first);
        hipfftDestroy(kv.second.second);
    }
    void* bufs[] = {c->d_sci,  c->d_var,  c->d_bad,  c->d_pspec,   c->d_vspec, c->d_rows,
                    c->d_prep, c->d_like, c->d_skip, c->d_partial, c->d_real,  c->d_spec};
    for (void* p : bufs)
        if (p) (void)hipFree(p);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return PSFMC_OK;
}

extern "C" int psfmc_row_len(const psfmc_ctx* c) { return c ? c->rlen : PSFMC_EINVAL; }

extern "C" int psfmc_set_option(psfmc_ctx* c, const char* key, double value) {
    if (!c || !key) return fail(PSFMC_EINVAL, "NULL argument");
    if (!strcmp(key, "chunk_walkers")) {
        int v = (int)value;
        if (v < 1 || v > c->max_walkers) return fail(PSFMC_EINVAL, "chunk_walkers out of range");
        HIP_TRY(hipSetDevice(c->device));
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->chunk = v;
        return alloc_work(c);
    }
    return fail(PSFMC_EINVAL, "unknown option '%s'", key);
}

extern "C" double psfmc_get_option(const psfmc_ctx* c, const char* key) {
    if (!c || !key) return NAN;
    if (!strcmp(key, "chunk_walkers")) return c->chunk;
    if (!strcmp(key, "backend")) return c->backend;
    if (!strcmp(key, "max_walkers")) return c->max_walkers;
    return NAN;
}

// ---------------------------------------------------------------------------
// hipFFT path: one chunk of walkers [w0, w0+n)
// ---------------------------------------------------------------------------
static int convolve_chunk(psfmc_ctx* c, int n, const double* d_prep, const uint8_t* d_skip,
                          hipStream_t st, int ps_only) {
    const size_t lds = (size_t)c->plen * sizeof(double);
    int prc = use_plans(c, 2 * n);
    if (prc != PSFMC_OK) return prc;
    hipLaunchKernelGGL(k_raster, dim3((c->S + 1023) / 1024, n), dim3(256), lds, st, d_prep, d_skip,
                       c->d_real, c->n_ps, c->n_sersic, c->ny, c->nx, ps_only);
    FFT_TRY(hipfftSetStream(c->plan_fwd, st));
    FFT_TRY(hipfftSetStream(c->plan_inv, st));
    FFT_TRY(hipfftExecD2Z(c->plan_fwd, c->d_real, (hipfftDoubleComplex*)c->d_spec));
    int gx = (c->F + 255) / 256;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(k_spec_mul, dim3(gx, n), dim3(256), 0, st, c->d_spec, c->d_pspec, c->d_vspec,
                       d_prep, d_skip, c->plen, c->ny, c->nxh, 1.0 / (double)c->S);
    FFT_TRY(hipfftExecZ2D(c->plan_inv, (hipfftDoubleComplex*)c->d_spec, c->d_real));
    return PSFMC_OK;
}

static int eval_device(psfmc_ctx* c, int W, const double* d_rows, const uint8_t* d_skip,
                       double* d_like, hipStream_t st) {
    hipLaunchKernelGGL(k_prep, dim3((W + 127) / 128), dim3(128), 0, st, d_rows, c->d_prep, W, c->n_ps,
                       c->n_sersic, c->ny, c->nx);
    for (int w0 = 0; w0 < W; w0 += c->chunk) {
        const int n = W - w0 < c->chunk ? W - w0 : c->chunk;
        const double* prep = c->d_prep + (size_t)w0 * c->plen;
        const uint8_t* skip = d_skip ? d_skip + w0 : nullptr;
        int rc = convolve_chunk(c, n, prep, skip, st, 0);
        if (rc != PSFMC_OK) return rc;
        hipLaunchKernelGGL(k_chi2, dim3(c->chi2_blocks, n), dim3(256), 0, st, c->d_real, c->d_sci,
                           c->d_var, c->d_bad, skip, c->d_partial + (size_t)w0 * c->chi2_blocks, c->S);
    }
    hipLaunchKernelGGL(k_finish, dim3((W + 127) / 128), dim3(128), 0, st, c->d_partial, d_skip, d_like,
                       W, c->chi2_blocks);
    HIP_TRY(hipGetLastError());
    return PSFMC_OK;
}

static int check_call(psfmc_ctx* c, int W, const void* rows, const void* out) {
    if (!c) return fail(PSFMC_EINVAL, "ctx is NULL");
    if (W < 0 || W > c->max_walkers)
        return fail(PSFMC_EINVAL, "W=%d outside [0, max_walkers=%d]", W, c->max_walkers);
    if (W > 0 && (!rows || !out)) return fail(PSFMC_EINVAL, "NULL buffer");
    return PSFMC_OK;
}

extern "C" int psfmc_eval_batch_device(psfmc_ctx* c, int W, const double* d_rows,
                                       const uint8_t* d_skip, double* d_like, void* stream) {
    int rc = check_call(c, W, d_rows, d_like);
    if (rc != PSFMC_OK || W == 0) return rc;
    HIP_TRY(hipSetDevice(c->device));
    return eval_device(c, W, d_rows, d_skip, d_like, stream ? (hipStream_t)stream : c->stream);
}

extern "C" int psfmc_eval_batch(psfmc_ctx* c, int W, const double* rows, const uint8_t* skip,
                                double* loglike) {
    int rc = check_call(c, W, rows, loglike);
    if (rc != PSFMC_OK || W == 0) return rc;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    HIP_TRY(hipMemcpyAsync(c->d_rows, rows, (size_t)W * c->rlen * sizeof(double), hipMemcpyHostToDevice, st));
    if (skip) HIP_TRY(hipMemcpyAsync(c->d_skip, skip, (size_t)W, hipMemcpyHostToDevice, st));
    rc = eval_device(c, W, c->d_rows, skip ? c->d_skip : nullptr, c->d_like, st);
    if (rc != PSFMC_OK) return rc;
    HIP_TRY(hipMemcpyAsync(loglike, c->d_like, (size_t)W * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return PSFMC_OK;
}

extern "C" int psfmc_eval_images(psfmc_ctx* c, int W, const double* rows, double* raw, double* conv,
                                 double* resid, double* ivm, double* ps_sub) {
    int rc = check_call(c, W, rows, rows);
    if (rc != PSFMC_OK || W == 0) return rc;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const size_t img = (size_t)c->S * sizeof(double);
    HIP_TRY(hipMemcpyAsync(c->d_rows, rows, (size_t)W * c->rlen * sizeof(double), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_prep, dim3((W + 127) / 128), dim3(128), 0, st, c->d_rows, c->d_prep, W, c->n_ps,
                       c->n_sersic, c->ny, c->nx);
    double* d_out = nullptr;     // [chunk][S] staging for derived images
    HIP_TRY(hipMalloc(&d_out, (size_t)c->chunk * img));
    for (int w0 = 0; w0 < W && rc == PSFMC_OK; w0 += c->chunk) {
        const int n = W - w0 < c->chunk ? W - w0 : c->chunk;
        const double* prep = c->d_prep + (size_t)w0 * c->plen;
        auto emit = [&](double* host, int comp, int op) -> int {
            if (!host) return PSFMC_OK;
            hipLaunchKernelGGL(k_image_out, dim3(64, n), dim3(256), 0, st, c->d_real, c->d_sci, c->d_var,
                               d_out, c->S, comp, op);
            HIP_TRY(hipMemcpyAsync(host + (size_t)w0 * c->S, d_out, (size_t)n * img,
                                   hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            return PSFMC_OK;
        };
        if (raw) {   // raw model before it is overwritten by the inverse transform
            hipLaunchKernelGGL(k_raster, dim3((c->S + 1023) / 1024, n), dim3(256),
                               (size_t)c->plen * sizeof(double), st, prep, (const uint8_t*)nullptr,
                               c->d_real, c->n_ps, c->n_sersic, c->ny, c->nx, 0);
            rc = emit(raw, 0, IMG_COPY);
            if (rc != PSFMC_OK) break;
        }
        if (conv || resid || ivm) {
            rc = convolve_chunk(c, n, prep, nullptr, st, 0);
            if (rc == PSFMC_OK) rc = emit(conv, 0, IMG_COPY);
            if (rc == PSFMC_OK) rc = emit(resid, 0, IMG_RESID);
            if (rc == PSFMC_OK) rc = emit(ivm, 1, IMG_IVM);
        }
        if (rc == PSFMC_OK && ps_sub) {
            rc = convolve_chunk(c, n, prep, nullptr, st, 1);
            if (rc == PSFMC_OK) rc = emit(ps_sub, 0, IMG_RESID);
        }
    }
    (void)hipStreamSynchronize(st);
    (void)hipFree(d_out);
    if (rc == PSFMC_OK) HIP_TRY(hipGetLastError());
    return rc;
}

extern "C" int psfmc_get_spectra(psfmc_ctx* c, double* psf_spec, double* var_spec) {
    if (!c || !psf_spec || !var_spec) return fail(PSFMC_EINVAL, "NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    const size_t bytes = (size_t)c->n_psf * c->F * sizeof(double2);
    HIP_TRY(hipMemcpy(psf_spec, c->d_pspec, bytes, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(var_spec, c->d_vspec, bytes, hipMemcpyDeviceToHost));
    return PSFMC_OK;
}
