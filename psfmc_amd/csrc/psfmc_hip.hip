// psfmc_hip.hip -- C ABI of libpsfmc_hip.so (see include/psfmc_hip.h).
// gfx950 only.  Build: psfmc_amd/csrc/Makefile (hipcc --offload-arch=gfx950).
#include "../../include/psfmc_hip.h"
#include "psfmc_side_costs.h"

// The file compiles either as one translation unit (PSFMC_NPARTS undefined: everything) or as
// PSFMC_NPARTS = 4 of them built in parallel (csrc/Makefile): every part instantiates the per-side
// kernels of its share of the sides behind psfmc_size_call_part<k>; part 0 is also the API.
#ifndef PSFMC_NPARTS
#define PSFMC_NPARTS 1
#define PSFMC_PART 0
#endif

#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "psfmc_device.h"
#if PSFMC_PART == 0
#include "psfmc_hipfft_path.h"
#endif
#include "psfmc_fused_path.h"
#include "psfmc_rows3_path.h"
#include "psfmc_theta.h"

using namespace psfmc;

// ---------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------
#define PSFMC_NOT_MINE (-1000)          /* a part's answer for a side another part builds */

int psfmc_fail_(int code, const char* fmt, ...);
#define fail psfmc_fail_
#if PSFMC_PART == 0
static thread_local std::string g_err;

int psfmc_fail_(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#endif

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

#define RC_TRY(expr)                     \
    do {                                 \
        int rc_ = (expr);                \
        if (rc_ != PSFMC_OK) return rc_; \
    } while (0)

// sides the fused kernels are instantiated for (FftShape in psfmc_fft.h): every power of two
// 64..1024 and the even sides with factors 3, 5, 7, 11, 13 listed there
#define PSFMC_FUSED_SIDES "64 84 88 96 98 100 104 110 112 120 126 128 130 132 140 144 150 156 160 168 176 180 192 196 200 208 210 220 224 240 250 252 256 260 264 280 286 288 294 300 308 312 320 330 336 350 352 360 364 384 390 392 400 416 420 440 448 480 484 500 504 512 520 528 560 572 576 600 616 624 630 640 650 660 672 676 700 704 720 728 768 780 784 800 832 840 896 900 960 1024 1152 1280 1536 2048"
// run BODY with `N_` a compile-time copy of the length n; in a split build only for this part's sides
// (side i of the list belongs to part i mod PSFMC_NPARTS)
#if PSFMC_NPARTS == 1
#define DISPATCH_LEN(n, BODY) \
    switch (n) { \
        case 64: { constexpr int N_ = 64; BODY; } break; \
        case 84: { constexpr int N_ = 84; BODY; } break; \
        case 88: { constexpr int N_ = 88; BODY; } break; \
        case 96: { constexpr int N_ = 96; BODY; } break; \
        case 98: { constexpr int N_ = 98; BODY; } break; \
        case 100: { constexpr int N_ = 100; BODY; } break; \
        case 104: { constexpr int N_ = 104; BODY; } break; \
        case 110: { constexpr int N_ = 110; BODY; } break; \
        case 112: { constexpr int N_ = 112; BODY; } break; \
        case 120: { constexpr int N_ = 120; BODY; } break; \
        case 126: { constexpr int N_ = 126; BODY; } break; \
        case 128: { constexpr int N_ = 128; BODY; } break; \
        case 130: { constexpr int N_ = 130; BODY; } break; \
        case 132: { constexpr int N_ = 132; BODY; } break; \
        case 140: { constexpr int N_ = 140; BODY; } break; \
        case 144: { constexpr int N_ = 144; BODY; } break; \
        case 150: { constexpr int N_ = 150; BODY; } break; \
        case 156: { constexpr int N_ = 156; BODY; } break; \
        case 160: { constexpr int N_ = 160; BODY; } break; \
        case 168: { constexpr int N_ = 168; BODY; } break; \
        case 176: { constexpr int N_ = 176; BODY; } break; \
        case 180: { constexpr int N_ = 180; BODY; } break; \
        case 192: { constexpr int N_ = 192; BODY; } break; \
        case 196: { constexpr int N_ = 196; BODY; } break; \
        case 200: { constexpr int N_ = 200; BODY; } break; \
        case 208: { constexpr int N_ = 208; BODY; } break; \
        case 210: { constexpr int N_ = 210; BODY; } break; \
        case 220: { constexpr int N_ = 220; BODY; } break; \
        case 224: { constexpr int N_ = 224; BODY; } break; \
        case 240: { constexpr int N_ = 240; BODY; } break; \
        case 250: { constexpr int N_ = 250; BODY; } break; \
        case 252: { constexpr int N_ = 252; BODY; } break; \
        case 256: { constexpr int N_ = 256; BODY; } break; \
        case 260: { constexpr int N_ = 260; BODY; } break; \
        case 264: { constexpr int N_ = 264; BODY; } break; \
        case 280: { constexpr int N_ = 280; BODY; } break; \
        case 286: { constexpr int N_ = 286; BODY; } break; \
        case 288: { constexpr int N_ = 288; BODY; } break; \
        case 294: { constexpr int N_ = 294; BODY; } break; \
        case 300: { constexpr int N_ = 300; BODY; } break; \
        case 308: { constexpr int N_ = 308; BODY; } break; \
        case 312: { constexpr int N_ = 312; BODY; } break; \
        case 320: { constexpr int N_ = 320; BODY; } break; \
        case 330: { constexpr int N_ = 330; BODY; } break; \
        case 336: { constexpr int N_ = 336; BODY; } break; \
        case 350: { constexpr int N_ = 350; BODY; } break; \
        case 352: { constexpr int N_ = 352; BODY; } break; \
        case 360: { constexpr int N_ = 360; BODY; } break; \
        case 364: { constexpr int N_ = 364; BODY; } break; \
        case 384: { constexpr int N_ = 384; BODY; } break; \
        case 390: { constexpr int N_ = 390; BODY; } break; \
        case 392: { constexpr int N_ = 392; BODY; } break; \
        case 400: { constexpr int N_ = 400; BODY; } break; \
        case 416: { constexpr int N_ = 416; BODY; } break; \
        case 420: { constexpr int N_ = 420; BODY; } break; \
        case 440: { constexpr int N_ = 440; BODY; } break; \
        case 448: { constexpr int N_ = 448; BODY; } break; \
        case 480: { constexpr int N_ = 480; BODY; } break; \
        case 484: { constexpr int N_ = 484; BODY; } break; \
        case 500: { constexpr int N_ = 500; BODY; } break; \
        case 504: { constexpr int N_ = 504; BODY; } break; \
        case 512: { constexpr int N_ = 512; BODY; } break; \
        case 520: { constexpr int N_ = 520; BODY; } break; \
        case 528: { constexpr int N_ = 528; BODY; } break; \
        case 560: { constexpr int N_ = 560; BODY; } break; \
        case 572: { constexpr int N_ = 572; BODY; } break; \
        case 576: { constexpr int N_ = 576; BODY; } break; \
        case 600: { constexpr int N_ = 600; BODY; } break; \
        case 616: { constexpr int N_ = 616; BODY; } break; \
        case 624: { constexpr int N_ = 624; BODY; } break; \
        case 630: { constexpr int N_ = 630; BODY; } break; \
        case 640: { constexpr int N_ = 640; BODY; } break; \
        case 650: { constexpr int N_ = 650; BODY; } break; \
        case 660: { constexpr int N_ = 660; BODY; } break; \
        case 672: { constexpr int N_ = 672; BODY; } break; \
        case 676: { constexpr int N_ = 676; BODY; } break; \
        case 700: { constexpr int N_ = 700; BODY; } break; \
        case 704: { constexpr int N_ = 704; BODY; } break; \
        case 720: { constexpr int N_ = 720; BODY; } break; \
        case 728: { constexpr int N_ = 728; BODY; } break; \
        case 768: { constexpr int N_ = 768; BODY; } break; \
        case 780: { constexpr int N_ = 780; BODY; } break; \
        case 784: { constexpr int N_ = 784; BODY; } break; \
        case 800: { constexpr int N_ = 800; BODY; } break; \
        case 832: { constexpr int N_ = 832; BODY; } break; \
        case 840: { constexpr int N_ = 840; BODY; } break; \
        case 896: { constexpr int N_ = 896; BODY; } break; \
        case 900: { constexpr int N_ = 900; BODY; } break; \
        case 960: { constexpr int N_ = 960; BODY; } break; \
        case 1024: { constexpr int N_ = 1024; BODY; } break; \
        case 1152: { constexpr int N_ = 1152; BODY; } break; \
        case 1280: { constexpr int N_ = 1280; BODY; } break; \
        case 1536: { constexpr int N_ = 1536; BODY; } break; \
        case 2048: { constexpr int N_ = 2048; BODY; } break; \
        default: return fail(PSFMC_EINVAL, "fused backend: side %d is not one of " PSFMC_FUSED_SIDES, n); \
    }
#elif PSFMC_PART == 0
#define DISPATCH_LEN(n, BODY) \
    switch (n) { \
        case 64: { constexpr int N_ = 64; BODY; } break; \
        case 98: { constexpr int N_ = 98; BODY; } break; \
        case 112: { constexpr int N_ = 112; BODY; } break; \
        case 130: { constexpr int N_ = 130; BODY; } break; \
        case 150: { constexpr int N_ = 150; BODY; } break; \
        case 176: { constexpr int N_ = 176; BODY; } break; \
        case 200: { constexpr int N_ = 200; BODY; } break; \
        case 224: { constexpr int N_ = 224; BODY; } break; \
        case 256: { constexpr int N_ = 256; BODY; } break; \
        case 286: { constexpr int N_ = 286; BODY; } break; \
        case 308: { constexpr int N_ = 308; BODY; } break; \
        case 336: { constexpr int N_ = 336; BODY; } break; \
        case 364: { constexpr int N_ = 364; BODY; } break; \
        case 400: { constexpr int N_ = 400; BODY; } break; \
        case 448: { constexpr int N_ = 448; BODY; } break; \
        case 504: { constexpr int N_ = 504; BODY; } break; \
        case 560: { constexpr int N_ = 560; BODY; } break; \
        case 616: { constexpr int N_ = 616; BODY; } break; \
        case 650: { constexpr int N_ = 650; BODY; } break; \
        case 700: { constexpr int N_ = 700; BODY; } break; \
        case 768: { constexpr int N_ = 768; BODY; } break; \
        case 832: { constexpr int N_ = 832; BODY; } break; \
        case 960: { constexpr int N_ = 960; BODY; } break; \
        case 1536: { constexpr int N_ = 1536; BODY; } break; \
        default: return PSFMC_NOT_MINE; \
    }
#elif PSFMC_PART == 1
#define DISPATCH_LEN(n, BODY) \
    switch (n) { \
        case 84: { constexpr int N_ = 84; BODY; } break; \
        case 100: { constexpr int N_ = 100; BODY; } break; \
        case 120: { constexpr int N_ = 120; BODY; } break; \
        case 132: { constexpr int N_ = 132; BODY; } break; \
        case 156: { constexpr int N_ = 156; BODY; } break; \
        case 180: { constexpr int N_ = 180; BODY; } break; \
        case 208: { constexpr int N_ = 208; BODY; } break; \
        case 240: { constexpr int N_ = 240; BODY; } break; \
        case 260: { constexpr int N_ = 260; BODY; } break; \
        case 288: { constexpr int N_ = 288; BODY; } break; \
        case 312: { constexpr int N_ = 312; BODY; } break; \
        case 350: { constexpr int N_ = 350; BODY; } break; \
        case 384: { constexpr int N_ = 384; BODY; } break; \
        case 416: { constexpr int N_ = 416; BODY; } break; \
        case 480: { constexpr int N_ = 480; BODY; } break; \
        case 512: { constexpr int N_ = 512; BODY; } break; \
        case 572: { constexpr int N_ = 572; BODY; } break; \
        case 624: { constexpr int N_ = 624; BODY; } break; \
        case 660: { constexpr int N_ = 660; BODY; } break; \
        case 704: { constexpr int N_ = 704; BODY; } break; \
        case 780: { constexpr int N_ = 780; BODY; } break; \
        case 840: { constexpr int N_ = 840; BODY; } break; \
        case 1024: { constexpr int N_ = 1024; BODY; } break; \
        case 2048: { constexpr int N_ = 2048; BODY; } break; \
        default: return PSFMC_NOT_MINE; \
    }
#elif PSFMC_PART == 2
#define DISPATCH_LEN(n, BODY) \
    switch (n) { \
        case 88: { constexpr int N_ = 88; BODY; } break; \
        case 104: { constexpr int N_ = 104; BODY; } break; \
        case 126: { constexpr int N_ = 126; BODY; } break; \
        case 140: { constexpr int N_ = 140; BODY; } break; \
        case 160: { constexpr int N_ = 160; BODY; } break; \
        case 192: { constexpr int N_ = 192; BODY; } break; \
        case 210: { constexpr int N_ = 210; BODY; } break; \
        case 250: { constexpr int N_ = 250; BODY; } break; \
        case 264: { constexpr int N_ = 264; BODY; } break; \
        case 294: { constexpr int N_ = 294; BODY; } break; \
        case 320: { constexpr int N_ = 320; BODY; } break; \
        case 352: { constexpr int N_ = 352; BODY; } break; \
        case 390: { constexpr int N_ = 390; BODY; } break; \
        case 420: { constexpr int N_ = 420; BODY; } break; \
        case 484: { constexpr int N_ = 484; BODY; } break; \
        case 520: { constexpr int N_ = 520; BODY; } break; \
        case 576: { constexpr int N_ = 576; BODY; } break; \
        case 630: { constexpr int N_ = 630; BODY; } break; \
        case 672: { constexpr int N_ = 672; BODY; } break; \
        case 720: { constexpr int N_ = 720; BODY; } break; \
        case 784: { constexpr int N_ = 784; BODY; } break; \
        case 896: { constexpr int N_ = 896; BODY; } break; \
        case 1152: { constexpr int N_ = 1152; BODY; } break; \
        default: return PSFMC_NOT_MINE; \
    }
#elif PSFMC_PART == 3
#define DISPATCH_LEN(n, BODY) \
    switch (n) { \
        case 96: { constexpr int N_ = 96; BODY; } break; \
        case 110: { constexpr int N_ = 110; BODY; } break; \
        case 128: { constexpr int N_ = 128; BODY; } break; \
        case 144: { constexpr int N_ = 144; BODY; } break; \
        case 168: { constexpr int N_ = 168; BODY; } break; \
        case 196: { constexpr int N_ = 196; BODY; } break; \
        case 220: { constexpr int N_ = 220; BODY; } break; \
        case 252: { constexpr int N_ = 252; BODY; } break; \
        case 280: { constexpr int N_ = 280; BODY; } break; \
        case 300: { constexpr int N_ = 300; BODY; } break; \
        case 330: { constexpr int N_ = 330; BODY; } break; \
        case 360: { constexpr int N_ = 360; BODY; } break; \
        case 392: { constexpr int N_ = 392; BODY; } break; \
        case 440: { constexpr int N_ = 440; BODY; } break; \
        case 500: { constexpr int N_ = 500; BODY; } break; \
        case 528: { constexpr int N_ = 528; BODY; } break; \
        case 600: { constexpr int N_ = 600; BODY; } break; \
        case 640: { constexpr int N_ = 640; BODY; } break; \
        case 676: { constexpr int N_ = 676; BODY; } break; \
        case 728: { constexpr int N_ = 728; BODY; } break; \
        case 800: { constexpr int N_ = 800; BODY; } break; \
        case 900: { constexpr int N_ = 900; BODY; } break; \
        case 1280: { constexpr int N_ = 1280; BODY; } break; \
        default: return PSFMC_NOT_MINE; \
    }
#endif

// ---------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------
struct psfmc_ctx {
    int device = 0;
    int ny = 0, nx = 0, nxh = 0, S = 0, F = 0;
    int nyp = 0;                  // column length of the T layout: ny rounded up to whole row groups
    // An image side the transforms are not built for is EMBEDDED in the next built side >= side + PSF side - 1
    // (psfmc_device.h WrapDesc): ny, nx above are then the TRANSFORM's sides, ly, lx the image's; every pixel
    // array that crosses the C ABI has the image's shape, every internal one the transform's.
    bool embed = false;
    int ly = 0, lx = 0;           // the image's own sides (= ny, nx without embedding)
    WrapDesc wrap{0, 0, 0, 0, 0, 0};
    int n_psf = 0, n_ps = 0, n_sersic = 0;   // n_psf: kernel spectra in all (fields x PSFs per field)
    int n_fields = 1, n_psf_field = 0;       // observed fields of this context (psfmc_ctx_create_fields), PSFs of each
    size_t field_len = 0;                    // packed pixels (FieldPx) of one field
    int max_walkers = 0, chunk = 0, backend = 0;
    // the records now in d_prep have their power tables behind them (k_pow_tables ran); false: small batch,
    // the forward row waves form the entries they need themselves (psfmc_device.h raster_row)
    bool prep_tabs_built = false;
    // every writer of d_prep clears this, launch_pow_tables sets it: a forward launch that rasterises from
    // records whose table mode was not decided for THIS batch is an error, not wrong pixels
    bool prep_tabs_valid = false;
    // false: this context's rasterising kernels evaluate log2 + exp2 per pixel (row lengths up to 256:
    // psfmc_device.h pow_tabs_side) and nothing builds tables
    bool use_pow_tabs = true;
    int single_cap = 0;                               // walkers T buffer 0 holds (>= chunk)
    int rlen = 0, plen = 0;
    int nblk = 0;                 // chi^2 partial sums per walker
    hipStream_t stream = nullptr;
    // shared field arrays (SURVEY row Cfg)
    double *d_sci = nullptr, *d_var = nullptr;
    uint8_t* d_bad = nullptr;
    // per-call staging
    double *d_rows = nullptr, *d_prep = nullptr, *d_like = nullptr, *d_partial = nullptr;
    uint8_t* d_skip = nullptr;
    // hipFFT path
    double2 *d_pspec = nullptr, *d_vspec = nullptr;   // [n_psf][ny][nxh] == numpy rfft2
    double* d_real = nullptr;                         // [2*chunk][S]
    double2* d_spec = nullptr;                        // [2*chunk][F]
    std::map<int, std::pair<hipfftHandle, hipfftHandle>> plans;   // batch -> (D2Z, Z2D)
    hipfftHandle plan_fwd = 0, plan_inv = 0;                      // the pair in use
    // fused path
    // [chunk][nxh][ny/RG][2][RG] half-spectra of one pass; pass i uses buffer and
    // side stream i % n_streams so neighbouring passes overlap
    static constexpr int kMaxStreams = 4;
    cd* d_Ts[kMaxStreams] = {nullptr, nullptr, nullptr, nullptr};
    cd* d_T = nullptr;        // = d_Ts[0]
    hipStream_t side[kMaxStreams] = {nullptr, nullptr, nullptr, nullptr};   // side[0] unused
    hipEvent_t ev_fork = nullptr, ev_join[kMaxStreams] = {nullptr, nullptr, nullptr, nullptr};
    int n_streams = 2;
    int stagger = 0;          // the second lane starts one forward-row kernel late (run_pipeline); set per shape
    hipEvent_t ev_stagger[kMaxStreams] = {nullptr, nullptr, nullptr, nullptr};
    // "exclusive" kernels (set_option "exclusive", bit 0 forward rows, bit 1 columns, bit 2 inverse rows): a kernel of
    // that kind of pass i + 1 (the other lane) waits for the same kind of pass i, so that the two lanes never run two
    // like kernels side by side -- a VALU-bound forward kernel then always meets the other lane's memory-bound ones.
    // Measured (round 4, profiles/r4_exclusive.txt): SLOWER in every combination, 1024^2 50.3 k -> 43.7 ... 46.4 k
    // evals/s, 512^2 268 k -> 249 ... 266 k: the lanes spend half their time like against like by themselves
    // (profiles/r4_lane_timeline_*.txt: forward + forward alone 32 % of the wall time at 1024^2) and that is the
    // better state -- two forward launches side by side are three whole rounds of row waves.  Off.
    int exclusive = 0;
    hipEvent_t ev_excl[3][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};
    bool t_f32 = false;       // T stored as complex64 (set_option "storage_f32"); arithmetic stays fp64
    bool plain_shape = false; // both sides power-of-two shapes (what storage_f32 is built for)
    bool row_fast = false;    // nx a power-of-two shape and ny a whole number of its row workgroups
    // which row kernels are the one-row-per-wave three-stage ones (psfmc_rows3_path.h): both above 1024, the inverse
    // one at the sides of rows3_inv_default; PSFMC_ROWS3 = 1 / 0 in the environment forces every built one / none
    bool rows3_fwd = false, rows3_inv = false;
    long long speculated_runs = 0;   // psfmc_stretch_run calls that took the whole-iteration route
    int speculate = -1;       // device sampler: ensembles of up to 2 x this many walkers run ONE pipeline pass per iteration (0 = never, -1 = the default rule)
    int cols3 = 1;            // column kernel on the wave-wide three-stage engines: 0 never, 1 the defaults (k_cols3f at 512 / 1536 / 2048, k_cols3 at 1024, k_cols3g at the other sides of fft3g_pick), 2 k_cols3g at those four as well, 3 round 3's k_cols3 at 512 / 1024, 4 k_cols3f at 1024 too
    bool use_graph = false;   // psfmc_stretch_run replays a captured iteration (set_option "graph"; measured: no gain,
                              // the iteration is kernel-time- not launch-bound)
    long long graph_launches = 0;
    int min_split = 1 << 30;  // split a single-pass batch over both streams from this size (off: no gain measured)
    // optional per-kernel timing with HIP events (set_option "profile")
    bool profile = false;
    struct ProfRec { int kind; hipEvent_t a, b; };
    std::vector<ProfRec> prof_pending;
    double prof_ms[3] = {0, 0, 0};
    long prof_n[3] = {0, 0, 0};
    cd* d_Kraw = nullptr;     // [n_psf][2][nxh][ny] kernel spectra, unscaled
    cd* d_Kt = nullptr;       // same * (-1)^(kx+ky) / S
    cd *d_twx = nullptr, *d_twy = nullptr;            // exp(-2 pi i k/n) tables
    double* d_rho = nullptr;  // [n_psf] power-of-two scale of the variance channel
    FieldPx* d_field = nullptr;   // [S] {sci|NaN, obs_var} in rows_inv lane order
    int rg_log2 = 0;          // log2(rows per wave of the row kernels)
    double *d_img0 = nullptr, *d_img1 = nullptr;      // [chunk][S] staging for eval_images
    int img_cap = 0;
    // raw-vector path (psfmc_set_layout)
    bool has_layout = false;
    ThetaLayout layout{};
    // fields 1.. of a multi-field context: their own layouts (same structure, own priors / constants)
    std::vector<ThetaLayout> more_layouts;
    std::vector<void*> more_blobs;
    std::vector<char> more_has;
    size_t theta_lds = 0;
    ThetaLayout* d_field_layouts = nullptr;  // [n_fields] device copies of the layouts (launches that span several fields)
    void* d_layout_blob = nullptr;           // one allocation behind the layout's pointers
    double *d_theta = nullptr, *d_extra = nullptr, *d_lnprior = nullptr;
    double* d_acc = nullptr;  // [n_fields][4][S] sums: raw, conv, model variance, PS-only conv
    // fused path: samples are first added to three LINEAR sums per PSF -- raw, raw^2, PS-only raw --
    // (k_raster_sums) and convolved into d_acc only when the images are asked for (flush_linear_sums)
    double* d_lin = nullptr;      // [n_psf][3][S]
    double* d_linpart = nullptr;  // [lin_groups][n_psf][3][S] partial sums of one call
    int lin_groups = 0;
    long long lin_pending = 0;    // samples in d_lin not yet convolved into d_acc
    bool linear_acc = true;       // set_option "linear_accumulation" 0: every sample through the full pipeline (round 1's way)
    double* d_rawstage = nullptr;   // [img_cap][S] raw-model staging for the sums
    long long acc_count = 0;        // samples in the sums of field 0 (the only field of an ordinary context)
    std::vector<long long> acc_more;   // ... of fields 1.. (psfmc_ctx_create_fields)
    int cols_grid = 0;
    // device-resident sampler state (psfmc_stretch_*): grow-only buffers
    struct Stretch {
        double *pos = nullptr, *lnp = nullptr, *q = nullptr, *newlnp = nullptr, *rand = nullptr;
        double *chain = nullptr, *lnchain = nullptr;
        int *partner = nullptr, *iter = nullptr;
        long long* nacc = nullptr;
        uint8_t* accflag = nullptr;        // whole-iteration launches: which first-half proposals were accepted
        size_t cap_acc = 0;
        bool spec = false;                 // this run proposes whole iterations (stretch_run_impl)
        size_t cap_pos = 0, cap_lnp = 0, cap_q = 0, cap_new = 0, cap_rand = 0, cap_chain = 0, cap_lnchain = 0,
               cap_partner = 0, cap_iter = 0, cap_nacc = 0;
        int W = 0, n_iter = 0;
        int F = 1;                // ensembles in the run (psfmc_stretch_run_fields: one per field)
        bool store = false, open = false;
    } stretch;
};

// the image's window inside the transform-shaped pixel arrays (all of them unless the image is embedded)
static ImgWindow img_window(const psfmc_ctx* c) { return ImgWindow{c->nx, c->ly, c->lx, c->wrap.ay, c->wrap.ax}; }

// samples in the posterior-image sums of one field
static long long& acc_n(psfmc_ctx* c, int field) { return field == 0 ? c->acc_count : c->acc_more[field - 1]; }

// per-side constants of the row kernels
struct RowShape { int rg, fast_waves, fast_rg_log2, regs; bool plain; };

// ---------------------------------------------------------------------------
// fused path launchers
// ---------------------------------------------------------------------------
// TS = the T element type: cd (complex128) or cf (complex64 STORAGE, set_option "storage_f32";
// built for the power-of-two shapes only)
template <int N> constexpr bool plain_side() {
    if constexpr (two_stage_side(N)) return FftShape<N>::kPlain;
    else return false;
}
template <int N, typename TS> constexpr bool storage_built() { return sizeof(TS) == sizeof(cd) || plain_side<N>(); }

// ---- the three-stage row kernels (psfmc_rows3_path.h) ----
template <int NX, bool FROM_IMAGE, bool WRAP>
static int launch_rows3_fwd_kernel(psfmc_ctx* c, int n, const double* prep, const uint8_t* skip, cd* Tbuf, int ps_only,
                                   const double* img, const double* img_scale, double* raw_out, hipStream_t st) {
    using S = typename Rows3<NX>::S;
    constexpr size_t lds = rows3_lds_bytes<S>();
    if constexpr (lds > 64 * 1024) {
        static thread_local int attr_device = -1;
        if (attr_device != c->device) {
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rows3_fwd<NX, FROM_IMAGE, WRAP>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_device = c->device;
        }
    }
    if (!FROM_IMAGE && !c->prep_tabs_valid)
        return fail(PSFMC_EINVAL, "internal: forward rows launched on prep records without a power-table decision");
    hipLaunchKernelGGL((k_rows3_fwd<NX, FROM_IMAGE, WRAP>), dim3((c->ny + rows3_waves(NX) - 1) / rows3_waves(NX), n),
                       dim3(rows3_threads(NX)), lds, st, prep, skip, c->d_twx, Tbuf, c->n_ps, c->n_sersic, c->ny, ps_only, img,
                       img_scale, raw_out, c->wrap, c->prep_tabs_built ? kPowTabsBuilt : kPowTabsInWave);
    return PSFMC_OK;
}
template <int NX, bool FROM_IMAGE>
static int launch_rows3_fwd(psfmc_ctx* c, int n, const double* prep, const uint8_t* skip, cd* Tbuf, int ps_only,
                            const double* img, const double* img_scale, double* raw_out, hipStream_t st) {
    if constexpr (!rows3_fwd_built(NX)) {
        return fail(PSFMC_EINVAL, "side %d has no three-stage row kernels", NX);
    } else {
        if constexpr (!FROM_IMAGE) {
            if (c->embed)
                return launch_rows3_fwd_kernel<NX, false, true>(c, n, prep, skip, Tbuf, ps_only, img, img_scale, raw_out, st);
        }
        return launch_rows3_fwd_kernel<NX, FROM_IMAGE, false>(c, n, prep, skip, Tbuf, ps_only, img, img_scale, raw_out, st);
    }
}
template <int NX, bool MULTI>
static int launch_rows3_inv_kernel(psfmc_ctx* c, int n, const cd* Tbuf, const double* prep, const uint8_t* skip,
                                   double* partial, double* conv_out, double* var_out, hipStream_t st) {
    using S = typename Rows3<NX>::S;
    constexpr size_t lds = rows3_lds_bytes<S>();
    if constexpr (lds > 64 * 1024) {
        static thread_local int attr_device = -1;
        if (attr_device != c->device) {
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rows3_inv<NX, MULTI>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_device = c->device;
        }
    }
    hipLaunchKernelGGL((k_rows3_inv<NX, MULTI>), dim3((c->ny + rows3_waves(NX) - 1) / rows3_waves(NX), n), dim3(rows3_threads(NX)), lds,
                       st, Tbuf, skip, c->d_twx, c->d_field, partial, c->ny, prep, c->plen, conv_out, var_out,
                       c->n_fields > 1 ? c->n_psf_field : 0, (unsigned)c->field_len);
    return PSFMC_OK;
}
template <int NX>
static int launch_rows3_inv(psfmc_ctx* c, int n, const cd* Tbuf, const double* prep, const uint8_t* skip,
                            double* partial, double* conv_out, double* var_out, hipStream_t st) {
    if constexpr (!rows3_side<NX>()) {
        return fail(PSFMC_EINVAL, "side %d has no three-stage row kernels", NX);
    } else {
        if (c->n_fields > 1) return launch_rows3_inv_kernel<NX, true>(c, n, Tbuf, prep, skip, partial, conv_out, var_out, st);
        return launch_rows3_inv_kernel<NX, false>(c, n, Tbuf, prep, skip, partial, conv_out, var_out, st);
    }
}

template <int NX, bool FROM_IMAGE, typename TS, bool FAST, bool WRAP>
static int launch_rows_fwd_kernel(psfmc_ctx* c, int n, const double* prep, const uint8_t* skip, TS* Tbuf,
                                  int ps_only, const double* img, const double* img_scale, double* raw_out,
                                  hipStream_t st) {
    constexpr size_t lds = fused_row_lds_bytes<NX, FAST>();
    if constexpr (lds > 64 * 1024) {
        static thread_local int attr_device = -1;
        if (attr_device != c->device) {
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rows_fwd<NX, FROM_IMAGE, TS, FAST, WRAP>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_device = c->device;
        }
    }
    constexpr int waves = row_waves<NX, FAST>();
    if (!FROM_IMAGE && !c->prep_tabs_valid)
        return fail(PSFMC_EINVAL, "internal: forward rows launched on prep records without a power-table decision");
    hipLaunchKernelGGL((k_rows_fwd<NX, FROM_IMAGE, TS, FAST, WRAP>), dim3((c->nblk + waves - 1) / waves, n),
                       dim3((row_threads<NX, FAST>())), lds, st, prep, skip, c->d_twx, Tbuf, c->n_ps, c->n_sersic,
                       c->ny, ps_only, img, img_scale, raw_out, c->wrap,
                       c->prep_tabs_built ? kPowTabsBuilt : kPowTabsInWave);
    return PSFMC_OK;
}

template <int NX, bool FROM_IMAGE, typename TS, bool FAST>
static int launch_rows_fwd_impl(psfmc_ctx* c, int n, const double* prep, const uint8_t* skip, TS* Tbuf,
                                int ps_only, const double* img, const double* img_scale, double* raw_out,
                                hipStream_t st) {
    // the rasteriser of an embedded image wraps its coordinates (images from memory -- the PSF canvases at
    // set-up, the posterior sums -- are in transform coordinates already)
    if constexpr (!FROM_IMAGE && sizeof(TS) == sizeof(cd)) {
        if (c->embed)
            return launch_rows_fwd_kernel<NX, FROM_IMAGE, TS, FAST, true>(c, n, prep, skip, Tbuf, ps_only, img,
                                                                          img_scale, raw_out, st);
    }
    if (!FROM_IMAGE && c->embed) return fail(PSFMC_EINVAL, "single-precision storage does not serve embedded images");
    return launch_rows_fwd_kernel<NX, FROM_IMAGE, TS, FAST, false>(c, n, prep, skip, Tbuf, ps_only, img, img_scale,
                                                                   raw_out, st);
}

// c->row_fast: the unguarded power-of-two row kernels (ny a whole number of their workgroups);
// otherwise the guarded general code path of the same shape
template <int NX, bool FROM_IMAGE, typename TS = cd>
static int launch_rows_fwd(psfmc_ctx* c, int n, const double* prep, const uint8_t* skip, void* Tvoid,
                           int ps_only, const double* img, const double* img_scale, double* raw_out,
                           hipStream_t st) {
    if constexpr (!storage_built<NX, TS>()) {
        return fail(PSFMC_EINVAL, "single-precision storage is built for power-of-two sides only");
    } else if constexpr (!two_stage_side(NX)) {
        return launch_rows3_fwd<NX, FROM_IMAGE>(c, n, prep, skip, static_cast<cd*>(Tvoid), ps_only, img, img_scale, raw_out, st);
    } else {
        TS* Tbuf = static_cast<TS*>(Tvoid);
        if constexpr (sizeof(TS) == sizeof(cd)) {
            if (c->rows3_fwd)
                return launch_rows3_fwd<NX, FROM_IMAGE>(c, n, prep, skip, static_cast<cd*>(Tvoid), ps_only, img, img_scale,
                                                        raw_out, st);
        }
        if constexpr (FftShape<NX>::kPlain) {
            if (c->row_fast)
                return launch_rows_fwd_impl<NX, FROM_IMAGE, TS, true>(c, n, prep, skip, Tbuf, ps_only, img,
                                                                      img_scale, raw_out, st);
        }
        if constexpr (sizeof(TS) == sizeof(cd)) {
            return launch_rows_fwd_impl<NX, FROM_IMAGE, TS, false>(c, n, prep, skip, Tbuf, ps_only, img,
                                                                   img_scale, raw_out, st);
        } else {
            return fail(PSFMC_EINVAL, "single-precision storage needs the unguarded row kernels");
        }
    }
}

template <int NY, bool CONVOLVE, typename TS = cd>
static int launch_cols(psfmc_ctx* c, void* Tvoid, int n_w, const double* prep, const uint8_t* skip,
                       hipStream_t st) {
  if constexpr (!storage_built<NY, TS>()) {
    return fail(PSFMC_EINVAL, "single-precision storage is built for power-of-two sides only");
  } else {
    TS* Tbuf = static_cast<TS*>(Tvoid);
    const int n_cols = n_w * 2 * c->nxh;
    if constexpr (cols3f_shape<Fft3gShape<NY>>() && sizeof(TS) == sizeof(cd)) {
        // ny = 8 m x 64 (512, 1024, 1536, 2048): the general three-stage engine run forward both ways (round 4).
        // In the flow (same box, whole step): 512^2 +1.6 %, 1536^2 +18 %, 2048^2 +2 % (+5.4 % more with its load
        // pipeline, cols3f_prefetch); at 1024 its third wave per SIMD measured SLOWER than k_cols3 (57.5 vs 52.7 us
        // per 6-walker pass, step -0.8 %: 6 MB of columns in flight per XCD against a 4-MiB L2 that also has to hold
        // the kernel-spectrum columns), with the load pipeline at two waves per SIMD it is the faster one (51.4 vs
        // 52.6 us in the flow, step +0.5 %).  cols3 = 3 asks for k_cols3 at 512 / 1024
        if (c->cols3 == 1 || c->cols3 == 4) {
            using S3 = Fft3gShape<NY>;
            constexpr size_t lds3 = fused_col3f_lds_bytes<S3>();
            static thread_local int attr3f_device = -1;
            if (attr3f_device != c->device) {
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cols3f<NY, CONVOLVE>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3));
                attr3f_device = c->device;
            }
            const int per_block = cols3f_waves<S3>();
            const int blocks = (n_cols + per_block - 1) / per_block;
            const int grid3 = blocks < 4 * c->cols_grid ? blocks : 4 * c->cols_grid;
            hipLaunchKernelGGL((k_cols3f<NY, CONVOLVE>), dim3(grid3), dim3(cols3f_threads<S3>()), lds3, st, Tbuf, c->d_Kt, prep, skip,
                               c->d_twy, c->plen, c->nxh, n_w, c->rg_log2);
            return PSFMC_OK;
        }
    }
    if constexpr (NY == 512 || NY == 1024) {          // long power-of-two columns: wave-wide three-stage engine
        if (c->cols3 == 3 || (c->cols3 && sizeof(TS) != sizeof(cd))) {
            constexpr size_t lds3 = fused_col3_lds_bytes<NY>();
            const int per_block = kColThreads / 64;
            const int blocks = (n_cols + per_block - 1) / per_block;
            const int grid3 = blocks < 4 * c->cols_grid ? blocks : 4 * c->cols_grid;
            hipLaunchKernelGGL((k_cols3<NY, CONVOLVE, TS>), dim3(grid3), dim3(kColThreads), lds3, st, Tbuf,
                               c->d_Kt, prep, skip, c->d_twy, c->plen, c->nxh, n_w, c->rg_log2);
            return PSFMC_OK;
        }
    }
    if constexpr (cols3g_side<NY>() && sizeof(TS) == sizeof(cd)) {   // the sides of psfmc_fft.h fft3g_pick: general three-stage engine
        if (c->cols3 && cols3g_layout_ok<Fft3gShape<NY>>(c->rg_log2)) {
            constexpr size_t lds3 = fused_col3g_lds_bytes<Fft3gShape<NY>>();
            static thread_local int attr3_device = -1;
            if (attr3_device != c->device) {
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cols3g<NY, CONVOLVE>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3));
                attr3_device = c->device;
            }
            const int per_block = cols3g_waves<Fft3gShape<NY>>();
            const int blocks = (n_cols + per_block - 1) / per_block;
            const int grid3 = blocks < 4 * c->cols_grid ? blocks : 4 * c->cols_grid;
            hipLaunchKernelGGL((k_cols3g<NY, CONVOLVE>), dim3(grid3), dim3(cols3g_threads<Fft3gShape<NY>>()), lds3, st, Tbuf, c->d_Kt, prep,
                               skip, c->d_twy, c->plen, c->nxh, n_w, c->rg_log2);
            return PSFMC_OK;
        }
    }
    if constexpr (!two_stage_side(NY)) {
        return fail(PSFMC_EINVAL, "side %d has only the three-stage column kernel (option cols3 = 0 and row groups "
                    "other than 4 do not apply)", NY);
    } else {
    constexpr size_t lds = fused_col_lds_bytes<NY>();
    static thread_local int attr_device = -1;          // raise the dynamic-LDS limit once per device
    if (attr_device != c->device) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cols<NY, CONVOLVE, TS>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_device = c->device;
    }
    const int groups = (n_cols + col_ffts_per_block<NY>() - 1) / col_ffts_per_block<NY>();
    const int grid = groups < c->cols_grid ? groups : c->cols_grid;
    hipLaunchKernelGGL((k_cols<NY, CONVOLVE, TS>), dim3(grid), dim3(kColThreads), lds, st, Tbuf, c->d_Kt,
                       prep, skip, c->d_twy, c->plen, c->nxh, n_w, c->rg_log2);
    return PSFMC_OK;
    }
  }
}

// which column kernel launch_cols takes for this context: 0 k_cols, 1 k_cols3, 2 k_cols3g, 3 k_cols3f (the same conditions)
template <int NY> static int col_engine_code(const psfmc_ctx* c) {
    if constexpr (cols3f_shape<Fft3gShape<NY>>()) {
        if ((c->cols3 == 1 || c->cols3 == 4) && !c->t_f32) return 3;
    }
    if constexpr (NY == 512 || NY == 1024) {
        if (c->cols3 == 3 || (c->cols3 && c->t_f32)) return 1;
    }
    if constexpr (cols3g_side<NY>()) {
        if (!c->t_f32 && c->cols3 && cols3g_layout_ok<Fft3gShape<NY>>(c->rg_log2)) return 2;
    }
    return 0;
}

template <int NX, typename TS, bool FAST, bool MULTI>
static int launch_rows_inv_kernel(psfmc_ctx* c, int n, const TS* Tbuf, const double* prep, const uint8_t* skip,
                                  double* partial, double* conv_out, double* var_out, hipStream_t st) {
    constexpr size_t lds = fused_row_lds_bytes<NX, FAST>();
    if constexpr (lds > 64 * 1024) {
        static thread_local int attr_device = -1;
        if (attr_device != c->device) {
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rows_inv<NX, TS, FAST, MULTI>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_device = c->device;
        }
    }
    constexpr int waves = row_waves<NX, FAST>();
    const int gx = (c->nblk + waves - 1) / waves;
    hipLaunchKernelGGL((k_rows_inv<NX, TS, FAST, MULTI>), dim3(gx, n),
                       dim3((row_threads<NX, FAST>())), lds, st, Tbuf, skip, c->d_twx, c->d_field, partial, c->ny,
                       prep, c->plen, conv_out, var_out, c->n_fields > 1 ? c->n_psf_field : 0, (unsigned)c->field_len);
    return PSFMC_OK;
}

template <int NX, typename TS, bool FAST>
static int launch_rows_inv_impl(psfmc_ctx* c, int n, const TS* Tbuf, const double* prep, const uint8_t* skip,
                                double* partial, double* conv_out, double* var_out, hipStream_t st) {
    if constexpr (sizeof(TS) == sizeof(cd) && NX >= 1024) {
        if (c->n_fields > 1)
            return launch_rows_inv_kernel<NX, TS, FAST, true>(c, n, Tbuf, prep, skip, partial, conv_out, var_out, st);
    } else if constexpr (sizeof(TS) != sizeof(cd)) {
        if (c->n_fields > 1) return fail(PSFMC_EINVAL, "single-precision storage serves contexts of one field");
    }
    return launch_rows_inv_kernel<NX, TS, FAST, false>(c, n, Tbuf, prep, skip, partial, conv_out, var_out, st);
}

template <int NX, typename TS = cd>
static int launch_rows_inv(psfmc_ctx* c, int n, const void* Tvoid, const double* prep, const uint8_t* skip,
                           double* partial, double* conv_out, double* var_out, hipStream_t st) {
    if constexpr (!storage_built<NX, TS>()) {
        return fail(PSFMC_EINVAL, "single-precision storage is built for power-of-two sides only");
    } else if constexpr (!two_stage_side(NX)) {
        return launch_rows3_inv<NX>(c, n, static_cast<const cd*>(Tvoid), prep, skip, partial, conv_out, var_out, st);
    } else {
        const TS* Tbuf = static_cast<const TS*>(Tvoid);
        if constexpr (sizeof(TS) == sizeof(cd)) {
            if (c->rows3_inv)
                return launch_rows3_inv<NX>(c, n, static_cast<const cd*>(Tvoid), prep, skip, partial, conv_out, var_out, st);
        }
        if constexpr (FftShape<NX>::kPlain) {
            if (c->row_fast)
                return launch_rows_inv_impl<NX, TS, true>(c, n, Tbuf, prep, skip, partial, conv_out, var_out, st);
        }
        if constexpr (sizeof(TS) == sizeof(cd)) {
            return launch_rows_inv_impl<NX, TS, false>(c, n, Tbuf, prep, skip, partial, conv_out, var_out, st);
        } else {
            return fail(PSFMC_EINVAL, "single-precision storage needs the unguarded row kernels");
        }
    }
}

template <int NX> static int pack_field(psfmc_ctx* c, int f) {
    const size_t px = (size_t)f * c->S;
    if constexpr (rows3_side<NX>()) {
        if (c->rows3_inv) {
            hipLaunchKernelGGL((k_pack_field3<NX>), dim3(256), dim3(256), 0, c->stream, c->d_sci + px, c->d_var + px,
                               c->d_bad + px, c->d_field + (size_t)f * c->field_len, c->ny);
            return PSFMC_OK;
        }
    }
    if constexpr (two_stage_side(NX)) {
        hipLaunchKernelGGL((k_pack_field<NX>), dim3(256), dim3(256), 0, c->stream, c->d_sci + px, c->d_var + px,
                           c->d_bad + px, c->d_field + (size_t)f * c->field_len, c->ny);
        return PSFMC_OK;
    } else {
        return fail(PSFMC_EINVAL, "side %d needs the three-stage row kernels", NX);
    }
}
// per-side constants of the row kernels the context will launch (rows3: the three-stage family)
template <int NX> static RowShape row_shape_of(bool rows3) {
    if constexpr (rows3_side<NX>()) {
        if (rows3 || !two_stage_side(NX)) return RowShape{1, rows3_waves(NX), rows3_rg_log2(NX), Rows3<NX>::S::R1, false};
    }
    if constexpr (two_stage_side(NX))
        return RowShape{row_group<NX>(), row_waves<NX, true>(), layout_rg_log2<NX, true>(), FftShape<NX>::R,
                        FftShape<NX>::kPlain};
    else
        return RowShape{1, rows3_waves(NX), rows3_rg_log2(NX), 0, false};
}
// bit 0: a two-stage family exists; bit 1: the three-stage inverse kernel is built; bit 2: the forward one;
// bit 3: the three-stage inverse kernel is the default
template <int NX> constexpr int row_families() {
    return (two_stage_side(NX) ? 1 : 0) | (rows3_side<NX>() ? 2 : 0) | (rows3_fwd_built(NX) ? 4 : 0) |
           (rows3_side<NX>() && rows3_inv_default(NX) ? 8 : 0);
}
template <int NX> static size_t field_len_of(int ny, bool rows3) {
    if constexpr (rows3_side<NX>()) {
        if (rows3 || !two_stage_side(NX)) return rows3_field_len<NX>(ny);
    }
    if constexpr (two_stage_side(NX)) return fused_field_len<NX>(ny);
    else return 0;
}
template <int NX> constexpr bool has_rows3() { return rows3_side<NX>(); }

template <int NX>
static int launch_raster_sums(psfmc_ctx* c, int n, const double* prep, int groups, int group_size, hipStream_t st,
                              int per_field, int f0) {
    constexpr int RG = RasterShape<NX>::TPW;
    if (c->embed) {
        hipLaunchKernelGGL((k_raster_sums<NX, true>), dim3((c->ny + RG - 1) / RG, groups), dim3(64), 0, st, prep,
                           c->plen, n, group_size, c->n_ps, c->n_sersic, c->ny, c->n_psf, c->d_linpart, per_field,
                           f0, c->n_psf_field, c->wrap);
        return PSFMC_OK;
    }
    hipLaunchKernelGGL((k_raster_sums<NX, false>), dim3((c->ny + RG - 1) / RG, groups), dim3(64), 0, st, prep, c->plen, n,
                       group_size, c->n_ps, c->n_sersic, c->ny, c->n_psf, c->d_linpart, per_field, f0,
                       c->n_psf_field, c->wrap);
    return PSFMC_OK;
}


// ---------------------------------------------------------------------------
// size-erased entry to the per-side launchers: what crosses the boundary between the parts
// ---------------------------------------------------------------------------
enum SizeOp { SZ_ROW_SHAPE, SZ_FIELD_LEN, SZ_PACK_FIELD, SZ_ROWS_FWD, SZ_COLS, SZ_ROWS_INV, SZ_RASTER_SUMS, SZ_COL_ENGINE };
struct SizeCall {
    psfmc_ctx* c = nullptr;
    int n = 0;                              // walkers
    const double* prep = nullptr;
    const uint8_t* skip = nullptr;
    void* T = nullptr;
    int ps_only = 0;
    const double *img = nullptr, *img_scale = nullptr;
    double *raw_out = nullptr, *partial = nullptr, *conv_out = nullptr, *var_out = nullptr;
    hipStream_t st = nullptr;
    bool from_image = false, convolve = true, f32 = false, rows3 = false;
    int field = 0, groups = 0, group_size = 0, ny = 0, per_field = 0;
    RowShape* shape = nullptr;
    size_t* len = nullptr;
    int* code = nullptr;
};

static int size_call_here(int op, int side, SizeCall& a) {
    psfmc_ctx* c = a.c;
    switch (op) {
        case SZ_ROW_SHAPE:
            // (a.rows3: the caller asks for the three-stage row family where the side has both; a.code: whether it has it)
            DISPATCH_LEN(side, (*a.shape = row_shape_of<N_>(a.rows3), *a.code = row_families<N_>()));
            return PSFMC_OK;
        case SZ_FIELD_LEN:
            DISPATCH_LEN(side, *a.len = field_len_of<N_>(a.ny, a.rows3));
            return PSFMC_OK;
        case SZ_PACK_FIELD:
            DISPATCH_LEN(side, RC_TRY(pack_field<N_>(c, a.field)));
            return PSFMC_OK;
        case SZ_ROWS_FWD:
            if (a.from_image) {
                DISPATCH_LEN(side, RC_TRY((launch_rows_fwd<N_, true>(c, a.n, a.prep, a.skip, a.T, a.ps_only, a.img,
                                                                     a.img_scale, a.raw_out, a.st))));
            } else if (a.f32) {
                DISPATCH_LEN(side, RC_TRY((launch_rows_fwd<N_, false, cf>(c, a.n, a.prep, a.skip, a.T, a.ps_only, a.img,
                                                                          a.img_scale, a.raw_out, a.st))));
            } else {
                DISPATCH_LEN(side, RC_TRY((launch_rows_fwd<N_, false>(c, a.n, a.prep, a.skip, a.T, a.ps_only, a.img,
                                                                      a.img_scale, a.raw_out, a.st))));
            }
            return PSFMC_OK;
        case SZ_COLS:
            if (!a.convolve) {
                DISPATCH_LEN(side, RC_TRY((launch_cols<N_, false>(c, a.T, a.n, a.prep, a.skip, a.st))));
            } else if (a.f32) {
                DISPATCH_LEN(side, RC_TRY((launch_cols<N_, true, cf>(c, a.T, a.n, a.prep, a.skip, a.st))));
            } else {
                DISPATCH_LEN(side, RC_TRY((launch_cols<N_, true>(c, a.T, a.n, a.prep, a.skip, a.st))));
            }
            return PSFMC_OK;
        case SZ_ROWS_INV:
            if (a.f32) {
                DISPATCH_LEN(side, RC_TRY((launch_rows_inv<N_, cf>(c, a.n, a.T, a.prep, a.skip, a.partial, a.conv_out,
                                                                   a.var_out, a.st))));
            } else {
                DISPATCH_LEN(side, RC_TRY((launch_rows_inv<N_>(c, a.n, a.T, a.prep, a.skip, a.partial, a.conv_out,
                                                               a.var_out, a.st))));
            }
            return PSFMC_OK;
        case SZ_RASTER_SUMS:
            DISPATCH_LEN(side, RC_TRY((launch_raster_sums<N_>(c, a.n, a.prep, a.groups, a.group_size, a.st,
                                                              a.per_field, a.field))));
            return PSFMC_OK;
        case SZ_COL_ENGINE:
            DISPATCH_LEN(side, *a.code = col_engine_code<N_>(c));
            return PSFMC_OK;
    }
    return fail(PSFMC_EINVAL, "unknown size operation %d", op);
}

#define PSFMC_PART_FN_(k) psfmc_size_call_part##k
#define PSFMC_PART_FN(k) PSFMC_PART_FN_(k)
int psfmc_size_call_part0(int op, int side, void* args);
#if PSFMC_NPARTS > 1
int psfmc_size_call_part1(int op, int side, void* args);
int psfmc_size_call_part2(int op, int side, void* args);
int psfmc_size_call_part3(int op, int side, void* args);
#endif
int PSFMC_PART_FN(PSFMC_PART)(int op, int side, void* args) { return size_call_here(op, side, *static_cast<SizeCall*>(args)); }

#if PSFMC_PART == 0
static int size_call(int op, int side, SizeCall& a) {
#if PSFMC_NPARTS == 1
    return size_call_here(op, side, a);
#else
    int (*const parts[])(int, int, void*) = {psfmc_size_call_part0, psfmc_size_call_part1, psfmc_size_call_part2,
                                             psfmc_size_call_part3};
    for (auto fn : parts) {
        const int rc = fn(op, side, &a);
        if (rc != PSFMC_NOT_MINE) return rc;
    }
    return fail(PSFMC_EINVAL, "fused backend: side %d is not one of " PSFMC_FUSED_SIDES, side);
#endif
}

// rows3_wanted: the shape of the three-stage row family where the side has both; *family: row_families' bits
static int row_shape_for(int nx, RowShape* out, bool rows3_wanted = false, int* family = nullptr) {
    SizeCall a;
    int code = 0;
    a.shape = out;
    a.rows3 = rows3_wanted;
    a.code = &code;
    const int rc = size_call(SZ_ROW_SHAPE, nx, a);
    if (family) *family = code;
    return rc;
}

static int flush_linear_sums(psfmc_ctx* c);   // posterior-image sums: see psfmc_reset_accumulated

// ---------------------------------------------------------------------------
// hipFFT plans, cached per batch size (a half-ensemble call and a full-ensemble
// call use different sizes)
// ---------------------------------------------------------------------------
static int use_plans(psfmc_ctx* c, int batch) {
    auto it = c->plans.find(batch);
    if (it == c->plans.end()) {
        if (c->plans.size() >= 8) {
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

static void free_work(psfmc_ctx* c) {
    void** bufs[] = {(void**)&c->d_real, (void**)&c->d_spec, (void**)&c->d_Ts[0], (void**)&c->d_Ts[1],
                     (void**)&c->d_Ts[2], (void**)&c->d_Ts[3]};
    for (void** p : bufs)
        if (*p) {
            (void)hipFree(*p);
            *p = nullptr;
        }
    c->d_T = nullptr;
}

static int alloc_work(psfmc_ctx* c) {
    free_work(c);
    if (c->backend == PSFMC_BACKEND_HIPFFT) {
        const size_t nimg = (size_t)2 * c->chunk;
        HIP_TRY(hipMalloc(&c->d_real, nimg * c->S * sizeof(double)));
        HIP_TRY(hipMalloc(&c->d_spec, nimg * c->F * sizeof(double2)));
        return use_plans(c, (int)nimg);
    }
    // buffer 0 also serves batches of up to two chunks that run as ONE pass (run_pipeline)
    c->single_cap = 2 * c->chunk < c->max_walkers ? 2 * c->chunk : c->max_walkers;
    if (c->single_cap < c->chunk) c->single_cap = c->chunk;
    for (int i = 0; i < c->n_streams; ++i)
        HIP_TRY(hipMalloc(&c->d_Ts[i], (size_t)(i ? c->chunk : c->single_cap) * 2 * c->nxh * c->nyp *
                                           (c->t_f32 ? sizeof(cf) : sizeof(cd))));
    c->d_T = c->d_Ts[0];
    return PSFMC_OK;
}

// Walkers per internal pass of the fused path: the transposed half-spectra of one pass
// (two passes in flight, together just under the 256 MiB Infinity Cache: measured
// best at 256^2 -- 104..120 walkers; 136 and more fall off -- see DESIGN.md).
// Never rounded UP past that budget: 32 walkers at 512^2 (2 x 135 MB) ran 6 % slower than
// 24, 16 at 1024^2 8 % slower than 6 (gpurun_out r2i sweep).
static int fused_pass_walkers(const psfmc_ctx* c) {
    const double per_walker = 2.0 * c->nxh * c->nyp * (c->t_f32 ? 8.0 : 16.0);
    // Up to 1024^2: 112 MiB per pass, two passes in flight (rounds 1-3, swept per size).  Above, the kernel spectra
    // and the packed field pixels -- each the size of one walker's T -- are a third of the Infinity Cache and count:
    // two passes + the two shared arrays inside 224 MiB (round 4, same box: 1536^2 2 walkers per pass 21.1 k
    // evals/s, 3: 18.2 k, 4: 17.3 k; 2048^2 1: 7.86 k, 2: 7.60 k; 1152^2 4: 28.4 k, 6: 25.4 k)
    const double t64 = 2.0 * c->nxh * c->nyp * 16.0;
    const int fit = t64 <= 17.0e6 ? (int)(112.0 * 1048576.0 / per_walker)
                                  : (int)((224.0 * 1048576.0 - 2.0 * t64) / (2.0 * per_walker));
    int chunk = fit >= 64 ? ((fit + 4) & ~7) : fit >= 16 ? (fit & ~7) : fit >= 4 ? (fit & ~1) : fit;
    // (sides above 1024: 2 walkers per pass at 1536^2, ONE at 2048^2 -- 67 MB of T each, two passes in flight)
    return chunk < 1 ? 1 : chunk;
}

static bool fused_side(int n) {
    static const int sides[] = {64,84,88,96,98,100,104,110,112,120,126,128,130,132,140,144,150,156,160,168,176,180,192,196,200,208,210,220,224,240,250,252,256,260,264,280,286,288,294,300,308,312,320,330,336,350,352,360,364,384,390,392,400,416,420,440,448,480,484,500,504,512,520,528,560,572,576,600,616,624,630,640,650,660,672,676,700,704,720,728,768,780,784,800,832,840,896,900,960,1024,1152,1280,1536,2048};
    for (int v : sides)
        if (v == n) return true;
    return false;
}


// per-kernel timing: bracket a launch with events on its own stream
enum { PROF_ROWS_FWD = 0, PROF_COLS = 1, PROF_ROWS_INV = 2 };
struct ProfScope {
    psfmc_ctx* c; int kind; hipStream_t st; hipEvent_t a = nullptr, b = nullptr;
    ProfScope(psfmc_ctx* c_, int kind_, hipStream_t st_) : c(c_), kind(kind_), st(st_) {
        if (c->profile && hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess)
            (void)hipEventRecord(a, st);
    }
    ~ProfScope() {
        if (c->profile && a && b) {
            (void)hipEventRecord(b, st);
            c->prof_pending.push_back({kind, a, b});
        }
    }
};

static void prof_collect(psfmc_ctx* c) {
    for (auto& r : c->prof_pending) {
        float ms = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            c->prof_ms[r.kind] += ms;
            c->prof_n[r.kind] += 1;
        }
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    c->prof_pending.clear();
}

// rasterise + both convolutions of `n` walkers; results stay in d_T (spectral
// rows after the column pass).  rows_inv is launched by the caller.
static int fused_rows_fwd(psfmc_ctx* c, int n, cd* Tbuf, const double* prep, const uint8_t* skip,
                          int ps_only, double* raw_out, hipStream_t st) {
    ProfScope ps(c, PROF_ROWS_FWD, st);
    SizeCall a;
    a.c = c; a.n = n; a.prep = prep; a.skip = skip; a.T = Tbuf; a.ps_only = ps_only; a.raw_out = raw_out; a.st = st;
    a.f32 = c->t_f32;
    return size_call(SZ_ROWS_FWD, c->nx, a);
}

static int fused_cols(psfmc_ctx* c, int n, cd* Tbuf, const double* prep, const uint8_t* skip, hipStream_t st) {
    ProfScope ps(c, PROF_COLS, st);
    SizeCall a;
    a.c = c; a.n = n; a.prep = prep; a.skip = skip; a.T = Tbuf; a.st = st; a.f32 = c->t_f32;
    return size_call(SZ_COLS, c->ny, a);
}

static int fused_forward(psfmc_ctx* c, int n, cd* Tbuf, const double* prep, const uint8_t* skip,
                         int ps_only, double* raw_out, hipStream_t st) {
    RC_TRY(fused_rows_fwd(c, n, Tbuf, prep, skip, ps_only, raw_out, st));
    return fused_cols(c, n, Tbuf, prep, skip, st);
}

static int fused_inverse(psfmc_ctx* c, int n, const cd* Tbuf, const double* prep, const uint8_t* skip,
                         double* partial, double* conv_out, double* var_out, hipStream_t st) {
    ProfScope ps(c, PROF_ROWS_INV, st);
    SizeCall a;
    a.c = c; a.n = n; a.prep = prep; a.skip = skip; a.T = const_cast<cd*>(Tbuf); a.partial = partial;
    a.conv_out = conv_out; a.var_out = var_out; a.st = st; a.f32 = c->t_f32;
    return size_call(SZ_ROWS_INV, c->nx, a);
}

static std::vector<cd> twiddle_table(int n) {
    std::vector<cd> t(n);
    const long double two_pi = 6.283185307179586476925286766559005768L;
    for (int k = 0; k < n; ++k) {
        const long double a = two_pi * (long double)k / (long double)n;
        t[k] = cd{(double)cosl(a), (double)-sinl(a)};
    }
    return t;
}

// ---------------------------------------------------------------------------
// F0: kernel spectra on the device (replaces utils.py:9-22, :126-133)
// d_canvas: [2*n_psf][S], image 2p = padded PSF p, image 2p+1 = its variance map
// ---------------------------------------------------------------------------
static int spectra_hipfft(psfmc_ctx* c, const double* d_canvas) {
    HIP_TRY(hipMalloc(&c->d_pspec, (size_t)c->n_psf * c->F * sizeof(double2)));
    HIP_TRY(hipMalloc(&c->d_vspec, (size_t)c->n_psf * c->F * sizeof(double2)));
    hipfftHandle plan;
    int n[2] = {c->ny, c->nx};
    FFT_TRY(hipfftPlanMany(&plan, 2, n, nullptr, 1, 0, nullptr, 1, 0, HIPFFT_D2Z, 1));
    int rc = PSFMC_OK;
    hipfftSetStream(plan, c->stream);
    for (int p = 0; p < c->n_psf && rc == PSFMC_OK; ++p) {
        double* a = const_cast<double*>(d_canvas) + (size_t)(2 * p) * c->S;
        if (hipfftExecD2Z(plan, a, (hipfftDoubleComplex*)(c->d_pspec + (size_t)p * c->F)) != HIPFFT_SUCCESS ||
            hipfftExecD2Z(plan, a + c->S, (hipfftDoubleComplex*)(c->d_vspec + (size_t)p * c->F)) !=
                HIPFFT_SUCCESS)
            rc = fail(PSFMC_EHIP, "hipfftExecD2Z (PSF spectra) failed");
    }
    (void)hipStreamSynchronize(c->stream);
    hipfftDestroy(plan);
    return rc;
}

static int spectra_fused(psfmc_ctx* c, const double* d_canvas) {
    const size_t n_el = (size_t)c->n_psf * 2 * c->nxh * c->ny;
    HIP_TRY(hipMalloc(&c->d_Kraw, (size_t)c->n_psf * 2 * c->nxh * c->nyp * sizeof(cd)));
    HIP_TRY(hipMemsetAsync(c->d_Kraw, 0, (size_t)c->n_psf * 2 * c->nxh * c->nyp * sizeof(cd), c->stream));
    HIP_TRY(hipMalloc(&c->d_Kt, n_el * sizeof(cd)));
    {
        SizeCall a;
        a.c = c; a.n = c->n_psf; a.T = c->d_Kraw; a.img = d_canvas; a.img_scale = c->d_rho; a.st = c->stream;
        a.from_image = true;
        RC_TRY(size_call(SZ_ROWS_FWD, c->nx, a));
        a.convolve = false;
        RC_TRY(size_call(SZ_COLS, c->ny, a));
    }
    hipLaunchKernelGGL(k_scale_kernel_spectrum, dim3(256), dim3(256), 0, c->stream, c->d_Kraw, c->d_Kt,
                       (int)n_el, c->ny, c->nxh, c->rg_log2, 0.25 / (double)c->S);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PSFMC_OK;
}

// ---------------------------------------------------------------------------
// API
// ---------------------------------------------------------------------------
extern "C" int psfmc_abi_version(void) { return 1; }

extern "C" const char* psfmc_last_error(void) { return g_err.c_str(); }

static int ctx_init(psfmc_ctx* c, const double* sci, const double* obs_var, const uint8_t* bad_px,
                    int psf_ny, int psf_nx, const double* psf, const double* psf_var) {
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    for (int i = 0; i < psfmc_ctx::kMaxStreams; ++i) HIP_TRY(hipEventCreateWithFlags(&c->ev_stagger[i], hipEventDisableTiming));
    for (int k = 0; k < 3; ++k)
        for (int i = 0; i < 2; ++i) HIP_TRY(hipEventCreateWithFlags(&c->ev_excl[k][i], hipEventDisableTiming));
    for (int i = 1; i < psfmc_ctx::kMaxStreams; ++i) {
        HIP_TRY(hipStreamCreateWithFlags(&c->side[i], hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&c->ev_join[i], hipEventDisableTiming));
    }
    const size_t all_px = (size_t)c->n_fields * c->S;             // [field][ny][nx]
    HIP_TRY(hipMalloc(&c->d_sci, all_px * sizeof(double)));
    HIP_TRY(hipMalloc(&c->d_var, all_px * sizeof(double)));
    HIP_TRY(hipMalloc(&c->d_bad, all_px));
    HIP_TRY(hipMemcpy(c->d_sci, sci, all_px * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->d_var, obs_var, all_px * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->d_bad, bad_px, all_px, hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc(&c->d_rows, (size_t)c->max_walkers * c->rlen * sizeof(double)));
    HIP_TRY(hipMalloc(&c->d_prep, (size_t)c->max_walkers * c->plen * sizeof(double)));
    HIP_TRY(hipMalloc(&c->d_like, (size_t)c->max_walkers * sizeof(double)));
    HIP_TRY(hipMalloc(&c->d_skip, (size_t)c->max_walkers));
    HIP_TRY(hipMalloc(&c->d_partial, (size_t)c->max_walkers * c->nblk * sizeof(double)));

    if (c->backend == PSFMC_BACKEND_FUSED) {
        const std::vector<cd> tx = twiddle_table(c->nx), ty = twiddle_table(c->ny);
        HIP_TRY(hipMalloc(&c->d_twx, tx.size() * sizeof(cd)));
        HIP_TRY(hipMalloc(&c->d_twy, ty.size() * sizeof(cd)));
        HIP_TRY(hipMemcpy(c->d_twx, tx.data(), tx.size() * sizeof(cd), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(c->d_twy, ty.data(), ty.size() * sizeof(cd), hipMemcpyHostToDevice));
        // rho[p] = 2^-round(log2(sum of the variance map)): brings the variance
        // channel of the packed complex transforms up to the model channel's scale
        std::vector<double> rho(c->n_psf, 1.0);
        const size_t small_n = (size_t)psf_ny * psf_nx;
        for (int p = 0; p < c->n_psf; ++p) {
            double tot = 0.0;
            for (size_t i = 0; i < small_n; ++i) tot += psf_var[(size_t)p * small_n + i];
            if (tot > 0.0 && std::isfinite(tot)) {
                int e;
                (void)frexp(tot, &e);
                rho[p] = ldexp(1.0, -e);
            }
        }
        HIP_TRY(hipMalloc(&c->d_rho, c->n_psf * sizeof(double)));
        HIP_TRY(hipMemcpy(c->d_rho, rho.data(), c->n_psf * sizeof(double), hipMemcpyHostToDevice));
        size_t field_len = 0;
        {
            SizeCall a;
            a.len = &field_len; a.ny = c->ny; a.rows3 = c->rows3_inv;
            RC_TRY(size_call(SZ_FIELD_LEN, c->nx, a));
        }
        c->field_len = field_len;
        HIP_TRY(hipMalloc(&c->d_field, (size_t)c->n_fields * field_len * sizeof(FieldPx)));
        for (int f = 0; f < c->n_fields; ++f) {
            SizeCall a;
            a.c = c; a.field = f;
            RC_TRY(size_call(SZ_PACK_FIELD, c->nx, a));
        }
    }

    // centre-padded canvases, interleaved (psf0, var0, psf1, var1, ...)
    const size_t small = (size_t)psf_ny * psf_nx;
    std::vector<double> inter((size_t)2 * c->n_psf * small);
    for (int p = 0; p < c->n_psf; ++p) {
        memcpy(&inter[(size_t)(2 * p) * small], psf + (size_t)p * small, small * sizeof(double));
        memcpy(&inter[(size_t)(2 * p + 1) * small], psf_var + (size_t)p * small, small * sizeof(double));
    }
    double *d_small = nullptr, *d_canvas = nullptr;
    HIP_TRY(hipMalloc(&d_small, inter.size() * sizeof(double)));
    int rc = PSFMC_OK;
    if (hipMalloc(&d_canvas, (size_t)2 * c->n_psf * c->S * sizeof(double)) != hipSuccess)
        rc = fail(PSFMC_ENOMEM, "hipMalloc(canvas) failed");
    if (rc == PSFMC_OK &&
        hipMemcpy(d_small, inter.data(), inter.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess)
        rc = fail(PSFMC_EHIP, "hipMemcpy(psf) failed");
    if (rc == PSFMC_OK) {
        hipLaunchKernelGGL(k_pad, dim3(256), dim3(256), 0, c->stream, d_small, d_canvas, 2 * c->n_psf,
                           psf_ny, psf_nx, c->ny, c->nx);
        rc = c->backend == PSFMC_BACKEND_FUSED ? spectra_fused(c, d_canvas) : spectra_hipfft(c, d_canvas);
    }
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(d_small);
    if (d_canvas) (void)hipFree(d_canvas);
    RC_TRY(rc);
    RC_TRY(alloc_work(c));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PSFMC_OK;
}

// the built sides in ascending order
static const int kFusedSides[] = {64,84,88,96,98,100,104,110,112,120,126,128,130,132,140,144,150,156,160,168,176,180,192,196,200,208,210,220,224,240,250,252,256,260,264,280,286,288,294,300,308,312,320,330,336,350,352,360,364,384,390,392,400,416,420,440,448,480,484,500,504,512,520,528,560,572,576,600,616,624,630,640,650,660,672,676,700,704,720,728,768,780,784,800,832,840,896,900,960,1024,1152,1280,1536,2048};

// One axis of an image whose side `l` the transforms are not built for: the smallest built side
// m >= l + pk - 1 (pk the PSF's side on that axis), the margin a in front of the image and the extent e of
// the filled transform pixels (psfmc_device.h WrapDesc).  The kernel's origin inside the centre-padded
// PSF, c = m/2 - (m - pk)/2 (utils.py:9-22 + the ifftshift of :32), is the same for every even m, so
// outputs [a, a + l) of the length-m circular convolution are those of the length-l one when a = pk - 1 - c.
static void embed_axis_at(int l, int pk, int m, int* a, int* e) {
    const int c = m / 2 - (m - pk) / 2;
    *a = pk - 1 - c;
    *e = l + pk - 1;
}

// The transform shape of an image with unbuilt sides.  Every built side >= l + pk - 1 would do; the kernels of
// the built sides differ by up to 2x per pixel (radix mix, lanes per transform, registers), so the smallest
// is often not the cheapest: the pair (my, mx) with the least ny nx (rows(nx) + cols(ny)) of the measured table
// psfmc_side_costs.h wins, if it beats the smallest pair by 4 % (the table's noise).  A built axis stays as it is.
static bool choose_embedding(int ly, int lx, int pky, int pkx, int* my, int* mx) {
    auto cost_of = [](int side) -> const SideCost* {
        for (const SideCost& sc : kSideCosts)
            if (sc.side == side) return &sc;
        return nullptr;
    };
    const bool fix_y = fused_side(ly), fix_x = fused_side(lx);
    const int need_y = fix_y ? ly : ly + pky - 1, need_x = fix_x ? lx : lx + pkx - 1;
    int small_y = 0, small_x = 0;
    for (int v : kFusedSides) {
        if (!small_y && v >= need_y) small_y = v;
        if (!small_x && v >= need_x) small_x = v;
    }
    if (!small_y || !small_x) return false;
    *my = small_y; *mx = small_x;
    const SideCost *sy0 = cost_of(small_y), *sx0 = cost_of(small_x);
    if (!sy0 || !sx0) return true;
    const double base = (double)small_y * small_x * (sx0->rows_ps + sy0->cols_ps);
    double best = base * 0.96;
    for (int vy : kFusedSides) {
        if (fix_y ? vy != ly : vy < need_y) continue;
        const SideCost* sy = cost_of(vy);
        if (!sy) continue;
        for (int vx : kFusedSides) {
            if (fix_x ? vx != lx : vx < need_x) continue;
            const SideCost* sx = cost_of(vx);
            if (!sx) continue;
            const double cst = (double)vy * vx * (sx->rows_ps + sy->cols_ps);
            if (cst < best) { best = cst; *my = vy; *mx = vx; }
        }
    }
    return true;
}

static int ctx_create_impl(psfmc_ctx** out, int device, int ny, int nx, int n_fields, const double* sci,
                           const double* obs_var, const uint8_t* bad_px, int n_psf,
                           int psf_ny, int psf_nx, const double* psf, const double* psf_var,
                           int n_ps, int n_sersic, int max_walkers, int backend) {
    if (!out) return fail(PSFMC_EINVAL, "out is NULL");
    if (n_fields < 1 || n_fields > 4096) return fail(PSFMC_EINVAL, "n_fields out of range");
    if (n_fields > 1 && backend != PSFMC_BACKEND_FUSED)
        return fail(PSFMC_EINVAL, "several fields per context need the fused back end");
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
    int row_tiles = 0;
    bool rows3_fwd = false, rows3_inv = false;
    RowShape rs{};
    const int ly = ny, lx = nx;                       // the image's own sides
    WrapDesc wrap{0, 0, 0, 0, 0, 0};
    bool embed = false;
    std::vector<double> pad_sci, pad_var;
    std::vector<uint8_t> pad_bad;
    if (backend == PSFMC_BACKEND_FUSED) {
        if (!fused_side(ny) || !fused_side(nx)) {
            // embed the axes the transforms are not built for (a built axis stays as it is: a = 0, e = l = m)
            embed = true;
            wrap = WrapDesc{lx, 0, lx, ly, 0, ly};
            if (!choose_embedding(ly, lx, psf_ny, psf_nx, &ny, &nx))
                return fail(PSFMC_EINVAL, "fused backend: image %d x %d + PSF %d x %d - 1 exceeds the largest built "
                            "side (2048)", ly, lx, psf_ny, psf_nx);
            if (nx != lx) embed_axis_at(lx, psf_nx, nx, &wrap.ax, &wrap.ex);
            if (ny != ly) embed_axis_at(ly, psf_ny, ny, &wrap.ay, &wrap.ey);
            // the field arrays in transform coordinates: the image at (ay, ax), every other pixel excluded
            const size_t S_t = (size_t)ny * nx, S_l = (size_t)ly * lx;
            pad_sci.assign((size_t)n_fields * S_t, 0.0);
            pad_var.assign((size_t)n_fields * S_t, 1.0);
            pad_bad.assign((size_t)n_fields * S_t, 1);
            for (int f = 0; f < n_fields; ++f)
                for (int y = 0; y < ly; ++y) {
                    const size_t dst = (size_t)f * S_t + (size_t)(y + wrap.ay) * nx + wrap.ax;
                    const size_t src = (size_t)f * S_l + (size_t)y * lx;
                    memcpy(&pad_sci[dst], sci + src, (size_t)lx * sizeof(double));
                    memcpy(&pad_var[dst], obs_var + src, (size_t)lx * sizeof(double));
                    memcpy(&pad_bad[dst], bad_px + src, (size_t)lx);
                }
            sci = pad_sci.data(); obs_var = pad_var.data(); bad_px = pad_bad.data();
        }
        const char* env3 = getenv("PSFMC_ROWS3");
        const int force = env3 ? (atoi(env3) != 0 ? 1 : 0) : -1;          // -1: the defaults
        int fam = 0;
        RC_TRY(row_shape_for(nx, &rs, false, &fam));                       // (the two-stage shape where there is one)
        const bool two = fam & 1;
        rows3_inv = !two || ((fam & 2) && (force == 1 || (force < 0 && (fam & 8))));
        rows3_fwd = !two || ((fam & 4) && force == 1);
        // a power-of-two side takes both kernels of a family (their layouts differ: row groups of 2 against 4)
        if (two && rs.plain && rows3_inv != rows3_fwd) rows3_inv = rows3_fwd = false;
        row_tiles = rows3_inv ? ny : (ny + rs.rg - 1) / rs.rg;             // chi^2 partial sums per walker
    }

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
    c->ly = ly; c->lx = lx; c->embed = embed; c->wrap = wrap;
    c->n_fields = n_fields; c->n_psf_field = n_psf;
    c->acc_more.assign(n_fields - 1, 0);
    c->n_psf = n_fields * n_psf; c->n_ps = n_ps; c->n_sersic = n_sersic;
    c->max_walkers = max_walkers; c->backend = backend;
    c->rlen = row_len(n_ps, n_sersic);
    c->plen = prep_len(n_ps, n_sersic);
    // the rasteriser's form of (rho^2)^p for this context's kernels (psfmc_device.h pow_tabs_side)
    c->use_pow_tabs = pow_tabs_side(nx);
    c->nyp = ny;
    if (backend == PSFMC_BACKEND_FUSED) {
        c->nblk = row_tiles;
        c->rows3_fwd = rows3_fwd; c->rows3_inv = rows3_inv;
        // the power-of-two row kernels run without row guards: whole workgroups of rows only;
        // any other ny takes the guarded code path of the same shape (layout groups of 4 rows)
        c->row_fast = !rows3_fwd && !rows3_inv && rs.plain && ny % (rs.rg * rs.fast_waves) == 0;
        c->rg_log2 = c->row_fast ? rs.fast_rg_log2 : (rows3_fwd || rows3_inv) ? rows3_rg_log2(nx) : 2;   // (rows3: 2; 0 above 1024)
        c->nyp = t_col_len(ny, c->rg_log2);
        RowShape cs{};
        RC_TRY(row_shape_for(ny, &cs));
        c->plain_shape = !rows3_fwd && !rows3_inv && rs.plain && cs.plain;
        c->cols_grid = prop.multiProcessorCount * 2;
        c->chunk = fused_pass_walkers(c);
        // measured (gpurun_out/stag*_quick.txt, same-box A/B): 64^2 +2.7 %, 128^2 +1.7 %, 256^2 +4.6 %
        // (+3.8 % with two Sersics); 384^2 -0.4 %, 512^2 -1.3...-2.5 %, 1024^2 0, 200^2 -1.6 %, 300^2 -7.7 %,
        // 400^2 -10.6 %: on for the small power-of-two shapes only
        c->stagger = (c->plain_shape && c->row_fast && nx <= 256 && ny <= 256) ? 1 : 0;
    } else {
        c->nblk = (c->S + 1023) / 1024;
        if (c->nblk > 64) c->nblk = 64;
        // keep the work space of the hipFFT path <= ~6 GiB
        const double per_walker = 2.0 * (c->S * 8.0 + c->F * 16.0);
        int chunk = (int)(6.0 * 1073741824.0 / per_walker);
        c->chunk = chunk < 1 ? 1 : chunk;
    }
    if (c->chunk > max_walkers) c->chunk = max_walkers;

    int rc = ctx_init(c, sci, obs_var, bad_px, psf_ny, psf_nx, psf, psf_var);
    if (rc != PSFMC_OK) {
        const std::string keep = g_err;
        psfmc_ctx_destroy(c);
        g_err = keep;
        return rc;
    }
    *out = c;
    return PSFMC_OK;
}

extern "C" int psfmc_ctx_create(psfmc_ctx** out, int device, int ny, int nx, const double* sci,
                                const double* obs_var, const uint8_t* bad_px, int n_psf,
                                int psf_ny, int psf_nx, const double* psf, const double* psf_var,
                                int n_ps, int n_sersic, int max_walkers, int backend) {
    return ctx_create_impl(out, device, ny, nx, 1, sci, obs_var, bad_px, n_psf, psf_ny, psf_nx, psf, psf_var, n_ps,
                           n_sersic, max_walkers, backend);
}

// Several observed fields of one shape in ONE context (fused back end): their walkers share the
// batches of psfmc_eval_theta_device_fields, so many small ensembles run at the rate of one large one
// (BASELINE config 5: independent fields x 256 walkers each).  sci / obs_var / bad_px: [n_fields][ny][nx];
// psf / psf_var: [n_fields][n_psf][psf_ny][psf_nx]; the same component counts for every field.
extern "C" int psfmc_ctx_create_fields(psfmc_ctx** out, int device, int ny, int nx, int n_fields,
                                       const double* sci, const double* obs_var, const uint8_t* bad_px,
                                       int n_psf, int psf_ny, int psf_nx, const double* psf,
                                       const double* psf_var, int n_ps, int n_sersic, int max_walkers) {
    return ctx_create_impl(out, device, ny, nx, n_fields, sci, obs_var, bad_px, n_psf, psf_ny, psf_nx, psf, psf_var,
                           n_ps, n_sersic, max_walkers, PSFMC_BACKEND_FUSED);
}

extern "C" int psfmc_ctx_destroy(psfmc_ctx* c) {
    if (!c) return PSFMC_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    prof_collect(c);
    for (auto& kv : c->plans) {
        hipfftDestroy(kv.second.first);
        hipfftDestroy(kv.second.second);
    }
    void* bufs[] = {c->d_sci,  c->d_var,  c->d_bad,     c->d_pspec, c->d_vspec, c->d_rows, c->d_prep,
                    c->d_like, c->d_skip, c->d_partial, c->d_real,  c->d_spec,  c->d_Ts[0], c->d_Kraw,
                    c->d_Kt,   c->d_twx,  c->d_twy,     c->d_img0,  c->d_img1, c->d_rho,   c->d_field, c->d_Ts[1], c->d_acc, c->d_lin, c->d_linpart,
                    c->d_layout_blob, c->d_theta, c->d_extra, c->d_lnprior, c->d_rawstage, c->d_field_layouts,
                    c->d_Ts[2], c->d_Ts[3], c->stretch.pos, c->stretch.lnp, c->stretch.q, c->stretch.newlnp,
                    c->stretch.rand, c->stretch.chain, c->stretch.lnchain, c->stretch.partner, c->stretch.iter,
                    c->stretch.nacc, c->stretch.accflag};
    for (void* p : bufs)
        if (p) (void)hipFree(p);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    for (int i = 1; i < psfmc_ctx::kMaxStreams; ++i) {
        if (c->side[i]) {
            (void)hipStreamSynchronize(c->side[i]);
            (void)hipStreamDestroy(c->side[i]);
        }
        if (c->ev_join[i]) (void)hipEventDestroy(c->ev_join[i]);
    }
    for (void* b : c->more_blobs)
        if (b) (void)hipFree(b);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    for (int i = 0; i < psfmc_ctx::kMaxStreams; ++i)
        if (c->ev_stagger[i]) (void)hipEventDestroy(c->ev_stagger[i]);
    for (int k = 0; k < 3; ++k)
        for (int i = 0; i < 2; ++i)
            if (c->ev_excl[k][i]) (void)hipEventDestroy(c->ev_excl[k][i]);
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
        HIP_TRY(hipDeviceSynchronize());
        c->chunk = v;
        if (c->d_img0) { (void)hipFree(c->d_img0); c->d_img0 = nullptr; }
        if (c->d_img1) { (void)hipFree(c->d_img1); c->d_img1 = nullptr; }
        if (c->d_rawstage) { (void)hipFree(c->d_rawstage); c->d_rawstage = nullptr; }
        c->img_cap = 0;
        return alloc_work(c);
    }
    if (!strcmp(key, "storage_f32")) {
        // keep the intermediate half-spectra as complex64 (arithmetic stays fp64): half the
        // traffic of every kernel, ~1e-7 relative error in the log-posterior -- the class of the
        // reference's own float32 raw model (psfMC/models.py:249), not an fp64 result
        const bool on = value != 0;
        if (on && (c->backend != PSFMC_BACKEND_FUSED || !c->plain_shape || !c->row_fast || c->embed))
            return fail(PSFMC_EINVAL, "storage_f32 needs the fused back end and power-of-two sides");
        HIP_TRY(hipSetDevice(c->device));
        HIP_TRY(hipDeviceSynchronize());
        c->t_f32 = on;
        c->chunk = fused_pass_walkers(c);
        if (c->chunk > c->max_walkers) c->chunk = c->max_walkers;
        if (c->d_img0) { (void)hipFree(c->d_img0); c->d_img0 = nullptr; }
        if (c->d_img1) { (void)hipFree(c->d_img1); c->d_img1 = nullptr; }
        if (c->d_rawstage) { (void)hipFree(c->d_rawstage); c->d_rawstage = nullptr; }
        c->img_cap = 0;
        return alloc_work(c);
    }
    if (!strcmp(key, "cols_grid")) {
        if (value < 1) return fail(PSFMC_EINVAL, "cols_grid must be >= 1");
        c->cols_grid = (int)value;
        return PSFMC_OK;
    }
    if (!strcmp(key, "stagger")) {
        c->stagger = (int)value;
        return PSFMC_OK;
    }
    if (!strcmp(key, "exclusive")) {
        c->exclusive = (int)value & 7;
        return PSFMC_OK;
    }
    if (!strcmp(key, "linear_accumulation")) {
        HIP_TRY(hipSetDevice(c->device));
        RC_TRY(flush_linear_sums(c));
        c->linear_acc = value != 0;
        return PSFMC_OK;
    }
    if (!strcmp(key, "cols3")) {
        c->cols3 = (int)value;
        return PSFMC_OK;
    }
    if (!strcmp(key, "speculate")) {
        c->speculate = (int)value;          // 0 never, n > 0 up to n walkers per half, -1 the default rule
        return PSFMC_OK;
    }
    if (!strcmp(key, "graph")) {
        c->use_graph = value != 0;
        return PSFMC_OK;
    }
    if (!strcmp(key, "min_split")) {
        c->min_split = value < 1 ? 1 : (int)value;
        return PSFMC_OK;
    }
    if (!strcmp(key, "profile")) {
        HIP_TRY(hipSetDevice(c->device));
        HIP_TRY(hipDeviceSynchronize());
        prof_collect(c);
        c->profile = value != 0;
        for (int i = 0; i < 3; ++i) { c->prof_ms[i] = 0; c->prof_n[i] = 0; }
        return PSFMC_OK;
    }
    if (!strcmp(key, "streams")) {
        if (value < 1 || value > psfmc_ctx::kMaxStreams) return fail(PSFMC_EINVAL, "streams must be 1..4");
        HIP_TRY(hipSetDevice(c->device));
        HIP_TRY(hipDeviceSynchronize());
        c->n_streams = (int)value;
        return alloc_work(c);
    }
    return fail(PSFMC_EINVAL, "unknown option '%s'", key);
}

extern "C" double psfmc_get_option(const psfmc_ctx* cc, const char* key) {
    if (!cc || !key) return NAN;
    psfmc_ctx* c = const_cast<psfmc_ctx*>(cc);
    static const char* kinds[3] = {"rows_fwd", "cols", "rows_inv"};
    for (int i = 0; i < 3; ++i) {
        char name[64];
        snprintf(name, sizeof name, "prof_ms_%s", kinds[i]);
        if (!strcmp(key, name)) { prof_collect(c); return c->prof_ms[i]; }
        snprintf(name, sizeof name, "prof_n_%s", kinds[i]);
        if (!strcmp(key, name)) { prof_collect(c); return (double)c->prof_n[i]; }
    }
    if (!strcmp(key, "row_group")) return 1 << c->rg_log2;
    if (!strcmp(key, "rows3")) return (c->rows3_fwd ? 1.0 : 0.0) + (c->rows3_inv ? 2.0 : 0.0);   // bit 0 forward, bit 1 inverse
    if (!strcmp(key, "column_engine")) {       // the column kernel this context launches NOW: 0 k_cols, 1 k_cols3, 2 k_cols3g, 3 k_cols3f
        if (c->backend != PSFMC_BACKEND_FUSED) return NAN;
        int code = 0;
        SizeCall a;
        a.c = c; a.code = &code;
        return size_call(SZ_COL_ENGINE, c->ny, a) == PSFMC_OK ? (double)code : NAN;
    }
    if (!strcmp(key, "speculate")) return c->speculate;
    if (!strcmp(key, "pow_tabs")) return c->use_pow_tabs ? 1.0 : 0.0;
    if (!strcmp(key, "speculated_runs")) return (double)c->speculated_runs;
    if (!strcmp(key, "transform_ny")) return c->ny;        // the transform shape (the image's own, or the one it is embedded in)
    if (!strcmp(key, "transform_nx")) return c->nx;
    if (!strcmp(key, "partials_per_walker")) return c->nblk;
    if (!strcmp(key, "storage_f32")) return c->t_f32 ? 1.0 : 0.0;
    if (!strcmp(key, "graph_launches")) return (double)c->graph_launches;
    if (!strcmp(key, "chunk_walkers")) return c->chunk;
    if (!strcmp(key, "backend")) return c->backend;
    if (!strcmp(key, "max_walkers")) return c->max_walkers;
    if (!strcmp(key, "cols_grid")) return c->cols_grid;
    if (!strcmp(key, "streams")) return c->n_streams;
    if (!strcmp(key, "stagger")) return c->stagger;
    if (!strcmp(key, "exclusive")) return c->exclusive;
    if (!strcmp(key, "linear_accumulation")) return c->linear_acc ? 1.0 : 0.0;
    return NAN;
}

// ---------------------------------------------------------------------------
// hipFFT path: rasterise + convolve one chunk; results in d_real (conv, var)
// ---------------------------------------------------------------------------
static int hipfft_convolve(psfmc_ctx* c, int n, const double* d_prep, const uint8_t* d_skip,
                           hipStream_t st, int ps_only) {
    const size_t lds = (size_t)prep_rec_len(c->n_ps, c->n_sersic) * sizeof(double);
    RC_TRY(use_plans(c, 2 * n));
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

// Walkers per internal pass for a batch of W (fused path).  Pass i runs on stream
// i % n_streams with its own T buffer, so the VALU-bound row kernels of one pass overlap the
// HBM-bound column kernel of its neighbours; passes of equal size (no short tail), an even
// number of them when two run at a time.
// A batch of up to two chunks runs as ONE pass: with only two passes the fork/join
// between the streams costs more than their overlap gains (256^2, MI355X: W = 128
// 148 vs 157 us, W = 224 228 vs 237 us, W = 256 264 vs 269 us, W = 320 353 vs 338 us).
static int pass_size(const psfmc_ctx* c, int W) {
    const bool fused = c->backend == PSFMC_BACKEND_FUSED;
    int chunk = c->chunk;
    if (fused && c->n_streams > 1 && W <= chunk && W / 2 >= c->min_split) chunk = ((W + 1) / 2 + 7) & ~7;
    if (fused && W > chunk && W <= c->single_cap && c->min_split > W / 2) {
        chunk = W;
    } else if (fused && W > chunk) {
        int np = (W + chunk - 1) / chunk;
        if (c->n_streams == 2 && (np & 1)) ++np;
        chunk = (((W + np - 1) / np) + 3) & ~3;
        // the T buffers hold c->chunk walkers: rounding up must never pass that
        if (chunk > c->chunk) chunk = c->chunk;
    } else if (chunk > c->chunk) {
        chunk = c->chunk;
    }
    return chunk;
}

extern "C" int psfmc_pass_size(const psfmc_ctx* c, int W) {
    if (!c) return fail(PSFMC_EINVAL, "ctx is NULL");
    if (W < 1 || W > c->max_walkers) return fail(PSFMC_EINVAL, "W=%d outside [1, max_walkers=%d]", W, c->max_walkers);
    return pass_size(c, W);
}

// the likelihood pipeline over walkers whose prep records are in c->d_prep;
// leaves the chi^2 partial sums in c->d_partial.  Fork/join on events keeps the caller's
// stream semantics.
static int run_pipeline(psfmc_ctx* c, int W, const uint8_t* d_skip, hipStream_t st, int w_off = 0) {
    const bool fused = c->backend == PSFMC_BACKEND_FUSED;
    const int chunk = pass_size(c, W);
    const int npass = (W + chunk - 1) / chunk;
    const int lanes = !fused ? 1 : (npass < c->n_streams ? npass : c->n_streams);
    if (lanes > 1) {
        HIP_TRY(hipEventRecord(c->ev_fork, st));
        for (int i = 1; i < lanes; ++i) HIP_TRY(hipStreamWaitEvent(c->side[i], c->ev_fork, 0));
    }
    int pass = 0;
    for (int w0 = 0; w0 < W; w0 += chunk, ++pass) {
        const int n = W - w0 < chunk ? W - w0 : chunk;
        // walkers [w_off, w_off + W) of the prep / skip / partial arrays
        const double* prep = c->d_prep + (size_t)(w_off + w0) * c->plen;
        const uint8_t* skip = d_skip ? d_skip + w_off + w0 : nullptr;
        double* partial = c->d_partial + (size_t)(w_off + w0) * c->nblk;
        if (fused) {
            const int lane = pass % lanes;
            hipStream_t s = lane ? c->side[lane] : st;
            cd* Tbuf = c->d_Ts[lane];
            if (c->stagger && lanes >= 2 && npass >= 8 && pass < lanes - 1) {
                // start the next lane one forward-row kernel late, so that its VALU-bound kernel
                // meets this lane's memory-bound ones instead of this lane's own copy of it (the idle
                // start costs about half a kernel per batch: worth it from ~8 passes on; 4 passes of
                // 64 walkers ran 249 instead of 235 us with it)
                RC_TRY(fused_rows_fwd(c, n, Tbuf, prep, skip, 0, nullptr, s));
                HIP_TRY(hipEventRecord(c->ev_stagger[pass], s));
                HIP_TRY(hipStreamWaitEvent(c->side[pass + 1], c->ev_stagger[pass], 0));
                RC_TRY(fused_cols(c, n, Tbuf, prep, skip, s));
            } else if (c->exclusive && lanes == 2) {
                // kind k of this pass starts only when kind k of the previous pass (the other lane) has finished
                auto gate = [&](int k) -> int {
                    if ((c->exclusive >> k) & 1) {
                        if (pass > 0) HIP_TRY(hipStreamWaitEvent(s, c->ev_excl[k][(pass - 1) & 1], 0));
                    }
                    return PSFMC_OK;
                };
                auto done = [&](int k) -> int {
                    if ((c->exclusive >> k) & 1) HIP_TRY(hipEventRecord(c->ev_excl[k][pass & 1], s));
                    return PSFMC_OK;
                };
                RC_TRY(gate(0));
                RC_TRY(fused_rows_fwd(c, n, Tbuf, prep, skip, 0, nullptr, s));
                RC_TRY(done(0));
                RC_TRY(gate(1));
                RC_TRY(fused_cols(c, n, Tbuf, prep, skip, s));
                RC_TRY(done(1));
                RC_TRY(gate(2));
                RC_TRY(fused_inverse(c, n, Tbuf, prep, skip, partial, nullptr, nullptr, s));
                RC_TRY(done(2));
                continue;
            } else {
                RC_TRY(fused_forward(c, n, Tbuf, prep, skip, 0, nullptr, s));
            }
            RC_TRY(fused_inverse(c, n, Tbuf, prep, skip, partial, nullptr, nullptr, s));
        } else {
            RC_TRY(hipfft_convolve(c, n, prep, skip, st, 0));
            hipLaunchKernelGGL(k_chi2, dim3(c->nblk, n), dim3(256), 0, st, c->d_real, c->d_sci, c->d_var,
                               c->d_bad, skip, partial, c->S);
        }
    }
    for (int i = 1; i < lanes; ++i) {
        HIP_TRY(hipEventRecord(c->ev_join[i], c->side[i]));
        HIP_TRY(hipStreamWaitEvent(st, c->ev_join[i], 0));
    }
    return PSFMC_OK;
}

// The power tables of every (walker, Sersic component) of a batch, behind the walkers' prep records
// (psfmc_device.h: what the fused rasteriser reads instead of a log2 and an exp2 per pixel).  One wave per
// pair; p = 1 / (2n) is the record's own value, so a table and the record it belongs to always agree.
__global__ void __launch_bounds__(256) k_pow_tables(double* __restrict__ prep, const uint8_t* __restrict__ skip,
                                                    int n_pairs, int n_ps, int n_sersic) {
    const int pair = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (pair >= n_pairs) return;                                   // wave-uniform
    const int w = pair / n_sersic, k = pair - w * n_sersic;
    if (skip && skip[w]) return;
    double* rec = prep + (size_t)w * prep_len(n_ps, n_sersic);
    const double p = rec[kPrepHead + kPrepPs * n_ps + kPrepSersic * k + 7];
    build_pow_table(p, rec + prep_rec_len(n_ps, n_sersic) + (size_t)k * kPowTab, lane);
}
// Small batches run WITHOUT the launch: their forward row waves form the table entries they read themselves (same
// function, same bits; ~400 instructions per wave and component) -- a kernel boundary plus a one-wave-per-pair
// kernel are 4 ... 5 us of a small ensemble's half-step.  "Small" = up to this many (row wave, component) pairs in
// the forward launch, about where the chip's wave slots are full and the extra instructions stop being free
// (64 (walker, component) pairs at 512^2, 16 at 1024^2, ~110 at 288^2).
constexpr int kInWavePowTabWaves = 8192;
// walkers [w_off, w_off + n) of c->d_prep; after the kernel that wrote their records, same stream.
// (Only the forward row kernels read the tables: k_raster_sums keeps the log2 + exp2 form at every size.)
static void launch_pow_tables(psfmc_ctx* c, int n, int w_off, const uint8_t* skip, hipStream_t st) {
    c->prep_tabs_valid = true;
    if (c->backend != PSFMC_BACKEND_FUSED || c->n_sersic == 0 || n <= 0 || !c->use_pow_tabs) return;
    const int pairs = n * c->n_sersic;
    const bool build = (long long)pairs * c->nblk > kInWavePowTabWaves;
    // a batch written in several pieces (w_off > 0) reads its tables from memory only if every piece has them
    c->prep_tabs_built = w_off == 0 ? build : (c->prep_tabs_built && build);
    if (!build) return;
    hipLaunchKernelGGL(k_pow_tables, dim3((pairs + 3) / 4), dim3(256), 0, st, c->d_prep + (size_t)w_off * c->plen,
                       skip, pairs, c->n_ps, c->n_sersic);
}

// field < 0: the rows' PSF index counts over every kernel spectrum of the context; field >= 0: within that field
static int eval_device(psfmc_ctx* c, int W, const double* d_rows, const uint8_t* d_skip,
                       double* d_like, hipStream_t st, int field = -1) {
    c->prep_tabs_valid = false;                       // d_prep is being rewritten
    hipLaunchKernelGGL(k_prep, dim3((W + 127) / 128), dim3(128), 0, st, d_rows, c->d_prep, W, c->n_ps,
                       c->n_sersic, c->ly, c->lx, c->d_rho, field < 0 ? c->n_psf : c->n_psf_field,
                       field < 0 ? 0 : field * c->n_psf_field);
    launch_pow_tables(c, W, 0, d_skip, st);
    RC_TRY(run_pipeline(c, W, d_skip, st));
    hipLaunchKernelGGL(k_finish, dim3(finish_blocks(W)), dim3(kFinishThreads), 0, st, c->d_partial, d_skip, d_like,
                       W, c->nblk);
    HIP_TRY(hipGetLastError());
    return PSFMC_OK;
}

// raw vectors (or, with sp.pos set, stretch-move proposals formed on the fly) -> prep
// records, log-priors and skip flags of W walkers
// `field` / `w_off`: walkers [w_off, w_off + W) of the batch belong to observed field `field` (0, 0
// for the usual one-field context): its layout, its block of kernel spectra
// `n_seg` > 1: ONE launch for the fields `field` .. `field + n_seg - 1`, W walkers each (blockIdx.y = field):
// the same per-walker arrays with field f's block at offset f W, every field's own layout
static void launch_theta_prep(psfmc_ctx* c, int W, const double* d_theta, const double* d_extra,
                              double* d_rows, hipStream_t st, const StretchIn& sp, int field = 0, int w_off = 0,
                              int n_seg = 1) {
    const ThetaLayout& L = field == 0 ? c->layout : c->more_layouts[field - 1];
    FieldSegs segs{nullptr, 0};
    if (n_seg > 1) segs = FieldSegs{c->d_field_layouts + field, c->n_psf_field};
    if (w_off == 0) c->prep_tabs_valid = false;       // d_prep is being rewritten (w_off > 0: a further piece of one batch)
    hipLaunchKernelGGL(k_theta_prep, dim3((W + kThetaThreads - 1) / kThetaThreads, n_seg),
                       dim3(kThetaThreads, theta_task_waves(c->n_ps, c->n_sersic)), c->theta_lds, st,
                       L, d_theta, d_extra, d_rows, c->d_prep + (size_t)w_off * c->plen, c->d_lnprior + w_off,
                       c->d_skip + w_off, W, c->ly, c->lx, c->d_rho, sp, field * c->n_psf_field, segs);
    launch_pow_tables(c, W * n_seg, w_off, c->d_skip + w_off, st);
}

// raw vectors -> log-posterior, everything on the device
static int eval_theta_device(psfmc_ctx* c, int W, const double* d_theta, const double* d_extra,
                             double* d_lnprob, hipStream_t st) {
    launch_theta_prep(c, W, d_theta, d_extra, nullptr, st, StretchIn{});
    RC_TRY(run_pipeline(c, W, c->d_skip, st));
    hipLaunchKernelGGL(k_finish_posterior, dim3(finish_blocks(W)), dim3(kFinishThreads), 0, st, c->d_partial, c->d_skip,
                       c->d_lnprior, d_lnprob, W, c->nblk);
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

static int eval_batch_impl(psfmc_ctx* c, int field, int W, const double* rows, const uint8_t* skip, double* loglike);

extern "C" int psfmc_eval_batch(psfmc_ctx* c, int W, const double* rows, const uint8_t* skip,
                                double* loglike) {
    return eval_batch_impl(c, -1, W, rows, skip, loglike);
}

// the log-likelihoods of derived rows of ONE field of a psfmc_ctx_create_fields context (the rows' PSF index
// counts within the field, as for psfmc_eval_images_field)
extern "C" int psfmc_eval_batch_field(psfmc_ctx* c, int field, int W, const double* rows, const uint8_t* skip,
                                      double* loglike) {
    if (c && (field < 0 || field >= c->n_fields)) return fail(PSFMC_EINVAL, "field %d of %d", field, c->n_fields);
    return eval_batch_impl(c, field, W, rows, skip, loglike);
}

static int eval_batch_impl(psfmc_ctx* c, int field, int W, const double* rows, const uint8_t* skip, double* loglike) {
    int rc = check_call(c, W, rows, loglike);
    if (rc != PSFMC_OK || W == 0) return rc;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    HIP_TRY(hipMemcpyAsync(c->d_rows, rows, (size_t)W * c->rlen * sizeof(double), hipMemcpyHostToDevice, st));
    if (skip) HIP_TRY(hipMemcpyAsync(c->d_skip, skip, (size_t)W, hipMemcpyHostToDevice, st));
    rc = eval_device(c, W, c->d_rows, skip ? c->d_skip : nullptr, c->d_like, st, field);
    if (rc != PSFMC_OK) return rc;
    HIP_TRY(hipMemcpyAsync(loglike, c->d_like, (size_t)W * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return PSFMC_OK;
}

// ---------------------------------------------------------------------------
// images (models.py:222-226)
// ---------------------------------------------------------------------------
static int ensure_image_staging(psfmc_ctx* c) {
    if (c->img_cap >= c->chunk) return PSFMC_OK;
    if (c->d_img0) { (void)hipFree(c->d_img0); c->d_img0 = nullptr; }
    if (c->d_img1) { (void)hipFree(c->d_img1); c->d_img1 = nullptr; }
    if (c->d_rawstage) { (void)hipFree(c->d_rawstage); c->d_rawstage = nullptr; }
    c->img_cap = 0;
    HIP_TRY(hipMalloc(&c->d_img0, (size_t)c->chunk * c->S * sizeof(double)));
    HIP_TRY(hipMalloc(&c->d_img1, (size_t)c->chunk * c->S * sizeof(double)));
    c->img_cap = c->chunk;
    return PSFMC_OK;
}

static int eval_images_impl(psfmc_ctx* c, int field, int W, const double* rows, double* raw, double* conv,
                            double* resid, double* ivm, double* ps_sub) {
    int rc = check_call(c, W, rows, rows);
    if (rc != PSFMC_OK || W == 0) return rc;
    if (field < 0 || field >= c->n_fields) return fail(PSFMC_EINVAL, "field %d of %d", field, c->n_fields);
    const double* f_sci = c->d_sci + (size_t)field * c->S;      // this field's pixels
    const double* f_var = c->d_var + (size_t)field * c->S;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const bool fused = c->backend == PSFMC_BACKEND_FUSED;
    const size_t S_img = (size_t)c->ly * c->lx;                 // pixels of a host image
    const size_t img = S_img * sizeof(double);
    HIP_TRY(hipMemcpyAsync(c->d_rows, rows, (size_t)W * c->rlen * sizeof(double), hipMemcpyHostToDevice, st));
    c->prep_tabs_valid = false;                       // d_prep is being rewritten
    hipLaunchKernelGGL(k_prep, dim3((W + 127) / 128), dim3(128), 0, st, c->d_rows, c->d_prep, W, c->n_ps,
                       c->n_sersic, c->ly, c->lx, c->d_rho, c->n_psf_field, field * c->n_psf_field);
    launch_pow_tables(c, W, 0, nullptr, st);
    RC_TRY(ensure_image_staging(c));
    double *d_out = nullptr, *d_rawdev = nullptr;     // [chunk] staging for derived images / the raw models (transform shape)
    HIP_TRY(hipMalloc(&d_out, (size_t)c->chunk * img));
    if (raw && fused && hipMalloc(&d_rawdev, (size_t)c->chunk * c->S * sizeof(double)) != hipSuccess) {
        (void)hipFree(d_out);
        return fail(PSFMC_ENOMEM, "hipMalloc (raw-model staging)");
    }
    // where the convolved model / model variance of walker w live after a pass
    const double* conv_src = fused ? c->d_img0 : c->d_real;
    const double* var_src = fused ? c->d_img1 : c->d_real;
    const int stride = fused ? 1 : 2, var_c = fused ? 0 : 1;
    for (int w0 = 0; w0 < W && rc == PSFMC_OK; w0 += c->chunk) {
        const int n = W - w0 < c->chunk ? W - w0 : c->chunk;
        const double* prep = c->d_prep + (size_t)w0 * c->plen;
        auto emit = [&](double* host, const double* src, int strd, int comp, int op) -> int {
            if (!host) return PSFMC_OK;
            hipLaunchKernelGGL(k_image_out, dim3(64, n), dim3(256), 0, st, src, f_sci, f_var, d_out,
                               c->S, strd, comp, op, img_window(c));
            HIP_TRY(hipMemcpyAsync(host + (size_t)w0 * S_img, d_out, (size_t)n * img,
                                   hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            return PSFMC_OK;
        };
        auto pass = [&](int ps_only, double* raw_dev) -> int {
            if (fused) {
                RC_TRY(fused_forward(c, n, c->d_T, prep, nullptr, ps_only, raw_dev, st));
                return fused_inverse(c, n, c->d_T, prep, nullptr, c->d_partial, c->d_img0, c->d_img1, st);
            }
            return hipfft_convolve(c, n, prep, nullptr, st, ps_only);
        };
        if (raw && !fused) {   // raw model before the inverse transform overwrites it
            hipLaunchKernelGGL(k_raster, dim3((c->S + 1023) / 1024, n), dim3(256),
                               (size_t)prep_rec_len(c->n_ps, c->n_sersic) * sizeof(double), st, prep,
                               (const uint8_t*)nullptr, c->d_real, c->n_ps, c->n_sersic, c->ny, c->nx, 0);
            rc = emit(raw, c->d_real, 2, 0, IMG_COPY);
            if (rc != PSFMC_OK) break;
        }
        if (conv || resid || ivm || (raw && fused)) {
            rc = pass(0, (raw && fused) ? d_rawdev : nullptr);
            if (rc == PSFMC_OK && raw && fused) rc = emit(raw, d_rawdev, 1, 0, IMG_COPY);
            if (rc == PSFMC_OK) rc = emit(conv, conv_src, stride, 0, IMG_COPY);
            if (rc == PSFMC_OK) rc = emit(resid, conv_src, stride, 0, IMG_RESID);
            if (rc == PSFMC_OK) rc = emit(ivm, var_src, stride, var_c, IMG_IVM);
        }
        if (rc == PSFMC_OK && ps_sub) {
            rc = pass(1, nullptr);
            if (rc == PSFMC_OK) rc = emit(ps_sub, conv_src, stride, 0, IMG_RESID);
        }
    }
    (void)hipStreamSynchronize(st);
    (void)hipFree(d_out);
    if (d_rawdev) (void)hipFree(d_rawdev);
    if (rc == PSFMC_OK) HIP_TRY(hipGetLastError());
    return rc;
}

extern "C" int psfmc_eval_images(psfmc_ctx* c, int W, const double* rows, double* raw, double* conv,
                                 double* resid, double* ivm, double* ps_sub) {
    if (c && c->n_fields > 1) return fail(PSFMC_EINVAL, "this entry point serves contexts of one field");
    return eval_images_impl(c, 0, W, rows, raw, conv, resid, ivm, ps_sub);
}

// the five images of walkers of ONE field of a psfmc_ctx_create_fields context (rows as for
// psfmc_eval_images; their PSF index counts within the field)
extern "C" int psfmc_eval_images_field(psfmc_ctx* c, int field, int W, const double* rows, double* raw,
                                       double* conv, double* resid, double* ivm, double* ps_sub) {
    return eval_images_impl(c, field, W, rows, raw, conv, resid, ivm, ps_sub);
}

// ---------------------------------------------------------------------------
// raw-vector path
// ---------------------------------------------------------------------------
static int set_layout_impl(psfmc_ctx* c, int field, int n_sky, int n_params, const int* slot_col,
                           const double* slot_const, const int* ps_method, const int* sersic_degrees,
                           double mag_zeropoint, const int* family, const double* p0,
                           const double* p1, const double* p2) {
    if (!c) return fail(PSFMC_EINVAL, "ctx is NULL");
    if (field < 0 || field >= c->n_fields) return fail(PSFMC_EINVAL, "field %d of %d", field, c->n_fields);
    if (field > 0 && (!c->has_layout || n_sky != c->layout.n_sky || n_params != c->layout.n_params))
        return fail(PSFMC_EINVAL, "set field 0's layout first; every field has the same slots and columns");
    if (n_sky < 0 || n_sky > 16 || n_params < 0 || n_params > 4096) return fail(PSFMC_EINVAL, "bad counts");
    const int ns = n_slots(n_sky, c->n_ps, c->n_sersic);
    if (!slot_col || !slot_const || (c->n_ps && !ps_method) || (c->n_sersic && !sersic_degrees) ||
        (n_params && (!family || !p0 || !p1 || !p2)))
        return fail(PSFMC_EINVAL, "NULL layout array");
    for (int i = 0; i < ns; ++i)
        if (slot_col[i] < -1 || slot_col[i] >= n_params)
            return fail(PSFMC_EINVAL, "slot %d refers to column %d of %d", i, slot_col[i], n_params);
    for (int i = 0; i < n_params; ++i)
        if (family[i] < 0 || family[i] > PRIOR_RANDINT)
            return fail(PSFMC_EINVAL, "unknown prior family %d for column %d", family[i], i);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipDeviceSynchronize());
    // pack everything into one device allocation (8-byte units)
    const size_t n_int = (size_t)ns + c->n_ps + c->n_sersic + n_params;
    const size_t n_dbl = (size_t)ns + 4 * (size_t)n_params;
    const size_t int_bytes = (n_int * sizeof(int) + 7) / 8 * 8;
    std::vector<unsigned char> blob(int_bytes + n_dbl * sizeof(double));
    int* ip = reinterpret_cast<int*>(blob.data());
    double* dp = reinterpret_cast<double*>(blob.data() + int_bytes);
    memcpy(ip, slot_col, ns * sizeof(int));
    if (c->n_ps) memcpy(ip + ns, ps_method, c->n_ps * sizeof(int));
    if (c->n_sersic) memcpy(ip + ns + c->n_ps, sersic_degrees, c->n_sersic * sizeof(int));
    if (n_params) memcpy(ip + ns + c->n_ps + c->n_sersic, family, n_params * sizeof(int));
    memcpy(dp, slot_const, ns * sizeof(double));
    if (n_params) {
        memcpy(dp + ns, p0, n_params * sizeof(double));
        memcpy(dp + ns + n_params, p1, n_params * sizeof(double));
        memcpy(dp + ns + 2 * n_params, p2, n_params * sizeof(double));
        for (int i = 0; i < n_params; ++i) dp[ns + 3 * n_params + i] = prior_log_norm(family[i], p0[i], p1[i], p2[i]);
    }
    if (field > 0 && c->more_layouts.size() < (size_t)c->n_fields - 1) {
        c->more_layouts.resize(c->n_fields - 1);
        c->more_blobs.resize(c->n_fields - 1, nullptr);
        c->more_has.resize(c->n_fields - 1, 0);
    }
    void** blob_slot = field == 0 ? &c->d_layout_blob : &c->more_blobs[field - 1];
    if (*blob_slot) { (void)hipFree(*blob_slot); *blob_slot = nullptr; }
    HIP_TRY(hipMalloc(blob_slot, blob.size()));
    HIP_TRY(hipMemcpy(*blob_slot, blob.data(), blob.size(), hipMemcpyHostToDevice));
    const int* dip = reinterpret_cast<const int*>(*blob_slot);
    const double* ddp = reinterpret_cast<const double*>(static_cast<unsigned char*>(*blob_slot) + int_bytes);
    ThetaLayout& L = field == 0 ? c->layout : c->more_layouts[field - 1];
    L.n_sky = n_sky; L.n_ps = c->n_ps; L.n_sersic = c->n_sersic; L.n_params = n_params; L.n_psf = c->n_psf_field;
    L.mag_zp = mag_zeropoint;
    L.slot_col = dip; L.ps_method = dip + ns; L.sersic_deg = dip + ns + c->n_ps;
    L.family = dip + ns + c->n_ps + c->n_sersic;
    L.slot_const = ddp; L.pa = ddp + ns; L.pb = ddp + ns + n_params; L.pc = ddp + ns + 2 * n_params;
    L.pk = ddp + ns + 3 * n_params;
    if (c->n_fields > 1) {
        if (!c->d_field_layouts) HIP_TRY(hipMalloc(&c->d_field_layouts, (size_t)c->n_fields * sizeof(ThetaLayout)));
        HIP_TRY(hipMemcpy(c->d_field_layouts + field, &L, sizeof(ThetaLayout), hipMemcpyHostToDevice));
    }
    if (field > 0) {
        c->more_has[field - 1] = 1;
        return PSFMC_OK;
    }
    for (double** p : {&c->d_theta, &c->d_extra, &c->d_lnprior})
        if (*p) { (void)hipFree(*p); *p = nullptr; }
    HIP_TRY(hipMalloc(&c->d_theta, (size_t)c->max_walkers * (n_params > 0 ? n_params : 1) * sizeof(double)));
    HIP_TRY(hipMalloc(&c->d_extra, (size_t)c->max_walkers * sizeof(double)));
    HIP_TRY(hipMalloc(&c->d_lnprior, (size_t)c->max_walkers * sizeof(double)));
    c->theta_lds = theta_prep_lds_bytes(n_sky, c->n_ps, c->n_sersic, n_params);
    if (c->theta_lds > 64 * 1024) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_theta_prep),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->theta_lds));
    }
    c->has_layout = true;
    return PSFMC_OK;
}

extern "C" int psfmc_set_layout(psfmc_ctx* c, int n_sky, int n_params, const int* slot_col,
                                const double* slot_const, const int* ps_method, const int* sersic_degrees,
                                double mag_zeropoint, const int* family, const double* p0,
                                const double* p1, const double* p2) {
    return set_layout_impl(c, 0, n_sky, n_params, slot_col, slot_const, ps_method, sersic_degrees, mag_zeropoint,
                           family, p0, p1, p2);
}

// the layout of one field of a psfmc_ctx_create_fields context (field 0 first; the same slot / column
// structure for every field, their own constants and priors)
extern "C" int psfmc_set_layout_field(psfmc_ctx* c, int field, int n_sky, int n_params, const int* slot_col,
                                      const double* slot_const, const int* ps_method, const int* sersic_degrees,
                                      double mag_zeropoint, const int* family, const double* p0,
                                      const double* p1, const double* p2) {
    return set_layout_impl(c, field, n_sky, n_params, slot_col, slot_const, ps_method, sersic_degrees,
                           mag_zeropoint, family, p0, p1, p2);
}

static int check_theta_call(psfmc_ctx* c, int W, const void* theta, const void* out) {
    RC_TRY(check_call(c, W, theta ? theta : out, out));
    if (!c->has_layout) return fail(PSFMC_EINVAL, "psfmc_set_layout has not been called");
    if (W > 0 && c->layout.n_params > 0 && !theta) return fail(PSFMC_EINVAL, "NULL theta");
    return PSFMC_OK;
}

extern "C" int psfmc_eval_theta_device(psfmc_ctx* c, int W, const double* d_theta, const double* d_extra,
                                       double* d_lnprob, void* stream) {
    int rc = check_theta_call(c, W, d_theta, d_lnprob);
    if (rc != PSFMC_OK || W == 0) return rc;
    HIP_TRY(hipSetDevice(c->device));
    return eval_theta_device(c, W, d_theta, d_extra, d_lnprob, stream ? (hipStream_t)stream : c->stream);
}

// Walkers of several fields in one batch: segment i = seg_count[i] consecutive walkers of field
// seg_field[i] (host arrays); d_theta [W][P] / d_lnprob [W] in that order, W = sum of the counts.
// One record derivation per segment (its field's layout), then ONE pass of the likelihood pipeline
// over all W walkers.
extern "C" int psfmc_eval_theta_device_fields(psfmc_ctx* c, int n_seg, const int* seg_field, const int* seg_count,
                                              const double* d_theta, const double* d_extra, double* d_lnprob,
                                              void* stream) {
    if (!c || n_seg < 0 || (n_seg && (!seg_field || !seg_count))) return fail(PSFMC_EINVAL, "bad segment list");
    if (!c->has_layout) return fail(PSFMC_EINVAL, "psfmc_set_layout has not been called");
    long long W = 0;
    for (int i = 0; i < n_seg; ++i) {
        if (seg_count[i] < 0 || seg_field[i] < 0 || seg_field[i] >= c->n_fields)
            return fail(PSFMC_EINVAL, "segment %d: field %d, count %d", i, seg_field[i], seg_count[i]);
        if (seg_field[i] > 0 && ((size_t)seg_field[i] > c->more_has.size() || !c->more_has[seg_field[i] - 1]))
            return fail(PSFMC_EINVAL, "field %d has no layout (psfmc_set_layout_field)", seg_field[i]);
        W += seg_count[i];
    }
    if (W > c->max_walkers) return fail(PSFMC_EINVAL, "W=%lld outside [0, max_walkers=%d]", W, c->max_walkers);
    if (W == 0) return PSFMC_OK;
    if (!d_lnprob || (c->layout.n_params > 0 && !d_theta)) return fail(PSFMC_EINVAL, "NULL buffer");
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = stream ? (hipStream_t)stream : c->stream;
    const int P = c->layout.n_params;
    int off = 0;
    for (int i = 0; i < n_seg; ++i) {
        if (!seg_count[i]) continue;
        launch_theta_prep(c, seg_count[i], d_theta + (size_t)off * P, d_extra ? d_extra + off : nullptr, nullptr, st,
                          StretchIn{}, seg_field[i], off);
        off += seg_count[i];
    }
    RC_TRY(run_pipeline(c, (int)W, c->d_skip, st));
    hipLaunchKernelGGL(k_finish_posterior, dim3(finish_blocks((int)W)), dim3(kFinishThreads), 0, st, c->d_partial,
                       c->d_skip, c->d_lnprior, d_lnprob, (int)W, c->nblk);
    HIP_TRY(hipGetLastError());
    return PSFMC_OK;
}

// host-buffer form of psfmc_eval_theta_device_fields
extern "C" int psfmc_eval_theta_fields(psfmc_ctx* c, int n_seg, const int* seg_field, const int* seg_count,
                                       const double* theta, const double* extra, double* lnprob) {
    if (!c || n_seg < 0 || (n_seg && !seg_count)) return fail(PSFMC_EINVAL, "bad segment list");
    if (!c->has_layout) return fail(PSFMC_EINVAL, "psfmc_set_layout has not been called");
    long long W = 0;
    for (int i = 0; i < n_seg; ++i) W += seg_count[i] > 0 ? seg_count[i] : 0;
    if (W > c->max_walkers) return fail(PSFMC_EINVAL, "W=%lld outside [0, max_walkers=%d]", W, c->max_walkers);
    if (W == 0) return PSFMC_OK;
    if (!lnprob || (c->layout.n_params > 0 && !theta)) return fail(PSFMC_EINVAL, "NULL buffer");
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    if (c->layout.n_params)
        HIP_TRY(hipMemcpyAsync(c->d_theta, theta, (size_t)W * c->layout.n_params * sizeof(double),
                               hipMemcpyHostToDevice, st));
    if (extra) HIP_TRY(hipMemcpyAsync(c->d_extra, extra, (size_t)W * sizeof(double), hipMemcpyHostToDevice, st));
    RC_TRY(psfmc_eval_theta_device_fields(c, n_seg, seg_field, seg_count, c->d_theta, extra ? c->d_extra : nullptr,
                                          c->d_like, st));
    HIP_TRY(hipMemcpyAsync(lnprob, c->d_like, (size_t)W * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return PSFMC_OK;
}

extern "C" int psfmc_eval_theta(psfmc_ctx* c, int W, const double* theta, const double* extra,
                                double* lnprob) {
    int rc = check_theta_call(c, W, theta, lnprob);
    if (rc != PSFMC_OK || W == 0) return rc;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    if (c->layout.n_params)
        HIP_TRY(hipMemcpyAsync(c->d_theta, theta, (size_t)W * c->layout.n_params * sizeof(double),
                               hipMemcpyHostToDevice, st));
    if (extra) HIP_TRY(hipMemcpyAsync(c->d_extra, extra, (size_t)W * sizeof(double), hipMemcpyHostToDevice, st));
    RC_TRY(eval_theta_device(c, W, c->d_theta, extra ? c->d_extra : nullptr, c->d_like, st));
    HIP_TRY(hipMemcpyAsync(lnprob, c->d_like, (size_t)W * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return PSFMC_OK;
}

extern "C" int psfmc_debug_theta_rows(psfmc_ctx* c, int W, const double* theta, double* rows,
                                      double* lnprior, uint8_t* skip) {
    int rc = check_theta_call(c, W, theta, rows);
    if (rc != PSFMC_OK || W == 0) return rc;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    if (c->layout.n_params)
        HIP_TRY(hipMemcpyAsync(c->d_theta, theta, (size_t)W * c->layout.n_params * sizeof(double),
                               hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(c->d_rows, 0, (size_t)W * c->rlen * sizeof(double), st));
    launch_theta_prep(c, W, c->d_theta, nullptr, c->d_rows, st, StretchIn{});
    HIP_TRY(hipMemcpyAsync(rows, c->d_rows, (size_t)W * c->rlen * sizeof(double), hipMemcpyDeviceToHost, st));
    if (lnprior) HIP_TRY(hipMemcpyAsync(lnprior, c->d_lnprior, (size_t)W * sizeof(double), hipMemcpyDeviceToHost, st));
    if (skip) HIP_TRY(hipMemcpyAsync(skip, c->d_skip, (size_t)W, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return PSFMC_OK;
}

// ---------------------------------------------------------------------------
// posterior-image accumulation (models.py:74-97)
// ---------------------------------------------------------------------------
extern "C" int psfmc_reset_accumulated(psfmc_ctx* c) {
    if (!c) return fail(PSFMC_EINVAL, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    const size_t acc_bytes = (size_t)c->n_fields * 4 * c->S * sizeof(double);
    if (!c->d_acc) HIP_TRY(hipMalloc(&c->d_acc, acc_bytes));
    HIP_TRY(hipMemsetAsync(c->d_acc, 0, acc_bytes, c->stream));
    if (c->d_lin) HIP_TRY(hipMemsetAsync(c->d_lin, 0, (size_t)c->n_psf * 3 * c->S * sizeof(double), c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int f = 0; f < c->n_fields; ++f) acc_n(c, f) = 0;
    c->lin_pending = 0;
    return PSFMC_OK;
}

// one field of a psfmc_ctx_create_fields context: its image sums, the linear sums of its kernel spectra
// (pending samples of OTHER fields stay pending) and its count
extern "C" int psfmc_reset_accumulated_field(psfmc_ctx* c, int field) {
    if (!c) return fail(PSFMC_EINVAL, "ctx is NULL");
    if (field < 0 || field >= c->n_fields) return fail(PSFMC_EINVAL, "field %d of %d", field, c->n_fields);
    if (!c->d_acc) return psfmc_reset_accumulated(c);
    HIP_TRY(hipSetDevice(c->device));
    const size_t S = (size_t)c->S;
    HIP_TRY(hipMemsetAsync(c->d_acc + (size_t)field * 4 * S, 0, 4 * S * sizeof(double), c->stream));
    if (c->d_lin)
        HIP_TRY(hipMemsetAsync(c->d_lin + (size_t)field * c->n_psf_field * 3 * S, 0,
                               (size_t)c->n_psf_field * 3 * S * sizeof(double), c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    acc_n(c, field) = 0;
    return PSFMC_OK;
}

// buffers of the linear sums (fused back end): allocated outside any stream capture
static int ensure_linear_sums(psfmc_ctx* c) {
    if (c->backend != PSFMC_BACKEND_FUSED || c->d_lin) return PSFMC_OK;
    if (c->n_fields > 1) {
        // several fields: a group's partial holds ONE field's sums; `lin_groups` = groups per field
        const size_t field_el = (size_t)c->n_psf_field * 3 * c->S;
        int gpf = (2048 + c->nblk * c->n_fields - 1) / (c->nblk * c->n_fields);
        const size_t budget = (size_t)96 << 20;
        while (gpf > 1 && (size_t)gpf * c->n_fields * field_el * sizeof(double) > budget) --gpf;
        if (gpf < 1) gpf = 1;
        HIP_TRY(hipMalloc(&c->d_lin, (size_t)c->n_fields * field_el * sizeof(double)));
        HIP_TRY(hipMemset(c->d_lin, 0, (size_t)c->n_fields * field_el * sizeof(double)));
        HIP_TRY(hipMalloc(&c->d_linpart, (size_t)gpf * c->n_fields * field_el * sizeof(double)));
        c->lin_groups = gpf;
        return PSFMC_OK;
    }
    const size_t per_group = (size_t)c->n_psf * 3 * c->S * sizeof(double);
    // walker groups of one call: enough waves to fill the chip (row groups x groups >= ~2048), the
    // partials within ~96 MB
    int groups = (2048 + c->nblk - 1) / c->nblk;
    const size_t budget = (size_t)96 << 20;
    if ((size_t)groups * per_group > budget) groups = (int)(budget / per_group);
    if (groups < 1) groups = 1;
    if (groups > 64) groups = 64;
    HIP_TRY(hipMalloc(&c->d_lin, per_group));
    HIP_TRY(hipMemset(c->d_lin, 0, per_group));
    HIP_TRY(hipMalloc(&c->d_linpart, (size_t)groups * per_group));
    c->lin_groups = groups;
    return PSFMC_OK;
}

// fused back end: add the W walkers whose prep records are in c->d_prep to the linear sums.  The sums are
// kept per kernel spectrum, i.e. per (field, PSF): walkers of several fields (`nf` consecutive fields from
// `f0`, `per` walkers each, W = nf * per) land in their own fields' sums by themselves.
static int accumulate_linear(psfmc_ctx* c, int W, hipStream_t st, int f0 = 0, int nf = 1, int per = -1) {
    if (per < 0) per = W;
    if (c->n_fields > 1) {
        // field-contiguous walkers: whole groups per field, each looking only at its field's kernel spectra
        // (with every group scanning all n_fields x PSFs for its walkers, eight fields accumulated at half the
        // rate of eight contexts)
        if ((long long)nf * per != W) return fail(PSFMC_EINVAL, "accumulation: %d fields x %d walkers != %d", nf, per, W);
        int gpf = c->lin_groups < per ? c->lin_groups : per;           // groups per field
        const int group_size = (per + gpf - 1) / gpf;
        gpf = (per + group_size - 1) / group_size;
        if (per % group_size) {
            // a field's last group is short: the next field's groups must still start at its first walker, which
            // g * group_size does not give -- one launch per field then
            for (int j = 0; j < nf; ++j) {
                SizeCall a;
                a.c = c; a.n = per; a.prep = c->d_prep + (size_t)j * per * c->plen; a.groups = gpf;
                a.group_size = group_size; a.st = st; a.per_field = per; a.field = f0 + j;
                RC_TRY(size_call(SZ_RASTER_SUMS, c->nx, a));
                const size_t n_el = (size_t)c->n_psf_field * 3 * c->S;
                hipLaunchKernelGGL(k_sum_partials_fields, dim3(512), dim3(256), 0, st, c->d_linpart, gpf, 1,
                                   c->d_lin + (size_t)(f0 + j) * n_el, n_el);
            }
        } else {
            SizeCall a;
            a.c = c; a.n = W; a.prep = c->d_prep; a.groups = gpf * nf; a.group_size = group_size; a.st = st;
            a.per_field = per; a.field = f0;
            RC_TRY(size_call(SZ_RASTER_SUMS, c->nx, a));
            const size_t n_el = (size_t)c->n_psf_field * 3 * c->S;
            hipLaunchKernelGGL(k_sum_partials_fields, dim3(512), dim3(256), 0, st, c->d_linpart, gpf, nf,
                               c->d_lin + (size_t)f0 * n_el, n_el);
        }
        c->lin_pending += W;
        for (int i = 0; i < nf; ++i) acc_n(c, f0 + i) += per;
        return PSFMC_OK;
    }
    const size_t n_el = (size_t)c->n_psf * 3 * c->S;
    int groups = c->lin_groups < W ? c->lin_groups : W;
    const int group_size = (W + groups - 1) / groups;
    groups = (W + group_size - 1) / group_size;
    {
        SizeCall a;
        a.c = c; a.n = W; a.prep = c->d_prep; a.groups = groups; a.group_size = group_size; a.st = st;
        RC_TRY(size_call(SZ_RASTER_SUMS, c->nx, a));
    }
    hipLaunchKernelGGL(k_sum_partials, dim3(512), dim3(256), 0, st, c->d_linpart, groups, c->d_lin, n_el);
    c->lin_pending += W;
    acc_n(c, f0) += W;
    return PSFMC_OK;
}

// convolve the pending linear sums into the four image sums of d_acc.  Per PSF three passes of
// the row / column kernels over ONE pseudo-walker read from memory: (sum raw, 0), (0, sum raw^2)
// and (sum PS-only raw, 0).  Each sum travels alone in its packed transform: added up, the samples'
// raw^2 is dominated by the brightest sample far more than their raw is, and packed together the
// two channels' rounding errors would leak into each other (measured on the `edge` fixture:
// 3e-6 relative in the weight map; alone, 1e-15).
static int flush_linear_sums(psfmc_ctx* c) {
    if (c->backend != PSFMC_BACKEND_FUSED || c->lin_pending == 0 || !c->d_lin) return PSFMC_OK;
    hipStream_t st = c->stream;
    HIP_TRY(hipStreamSynchronize(st));
    const size_t S = (size_t)c->S;
    std::vector<double> host(S), rho(c->n_psf);
    HIP_TRY(hipMemcpy(rho.data(), c->d_rho, c->n_psf * sizeof(double), hipMemcpyDeviceToHost));
    double *d_img = nullptr, *d_scale = nullptr, *d_fprep = nullptr, *d_out = nullptr;
    HIP_TRY(hipMalloc(&d_img, 2 * S * sizeof(double)));
    int rc = PSFMC_OK;
    if (hipMalloc(&d_scale, sizeof(double)) != hipSuccess ||
        hipMalloc(&d_fprep, (size_t)c->plen * sizeof(double)) != hipSuccess ||
        hipMalloc(&d_out, 2 * S * sizeof(double)) != hipSuccess)
        rc = fail(PSFMC_ENOMEM, "hipMalloc (posterior-image flush)");
    int field_of_p = 0;       // the field whose sums the current kernel spectrum's images go to
    auto acc = [&](const double* src, int slot) {
        hipLaunchKernelGGL(k_accumulate, dim3(256), dim3(256), 0, st, src,
                           c->d_acc + ((size_t)field_of_p * 4 + slot) * S, (int)S, 1, 1, 0);
    };
    // one pseudo-walker: z = img0 + i scale img1 through rows_fwd (from memory), cols, rows_inv;
    // d_out[0] = convolution of img0 with the PSF, d_out[1] = of img1 with the PSF variance map
    auto pass = [&](int psf, const double* img0, const double* img1, double scale) -> int {
        std::vector<double> fprep((size_t)c->plen, 0.0);
        fprep[kPrepPsfIdx] = (double)psf;
        fprep[kPrepMu] = scale;
        fprep[kPrepInvLambda] = 1.0 / (scale * rho[psf]);
        for (int h = 0; h < 2; ++h) {
            const double* src = h ? img1 : img0;
            if (src) HIP_TRY(hipMemcpyAsync(d_img + h * S, src, S * sizeof(double), hipMemcpyDeviceToDevice, st));
            else HIP_TRY(hipMemsetAsync(d_img + h * S, 0, S * sizeof(double), st));
        }
        HIP_TRY(hipMemcpyAsync(d_scale, &scale, sizeof(double), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d_fprep, fprep.data(), fprep.size() * sizeof(double), hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));                 // (fprep / scale are stack data)
        SizeCall a;
        a.c = c; a.n = 1; a.T = c->d_T; a.img = d_img; a.img_scale = d_scale; a.st = st; a.from_image = true;
        RC_TRY(size_call(SZ_ROWS_FWD, c->nx, a));
        a.from_image = false; a.prep = d_fprep;
        RC_TRY(size_call(SZ_COLS, c->ny, a));
        a.partial = c->d_partial; a.conv_out = d_out; a.var_out = d_out + S;
        RC_TRY(size_call(SZ_ROWS_INV, c->nx, a));
        return PSFMC_OK;
    };
    for (int p = 0; p < c->n_psf && rc == PSFMC_OK; ++p) {
        field_of_p = p / c->n_psf_field;
        const double* lin = c->d_lin + (size_t)p * 3 * S;
        // the variance channel's power-of-two scale, and whether any sample used this PSF
        if (hipMemcpy(host.data(), lin + S, S * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) {
            rc = fail(PSFMC_EHIP, "posterior-image flush: copy failed");
            break;
        }
        double peak_b = 0.0;
        bool any = false;
        for (size_t i = 0; i < S; ++i) {
            const double b = fabs(host[i]);
            any |= host[i] != 0.0;                                // (NaN counts)
            if (b > peak_b && b < 1e300) peak_b = b;
        }
        if (!any) continue;                                       // (raw^2 = 0 everywhere: no sample, or an empty model)
        double sc = 1.0;
        if (peak_b > 0.0) {
            int eb;
            (void)frexp(peak_b, &eb);
            sc = ldexp(1.0, -eb);
        }
        rc = pass(p, lin, nullptr, 1.0);
        if (rc != PSFMC_OK) break;
        acc(lin, 0);                          // raw
        acc(d_out, 1);                        // convolved model
        rc = pass(p, nullptr, lin + S, sc);
        if (rc != PSFMC_OK) break;
        acc(d_out + S, 2);                    // model variance
        rc = pass(p, lin + 2 * S, nullptr, 1.0);
        if (rc != PSFMC_OK) break;
        acc(d_out, 3);                        // PS-only convolved
        if (hipStreamSynchronize(st) != hipSuccess) rc = fail(PSFMC_EHIP, "posterior-image flush failed: %s",
                                                              hipGetErrorString(hipGetLastError()));
    }
    if (rc == PSFMC_OK) {
        (void)hipMemsetAsync(c->d_lin, 0, (size_t)c->n_psf * 3 * S * sizeof(double), st);
        (void)hipStreamSynchronize(st);
        c->lin_pending = 0;
    }
    (void)hipFree(d_img);
    if (d_scale) (void)hipFree(d_scale);
    if (d_fprep) (void)hipFree(d_fprep);
    if (d_out) (void)hipFree(d_out);
    return rc;
}

// add the images of the W walkers whose prep records are in c->d_prep to the sums
static int accumulate_from_prep(psfmc_ctx* c, int W, hipStream_t st, int f0 = 0, int nf = 1, int per = -1) {
    const bool fused = c->backend == PSFMC_BACKEND_FUSED;
    if (fused && c->linear_acc) {
        if (!c->d_lin) return fail(PSFMC_EINVAL, "linear sums not allocated");
        return accumulate_linear(c, W, st, f0, nf, per);
    }
    if (c->n_fields > 1) return fail(PSFMC_EINVAL, "contexts of several fields accumulate images as linear sums only");
    RC_TRY(ensure_image_staging(c));
    if (fused && !c->d_rawstage) HIP_TRY(hipMalloc(&c->d_rawstage, (size_t)c->img_cap * c->S * sizeof(double)));
    const double* conv_src = fused ? c->d_img0 : c->d_real;
    const double* var_src = fused ? c->d_img1 : c->d_real;
    const int stride = fused ? 1 : 2, var_c = fused ? 0 : 1;
    auto acc = [&](const double* src, int strd, int comp, int slot, int n) {
        hipLaunchKernelGGL(k_accumulate, dim3(256), dim3(256), 0, st, src, c->d_acc + (size_t)slot * c->S,
                           c->S, n, strd, comp);
    };
    for (int w0 = 0; w0 < W; w0 += c->chunk) {
        const int n = W - w0 < c->chunk ? W - w0 : c->chunk;
        const double* prep = c->d_prep + (size_t)w0 * c->plen;
        if (fused) {
            RC_TRY(fused_forward(c, n, c->d_T, prep, nullptr, 0, c->d_rawstage, st));
            RC_TRY(fused_inverse(c, n, c->d_T, prep, nullptr, c->d_partial, c->d_img0, c->d_img1, st));
            acc(c->d_rawstage, 1, 0, 0, n);
        } else {
            hipLaunchKernelGGL(k_raster, dim3((c->S + 1023) / 1024, n), dim3(256),
                               (size_t)prep_rec_len(c->n_ps, c->n_sersic) * sizeof(double), st, prep,
                               (const uint8_t*)nullptr, c->d_real, c->n_ps, c->n_sersic, c->ny, c->nx, 0);
            acc(c->d_real, 2, 0, 0, n);
            RC_TRY(hipfft_convolve(c, n, prep, nullptr, st, 0));
        }
        acc(conv_src, stride, 0, 1, n);
        acc(var_src, stride, var_c, 2, n);
        if (fused) {
            RC_TRY(fused_forward(c, n, c->d_T, prep, nullptr, 1, nullptr, st));
            RC_TRY(fused_inverse(c, n, c->d_T, prep, nullptr, c->d_partial, c->d_img0, c->d_img1, st));
        } else {
            RC_TRY(hipfft_convolve(c, n, prep, nullptr, st, 1));
        }
        acc(conv_src, stride, 0, 3, n);
    }
    c->acc_count += W;
    return PSFMC_OK;
}

extern "C" int psfmc_accumulate_images(psfmc_ctx* c, int W, const double* rows) {
    if (c && c->n_fields > 1) return fail(PSFMC_EINVAL, "this entry point serves contexts of one field");
    int rc = check_call(c, W, rows, rows);
    if (rc != PSFMC_OK || W == 0) return rc;
    HIP_TRY(hipSetDevice(c->device));
    if (!c->d_acc) RC_TRY(psfmc_reset_accumulated(c));
    RC_TRY(ensure_linear_sums(c));
    hipStream_t st = c->stream;
    HIP_TRY(hipMemcpyAsync(c->d_rows, rows, (size_t)W * c->rlen * sizeof(double), hipMemcpyHostToDevice, st));
    c->prep_tabs_valid = false;                       // d_prep is being rewritten
    hipLaunchKernelGGL(k_prep, dim3((W + 127) / 128), dim3(128), 0, st, c->d_rows, c->d_prep, W, c->n_ps,
                       c->n_sersic, c->ly, c->lx, c->d_rho, c->n_psf, 0);
    launch_pow_tables(c, W, 0, nullptr, st);
    rc = accumulate_from_prep(c, W, st);
    (void)hipStreamSynchronize(st);
    if (rc == PSFMC_OK) HIP_TRY(hipGetLastError());
    return rc;
}

// raw parameter vectors (host) -> posterior-image sums: the records are derived on the device like
// psfmc_eval_theta does (no host-side gammaincinv per sample when the images of a whole database
// are recomputed, analysis/images.py:62-74)
static int accumulate_theta_impl(psfmc_ctx* c, int field, int W, const double* theta) {
    int rc = check_theta_call(c, W, theta, theta ? (const void*)theta : (const void*)c);
    if (rc != PSFMC_OK || W == 0) return rc;
    if (field < 0 || field >= c->n_fields) return fail(PSFMC_EINVAL, "field %d of %d", field, c->n_fields);
    if (field > 0 && ((size_t)field > c->more_has.size() || !c->more_has[field - 1]))
        return fail(PSFMC_EINVAL, "field %d has no layout (psfmc_set_layout_field)", field);
    HIP_TRY(hipSetDevice(c->device));
    if (!c->d_acc) RC_TRY(psfmc_reset_accumulated(c));
    RC_TRY(ensure_linear_sums(c));
    hipStream_t st = c->stream;
    if (c->layout.n_params)
        HIP_TRY(hipMemcpyAsync(c->d_theta, theta, (size_t)W * c->layout.n_params * sizeof(double),
                               hipMemcpyHostToDevice, st));
    launch_theta_prep(c, W, c->d_theta, nullptr, nullptr, st, StretchIn{}, field, 0);
    rc = accumulate_from_prep(c, W, st, field, 1, W);
    (void)hipStreamSynchronize(st);
    if (rc == PSFMC_OK) HIP_TRY(hipGetLastError());
    return rc;
}

extern "C" int psfmc_accumulate_theta(psfmc_ctx* c, int W, const double* theta) {
    if (c && c->n_fields > 1) return fail(PSFMC_EINVAL, "this entry point serves contexts of one field");
    return accumulate_theta_impl(c, 0, W, theta);
}

// the same for one field of a psfmc_ctx_create_fields context
extern "C" int psfmc_accumulate_theta_field(psfmc_ctx* c, int field, int W, const double* theta) {
    return accumulate_theta_impl(c, field, W, theta);
}

// ---------------------------------------------------------------------------
// device-resident stretch-move sampling
// ---------------------------------------------------------------------------
// Sampler buffers live in the context and only ever grow: a block of iterations used to pay
// a dozen hipMalloc / hipFree.
template <typename Tp>
static int grow(Tp** p, size_t* cap, size_t need) {
    if (need <= *cap && *p) return PSFMC_OK;
    if (*p) { (void)hipFree(*p); *p = nullptr; *cap = 0; }
    HIP_TRY(hipMalloc(p, (need ? need : 1) * sizeof(Tp)));
    *cap = need ? need : 1;
    return PSFMC_OK;
}

// F: ensembles of the run -- 1 (an ordinary context), or every field of a psfmc_ctx_create_fields context
static int stretch_check(psfmc_ctx* c, int W, int F = 1) {
    if (!c) return fail(PSFMC_EINVAL, "ctx is NULL");
    if (F != c->n_fields)
        return fail(PSFMC_EINVAL, F == 1 ? "this entry point serves contexts of one field"
                                         : "psfmc_stretch_run_fields samples every field of the context");
    if (!c->has_layout) return fail(PSFMC_EINVAL, "psfmc_set_layout has not been called");
    if (W < 2 || (W & 1) || (long long)W * F > c->max_walkers)
        return fail(PSFMC_EINVAL, "W must be even and fields x W within max_walkers");
    const int P = c->layout.n_params;
    if (P < 1) return fail(PSFMC_EINVAL, "model has no free parameter");
    // every prior must be evaluated on the device
    std::vector<int> fam(P);
    HIP_TRY(hipSetDevice(c->device));
    for (int f = 0; f < F; ++f) {
        if (f > 0 && ((size_t)f > c->more_has.size() || !c->more_has[f - 1]))
            return fail(PSFMC_EINVAL, "field %d has no layout (psfmc_set_layout_field)", f);
        const ThetaLayout& L = f == 0 ? c->layout : c->more_layouts[f - 1];
        HIP_TRY(hipMemcpy(fam.data(), L.family, P * sizeof(int), hipMemcpyDeviceToHost));
        for (int v : fam)
            if (v == PRIOR_HOST) return fail(PSFMC_EINVAL, "a prior is evaluated on the host; use the host sampler");
    }
    return PSFMC_OK;
}

// upload the walkers, counters and the block's random numbers; size the chain buffers.  F ensembles of W
// walkers each: host arrays carry the ensemble as their leading dimension (pos [F][W][P], z [F][n_iter][2][half],
// ...); on the device the three random arrays are [3][F][n_iter W].
static int stretch_upload(psfmc_ctx* c, int W, int n_iter, const double* pos, const double* lnprob,
                          int lnprob_valid, const double* z, const double* lz, const int* partner,
                          const double* log_u, const long long* naccepted, bool store, hipStream_t st, int F = 1) {
    psfmc_ctx::Stretch& S = c->stretch;
    const int P = c->layout.n_params, half = W / 2;
    const size_t n_rand = (size_t)n_iter * W * F;             // per random array, all ensembles
    const size_t WF = (size_t)W * F;
    RC_TRY(grow(&S.pos, &S.cap_pos, WF * P));
    RC_TRY(grow(&S.lnp, &S.cap_lnp, WF));
    RC_TRY(grow(&S.q, &S.cap_q, (size_t)half * F * P * (F == 1 ? 3 : 1)));       // (three proposal sets: whole-iteration launches)
    RC_TRY(grow(&S.accflag, &S.cap_acc, (size_t)half));
    RC_TRY(grow(&S.newlnp, &S.cap_new, (size_t)half * F));
    RC_TRY(grow(&S.nacc, &S.cap_nacc, WF));
    RC_TRY(grow(&S.rand, &S.cap_rand, 3 * n_rand));
    RC_TRY(grow(&S.partner, &S.cap_partner, n_rand));
    RC_TRY(grow(&S.iter, &S.cap_iter, (size_t)1));
    if (store && n_iter) {
        RC_TRY(grow(&S.chain, &S.cap_chain, WF * n_iter * P));
        RC_TRY(grow(&S.lnchain, &S.cap_lnchain, WF * n_iter));
    }
    S.W = W; S.n_iter = n_iter; S.F = F; S.store = store && n_iter; S.open = true;
    if (n_rand) {
        HIP_TRY(hipMemcpyAsync(S.rand, z, n_rand * sizeof(double), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(S.rand + n_rand, lz, n_rand * sizeof(double), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(S.rand + 2 * n_rand, log_u, n_rand * sizeof(double), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(S.partner, partner, n_rand * sizeof(int), hipMemcpyHostToDevice, st));
    }
    HIP_TRY(hipMemcpyAsync(S.pos, pos, WF * P * sizeof(double), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(S.nacc, naccepted, WF * sizeof(long long), hipMemcpyHostToDevice, st));
    if (lnprob_valid)
        HIP_TRY(hipMemcpyAsync(S.lnp, lnprob, WF * sizeof(double), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(S.iter, 0, sizeof(int), st));
    return PSFMC_OK;
}

static int stretch_download(psfmc_ctx* c, double* pos, double* lnprob, double* chain, double* lnprob_chain,
                            long long* naccepted, hipStream_t st) {
    psfmc_ctx::Stretch& S = c->stretch;
    const int P = c->layout.n_params;
    const size_t W = (size_t)S.W * S.F;
    HIP_TRY(hipMemcpyAsync(pos, S.pos, W * P * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(lnprob, S.lnp, W * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(naccepted, S.nacc, W * sizeof(long long), hipMemcpyDeviceToHost, st));
    if (S.store && chain) {
        HIP_TRY(hipMemcpyAsync(chain, S.chain, W * S.n_iter * P * sizeof(double), hipMemcpyDeviceToHost, st));
        if (lnprob_chain)
            HIP_TRY(hipMemcpyAsync(lnprob_chain, S.lnchain, W * S.n_iter * sizeof(double),
                                   hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(hipStreamSynchronize(st));
    return PSFMC_OK;
}

// first stage of a half-step: the stretch proposals of half `h` at iteration `it`, their
// priors and prep records (every walker of the half: the accept step needs every proposal)
static void stretch_propose(psfmc_ctx* c, int it, int h, bool graph_iter, hipStream_t st) {
    psfmc_ctx::Stretch& S = c->stretch;
    const int half = S.W / 2, P = c->layout.n_params;
    const size_t per_field = (size_t)S.n_iter * S.W;          // random numbers of one ensemble
    launch_theta_prep(c, half, nullptr, nullptr, nullptr, st,
                      StretchIn{S.pos, S.q, S.rand, S.partner, graph_iter ? S.iter : nullptr, it, half, h,
                                (size_t)S.W * P, (size_t)half * P, per_field},
                      0, 0, S.F);
}

// last stage: sum (or take the gathered values), accept, move, chain entry
static void stretch_accept(psfmc_ctx* c, int it, int h, const double* d_newlnp, bool graph_iter, hipStream_t st) {
    psfmc_ctx::Stretch& S = c->stretch;
    const int half = S.W / 2;
    const size_t per_field = (size_t)S.n_iter * S.W;
    const size_t n_rand = per_field * S.F;
    // whole-iteration launches: the first half records its outcomes, the second half selects its rows by them
    uint8_t* acc_out = S.spec && h == 0 ? S.accflag : nullptr;
    const uint8_t* acc_in = S.spec && h == 1 ? S.accflag : nullptr;
    hipLaunchKernelGGL(k_stretch_finish, dim3(finish_blocks(half), S.F), dim3(kFinishThreads), 0, st, c->d_partial,
                       c->d_skip, c->d_lnprior, c->nblk, d_newlnp, S.pos, S.lnp, S.q, S.rand + n_rand,
                       S.rand + 2 * n_rand, S.nacc, S.store ? S.chain : nullptr, S.store ? S.lnchain : nullptr,
                       graph_iter ? S.iter : nullptr, it, S.n_iter, half, h, c->layout.n_params, per_field,
                       acc_out, acc_in, S.partner);
}

// a whole iteration's proposals in one launch (StretchIn::spec): 3 half walkers
static void stretch_propose_iteration(psfmc_ctx* c, int it, bool graph_iter, hipStream_t st) {
    psfmc_ctx::Stretch& S = c->stretch;
    const int half = S.W / 2, P = c->layout.n_params;
    StretchIn sp{S.pos, S.q, S.rand, S.partner, graph_iter ? S.iter : nullptr, it, half, 0,
                 (size_t)S.W * P, (size_t)half * P, (size_t)S.n_iter * S.W, 1};
    launch_theta_prep(c, 3 * half, nullptr, nullptr, nullptr, st, sp);
}

static int stretch_prepare_accumulation(psfmc_ctx* c) {
    if (!c->d_acc) RC_TRY(psfmc_reset_accumulated(c));
    RC_TRY(ensure_linear_sums(c));
    RC_TRY(ensure_image_staging(c));
    if (c->backend == PSFMC_BACKEND_FUSED && !c->linear_acc && !c->d_rawstage)
        HIP_TRY(hipMalloc(&c->d_rawstage, (size_t)c->img_cap * c->S * sizeof(double)));
    return PSFMC_OK;
}

// log-posteriors of the F ensembles' current positions (S.pos -> S.lnp): one record derivation per field,
// one pass of the likelihood pipeline over all F W walkers
static int stretch_eval_positions(psfmc_ctx* c, hipStream_t st) {
    psfmc_ctx::Stretch& S = c->stretch;
    if (S.F == 1) return eval_theta_device(c, S.W, S.pos, nullptr, S.lnp, st);
    launch_theta_prep(c, S.W, S.pos, nullptr, nullptr, st, StretchIn{}, 0, 0, S.F);
    RC_TRY(run_pipeline(c, S.W * S.F, c->d_skip, st));
    hipLaunchKernelGGL(k_finish_posterior, dim3(finish_blocks(S.W * S.F)), dim3(kFinishThreads), 0, st, c->d_partial,
                       c->d_skip, c->d_lnprior, S.lnp, S.W * S.F, c->nblk);
    HIP_TRY(hipGetLastError());
    return PSFMC_OK;
}

static int stretch_run_impl(psfmc_ctx* c, int F, int W, int n_iter, double* pos, double* lnprob,
                            int lnprob_valid, const double* z, const double* lz, const int* partner,
                            const double* log_u, double* chain, double* lnprob_chain,
                            long long* naccepted, int accumulate) {
    RC_TRY(stretch_check(c, W, F));
    if (n_iter < 0 || !pos || !lnprob || !naccepted || (n_iter && (!z || !lz || !partner || !log_u)))
        return fail(PSFMC_EINVAL, "NULL buffer");
    const int half = W / 2;
    hipStream_t st = c->stream;
    psfmc_ctx::Stretch& S = c->stretch;
    RC_TRY(stretch_upload(c, W, n_iter, pos, lnprob, lnprob_valid, z, lz, partner, log_u, naccepted,
                          chain != nullptr, st, F));
    if (!lnprob_valid) RC_TRY(stretch_eval_positions(c, st));
    if (accumulate) RC_TRY(stretch_prepare_accumulation(c));
    // one iteration: two half-ensemble proposals (chain entries included), optional image
    // sums.  A half-step is three stages: proposals + priors + prep records (one kernel), the
    // likelihood pipeline, and sum + accept + move + chain entry (one kernel).  The kernels take
    // the iteration number by value, or from *S.iter when one captured iteration is replayed as
    // a hipGraph.
    const bool use_d_iter = n_iter > 2 && c->use_graph && !c->profile && c->backend == PSFMC_BACKEND_FUSED && F == 1;
    int it_host = 0;
    // Small ensembles (the reference's default 2 P + 2 walkers, fitting.py:52-53) are latency-bound: a half-step of
    // 11 walkers takes as long as one of 33.  They run ONE pipeline pass per iteration over the first half's
    // proposals and BOTH candidate proposals of every second-half walker (partner moved / partner stayed), and the
    // second accept step picks by the first one's outcomes: the same chain bit for bit (per-walker results do not
    // depend on the batch), 6 launches per iteration instead of 10.
    // default rule, measured (tools/time_small_sampler.py; profiles/r3_small_sampler.txt): the single pass wins while
    // half an ensemble is at most ~3 M transform pixels -- 183 walkers at 128^2 (x1.55 at 22 walkers ... x1.10 at
    // 256), 45 at 256^2 (x1.31 at 22, x1.09 at 64), 11 at 512^2 (x1.07); beyond it the extra half-ensemble of
    // evaluations costs more than the launches it saves
    const int spec_bound = c->speculate >= 0 ? c->speculate : (int)(3.0e6 / ((double)c->ny * c->nx));
    S.spec = F == 1 && half <= spec_bound && 3 * half <= c->max_walkers && c->backend == PSFMC_BACKEND_FUSED;
    if (S.spec) ++c->speculated_runs;
    auto iteration = [&]() -> int {
        if (S.spec) {
            stretch_propose_iteration(c, it_host, use_d_iter, st);
            RC_TRY(run_pipeline(c, 3 * half, c->d_skip, st));
            stretch_accept(c, it_host, 0, nullptr, use_d_iter, st);
            stretch_accept(c, it_host, 1, nullptr, use_d_iter, st);
        } else for (int h = 0; h < 2; ++h) {
            stretch_propose(c, it_host, h, use_d_iter, st);
            RC_TRY(run_pipeline(c, half * F, c->d_skip, st));
            stretch_accept(c, it_host, h, nullptr, use_d_iter, st);
        }
        if (accumulate) {
            launch_theta_prep(c, W, S.pos, nullptr, nullptr, st, StretchIn{}, 0, 0, F);
            RC_TRY(accumulate_from_prep(c, W * F, st, 0, F, W));
        }
        if (use_d_iter) hipLaunchKernelGGL(k_stretch_next, dim3(1), dim3(1), 0, st, S.iter);
        ++it_host;
        return PSFMC_OK;
    };
    // Optionally capture one iteration into a hipGraph and replay it (set_option "graph";
    // measured: no gain, the iteration is kernel-time- not launch-bound).
    int rc = PSFMC_OK;
    hipGraphExec_t exec = nullptr;
    if (use_d_iter) {
        hipGraph_t graph = nullptr;
        // un-captured warm-up of the pipeline: one-time attribute calls must not fall
        // inside the capture
        rc = eval_theta_device(c, half, S.pos, nullptr, S.newlnp, st);      // (F == 1 here)
        const long long count_before = c->acc_count, pending_before = c->lin_pending;
        if (rc == PSFMC_OK && hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            const int crc = iteration();
            const hipError_t e = hipStreamEndCapture(st, &graph);
            if (crc != PSFMC_OK || e != hipSuccess || !graph ||
                hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess)
                exec = nullptr;
            if (graph) (void)hipGraphDestroy(graph);
            it_host = 0;
        }
        (void)hipGetLastError();
        c->acc_count = count_before;            // the captured pass did not run
        c->lin_pending = pending_before;
    }
    for (int it = 0; it < n_iter && rc == PSFMC_OK; ++it) {
        if (exec) {
            if (hipGraphLaunch(exec, st) != hipSuccess) rc = fail(PSFMC_EHIP, "hipGraphLaunch failed");
            else {
                ++c->graph_launches;
                if (accumulate) {
                    c->acc_count += W;
                    if (c->backend == PSFMC_BACKEND_FUSED && c->linear_acc) c->lin_pending += W;
                }
            }
        } else {
            rc = iteration();
        }
    }
    if (exec) {
        (void)hipStreamSynchronize(st);
        (void)hipGraphExecDestroy(exec);
    }
    if (rc == PSFMC_OK) rc = stretch_download(c, pos, lnprob, chain, lnprob_chain, naccepted, st);
    (void)hipStreamSynchronize(st);
    if (rc == PSFMC_OK && hipGetLastError() != hipSuccess) rc = fail(PSFMC_EHIP, "kernel launch failed");
    S.open = false;
    S.spec = false;
    return rc;
}

extern "C" int psfmc_stretch_run(psfmc_ctx* c, int W, int n_iter, double* pos, double* lnprob,
                                 int lnprob_valid, const double* z, const double* lz, const int* partner,
                                 const double* log_u, double* chain, double* lnprob_chain,
                                 long long* naccepted, int accumulate) {
    return stretch_run_impl(c, 1, W, n_iter, pos, lnprob, lnprob_valid, z, lz, partner, log_u, chain, lnprob_chain,
                            naccepted, accumulate);
}

// Every field of a psfmc_ctx_create_fields context sampled together: n_fields independent ensembles of W
// walkers, each with its own random numbers, their half-step proposals evaluated in ONE batch of
// n_fields W / 2 walkers.  All arrays carry the field as their leading dimension: pos [F][W][P],
// lnprob [F][W], z / lz / log_u / partner [F][n_iter][2][W/2], chain [F][W][n_iter][P],
// lnprob_chain [F][W][n_iter], naccepted [F][W].  A field's chain equals the chain of its own
// one-field context given the same start and random numbers, bit for bit (per-walker results do not
// depend on the batch).
extern "C" int psfmc_stretch_run_fields(psfmc_ctx* c, int W, int n_iter, double* pos, double* lnprob,
                                        int lnprob_valid, const double* z, const double* lz, const int* partner,
                                        const double* log_u, double* chain, double* lnprob_chain,
                                        long long* naccepted, int accumulate) {
    if (!c) return fail(PSFMC_EINVAL, "ctx is NULL");
    return stretch_run_impl(c, c->n_fields, W, n_iter, pos, lnprob, lnprob_valid, z, lz, partner, log_u, chain,
                            lnprob_chain, naccepted, accumulate);
}

// ---------------------------------------------------------------------------
// The same sampler, one half-step at a time, for walkers sharded over several GPUs: every
// rank holds ALL walkers and the same random numbers, evaluates the log-posterior of ITS
// block of each half-step's proposals, the blocks are all-gathered by the caller (RCCL /
// torch.distributed: the one collective of the path), and every rank applies the identical
// accept / move.  Per-walker log-posteriors do not depend on the batch they are evaluated
// in, so the chain equals the single-GPU chain bit for bit.
// ---------------------------------------------------------------------------
extern "C" int psfmc_stretch_open(psfmc_ctx* c, int W, int n_iter, const double* pos, const double* lnprob,
                                  const double* z, const double* lz, const int* partner,
                                  const double* log_u, const long long* naccepted, int store_chain) {
    RC_TRY(stretch_check(c, W));
    if (n_iter < 1 || !pos || !lnprob || !naccepted || !z || !lz || !partner || !log_u)
        return fail(PSFMC_EINVAL, "NULL buffer or n_iter < 1");
    RC_TRY(stretch_upload(c, W, n_iter, pos, lnprob, 1, z, lz, partner, log_u, naccepted, store_chain != 0,
                          c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PSFMC_OK;
}

static int stretch_step_check(psfmc_ctx* c, int it, int h) {
    if (!c) return fail(PSFMC_EINVAL, "ctx is NULL");
    if (!c->stretch.open) return fail(PSFMC_EINVAL, "psfmc_stretch_open has not been called");
    if (it < 0 || it >= c->stretch.n_iter || (h != 0 && h != 1))
        return fail(PSFMC_EINVAL, "iteration %d / half %d out of range", it, h);
    return PSFMC_OK;
}

extern "C" int psfmc_stretch_half_eval(psfmc_ctx* c, int it, int h, int lo, int n, double* d_newlnp_block,
                                       void* stream) {
    RC_TRY(stretch_step_check(c, it, h));
    const int half = c->stretch.W / 2;
    if (lo < 0 || n < 0 || lo + n > half) return fail(PSFMC_EINVAL, "block [%d, %d) outside the half-ensemble of %d", lo, lo + n, half);
    if (n > 0 && !d_newlnp_block) return fail(PSFMC_EINVAL, "NULL output");
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = stream ? (hipStream_t)stream : c->stream;
    stretch_propose(c, it, h, false, st);
    if (n > 0) {
        RC_TRY(run_pipeline(c, n, c->d_skip, st, lo));
        hipLaunchKernelGGL(k_finish_posterior, dim3(finish_blocks(n)), dim3(kFinishThreads), 0, st,
                           c->d_partial + (size_t)lo * c->nblk, c->d_skip + lo, c->d_lnprior + lo,
                           d_newlnp_block, n, c->nblk);
    }
    HIP_TRY(hipGetLastError());
    return PSFMC_OK;
}

extern "C" int psfmc_stretch_half_accept(psfmc_ctx* c, int it, int h, const double* d_newlnp_half, void* stream) {
    RC_TRY(stretch_step_check(c, it, h));
    if (!d_newlnp_half) return fail(PSFMC_EINVAL, "NULL input");
    HIP_TRY(hipSetDevice(c->device));
    stretch_accept(c, it, h, d_newlnp_half, false, stream ? (hipStream_t)stream : c->stream);
    HIP_TRY(hipGetLastError());
    return PSFMC_OK;
}

extern "C" int psfmc_stretch_accumulate(psfmc_ctx* c, int lo, int n, void* stream) {
    if (!c) return fail(PSFMC_EINVAL, "ctx is NULL");
    psfmc_ctx::Stretch& S = c->stretch;
    if (!S.open) return fail(PSFMC_EINVAL, "psfmc_stretch_open has not been called");
    if (lo < 0 || n < 0 || lo + n > S.W) return fail(PSFMC_EINVAL, "block outside the ensemble");
    if (n == 0) return PSFMC_OK;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = stream ? (hipStream_t)stream : c->stream;
    RC_TRY(stretch_prepare_accumulation(c));
    launch_theta_prep(c, n, S.pos + (size_t)lo * c->layout.n_params, nullptr, nullptr, st, StretchIn{});
    RC_TRY(accumulate_from_prep(c, n, st));
    HIP_TRY(hipGetLastError());
    return PSFMC_OK;
}

extern "C" int psfmc_stretch_close(psfmc_ctx* c, double* pos, double* lnprob, double* chain,
                                   double* lnprob_chain, long long* naccepted, void* stream) {
    if (!c) return fail(PSFMC_EINVAL, "ctx is NULL");
    if (!c->stretch.open) return fail(PSFMC_EINVAL, "psfmc_stretch_open has not been called");
    if (!pos || !lnprob || !naccepted) return fail(PSFMC_EINVAL, "NULL buffer");
    HIP_TRY(hipSetDevice(c->device));
    const int rc = stretch_download(c, pos, lnprob, chain, lnprob_chain, naccepted,
                                    stream ? (hipStream_t)stream : c->stream);
    c->stretch.open = false;
    return rc;
}

// raw sums of the posterior images ([4][ny][nx]: raw, convolved, model variance, PS-only
// convolved) and their sample count, for adding up the ranks' shares
extern "C" int psfmc_get_accumulated_sums(psfmc_ctx* c, double* sums, long long* count) {
    if (!c || !sums || !count) return fail(PSFMC_EINVAL, "NULL argument");
    if (c->n_fields > 1) return fail(PSFMC_EINVAL, "this entry point serves contexts of one field");
    HIP_TRY(hipSetDevice(c->device));
    RC_TRY(flush_linear_sums(c));
    *count = c->acc_count;
    const size_t S_img = (size_t)c->ly * c->lx;
    if (!c->d_acc) { memset(sums, 0, 4 * S_img * sizeof(double)); return PSFMC_OK; }
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (!c->embed) {
        HIP_TRY(hipMemcpy(sums, c->d_acc, 4 * S_img * sizeof(double), hipMemcpyDeviceToHost));
        return PSFMC_OK;
    }
    std::vector<double> full((size_t)4 * c->S);              // transform shape -> the image's window
    HIP_TRY(hipMemcpy(full.data(), c->d_acc, full.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (int k = 0; k < 4; ++k)
        for (int y = 0; y < c->ly; ++y)
            memcpy(sums + k * S_img + (size_t)y * c->lx,
                   &full[(size_t)k * c->S + (size_t)(y + c->wrap.ay) * c->nx + c->wrap.ax], (size_t)c->lx * sizeof(double));
    return PSFMC_OK;
}

extern "C" int psfmc_set_accumulated_sums(psfmc_ctx* c, const double* sums, long long count) {
    if (!c || !sums || count < 0) return fail(PSFMC_EINVAL, "bad argument");
    if (c->n_fields > 1) return fail(PSFMC_EINVAL, "this entry point serves contexts of one field");
    HIP_TRY(hipSetDevice(c->device));
    if (!c->d_acc) RC_TRY(psfmc_reset_accumulated(c));
    if (c->d_lin) HIP_TRY(hipMemset(c->d_lin, 0, (size_t)c->n_psf * 3 * c->S * sizeof(double)));
    c->lin_pending = 0;                                  // the new sums replace everything gathered so far
    if (!c->embed) {
        HIP_TRY(hipMemcpy(c->d_acc, sums, (size_t)4 * c->S * sizeof(double), hipMemcpyHostToDevice));
    } else {
        const size_t S_img = (size_t)c->ly * c->lx;
        std::vector<double> full((size_t)4 * c->S, 0.0);     // only the image's window is ever read back
        for (int k = 0; k < 4; ++k)
            for (int y = 0; y < c->ly; ++y)
                memcpy(&full[(size_t)k * c->S + (size_t)(y + c->wrap.ay) * c->nx + c->wrap.ax],
                       sums + k * S_img + (size_t)y * c->lx, (size_t)c->lx * sizeof(double));
        HIP_TRY(hipMemcpy(c->d_acc, full.data(), full.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    c->acc_count = count;
    return PSFMC_OK;
}

static int get_accumulated_impl(psfmc_ctx* c, int field, double* raw, double* conv, double* resid, double* ivm,
                                double* ps_sub, long long* count) {
    if (!c) return fail(PSFMC_EINVAL, "ctx is NULL");
    if (field < 0 || field >= c->n_fields) return fail(PSFMC_EINVAL, "field %d of %d", field, c->n_fields);
    const long long n = acc_n(c, field);
    if (count) *count = n;
    if (n == 0 || !c->d_acc) return PSFMC_OK;
    HIP_TRY(hipSetDevice(c->device));
    RC_TRY(flush_linear_sums(c));
    double* d_out = nullptr;
    const size_t S_img = (size_t)c->ly * c->lx;
    HIP_TRY(hipMalloc(&d_out, S_img * sizeof(double)));
    const double inv_n = 1.0 / (double)n;
    const size_t px = (size_t)field * c->S;                 // this field's pixels in d_sci / d_var
    struct { double* host; int slot, op; } outs[] = {{raw, 0, 0}, {conv, 1, 0}, {resid, 1, 1},
                                                     {ivm, 2, 2}, {ps_sub, 3, 1}};
    int rc = PSFMC_OK;
    for (auto& o : outs) {
        if (!o.host) continue;
        hipLaunchKernelGGL(k_accumulated_out, dim3(256), dim3(256), 0, c->stream,
                           c->d_acc + ((size_t)field * 4 + o.slot) * c->S, c->d_sci + px, c->d_var + px, d_out,
                           inv_n, o.op, img_window(c));
        if (hipMemcpyAsync(o.host, d_out, S_img * sizeof(double), hipMemcpyDeviceToHost, c->stream) !=
                hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) {
            rc = fail(PSFMC_EHIP, "accumulated image copy failed");
            break;
        }
    }
    (void)hipFree(d_out);
    return rc;
}

extern "C" int psfmc_get_accumulated(psfmc_ctx* c, double* raw, double* conv, double* resid, double* ivm,
                                     double* ps_sub, long long* count) {
    return get_accumulated_impl(c, 0, raw, conv, resid, ivm, ps_sub, count);
}

// the posterior images of one field of a psfmc_ctx_create_fields context
extern "C" int psfmc_get_accumulated_field(psfmc_ctx* c, int field, double* raw, double* conv, double* resid,
                                           double* ivm, double* ps_sub, long long* count) {
    return get_accumulated_impl(c, field, raw, conv, resid, ivm, ps_sub, count);
}

extern "C" int psfmc_get_spectra(psfmc_ctx* c, double* psf_spec, double* var_spec) {
    if (!c || !psf_spec || !var_spec) return fail(PSFMC_EINVAL, "NULL argument");
    if (c->embed) return fail(PSFMC_EINVAL, "the image is embedded in a %d x %d transform: its kernel spectra are not "
                              "those of the %d x %d image", c->ny, c->nx, c->ly, c->lx);
    HIP_TRY(hipSetDevice(c->device));
    const size_t bytes = (size_t)c->n_psf * c->F * sizeof(double2);
    if (c->backend == PSFMC_BACKEND_HIPFFT) {
        HIP_TRY(hipMemcpy(psf_spec, c->d_pspec, bytes, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(var_spec, c->d_vspec, bytes, hipMemcpyDeviceToHost));
        return PSFMC_OK;
    }
    cd* tmp = nullptr;
    HIP_TRY(hipMalloc(&tmp, bytes));
    int rc = PSFMC_OK;
    for (int comp = 0; comp < 2 && rc == PSFMC_OK; ++comp) {
        hipLaunchKernelGGL(k_untranspose_spectrum, dim3(256), dim3(256), 0, c->stream, c->d_Kraw, tmp,
                           c->n_psf, comp, c->ny, c->nxh, c->rg_log2, c->d_rho);
        if (hipStreamSynchronize(c->stream) != hipSuccess ||
            hipMemcpy(comp ? var_spec : psf_spec, tmp, bytes, hipMemcpyDeviceToHost) != hipSuccess)
            rc = fail(PSFMC_EHIP, "spectrum copy failed");
    }
    (void)hipFree(tmp);
    return rc;
}

// ---------------------------------------------------------------------------
// psfmc_group: ONE process driving several GPUs: a context per
// device with the field replicated, walkers split into contiguous blocks, every device's
// upload / evaluation / download enqueued on its own stream before any is waited for, the
// blocks landing at their offsets of the caller's host array (for a host-side result that
// is the whole "gather"; the one-process-per-GPU form with an RCCL all-gather of device
// tensors is psfmc_amd/parallel.py).
// ---------------------------------------------------------------------------
struct psfmc_group {
    std::vector<psfmc_ctx*> ctx;
};

static void group_block(int W, int n, int r, int* lo, int* hi) {
    const int base = W / n, extra = W % n;
    *lo = r * base + (r < extra ? r : extra);
    *hi = *lo + base + (r < extra ? 1 : 0);
}

extern "C" int psfmc_group_create(psfmc_group** out, int n_dev, const int* devices, int ny, int nx,
                                  const double* sci, const double* obs_var, const uint8_t* bad_px,
                                  int n_psf, int psf_ny, int psf_nx, const double* psf,
                                  const double* psf_var, int n_ps, int n_sersic, int max_walkers,
                                  int backend) {
    if (!out) return fail(PSFMC_EINVAL, "out is NULL");
    *out = nullptr;
    if (n_dev < 1 || n_dev > 64 || !devices) return fail(PSFMC_EINVAL, "need 1..64 devices");
    psfmc_group* g = new psfmc_group;
    // every device may be handed the whole batch's share rounded up
    const int per_dev = (max_walkers + n_dev - 1) / n_dev;
    for (int i = 0; i < n_dev; ++i) {
        psfmc_ctx* c = nullptr;
        const int rc = psfmc_ctx_create(&c, devices[i], ny, nx, sci, obs_var, bad_px, n_psf, psf_ny, psf_nx,
                                        psf, psf_var, n_ps, n_sersic, per_dev < 1 ? 1 : per_dev, backend);
        if (rc != PSFMC_OK) {
            const std::string keep = g_err;
            for (psfmc_ctx* d : g->ctx) psfmc_ctx_destroy(d);
            delete g;
            g_err = keep;
            return rc;
        }
        g->ctx.push_back(c);
    }
    *out = g;
    return PSFMC_OK;
}

extern "C" int psfmc_group_destroy(psfmc_group* g) {
    if (!g) return PSFMC_OK;
    for (psfmc_ctx* c : g->ctx) psfmc_ctx_destroy(c);
    delete g;
    return PSFMC_OK;
}

extern "C" int psfmc_group_size(const psfmc_group* g) { return g ? (int)g->ctx.size() : PSFMC_EINVAL; }

extern "C" int psfmc_group_set_layout(psfmc_group* g, int n_sky, int n_params, const int* slot_col,
                                      const double* slot_const, const int* ps_method,
                                      const int* sersic_degrees, double mag_zeropoint, const int* family,
                                      const double* p0, const double* p1, const double* p2) {
    if (!g) return fail(PSFMC_EINVAL, "group is NULL");
    for (psfmc_ctx* c : g->ctx)
        RC_TRY(psfmc_set_layout(c, n_sky, n_params, slot_col, slot_const, ps_method, sersic_degrees,
                                mag_zeropoint, family, p0, p1, p2));
    return PSFMC_OK;
}

// rows != nullptr: log-likelihoods of derived rows; else log-posteriors of raw vectors
static int group_eval(psfmc_group* g, int W, const double* rows, const uint8_t* skip, const double* theta,
                      const double* extra, double* out) {
    if (!g) return fail(PSFMC_EINVAL, "group is NULL");
    if (W < 0 || (W > 0 && (!out || (!rows && !theta)))) return fail(PSFMC_EINVAL, "bad argument");
    const int n = (int)g->ctx.size();
    int rc = PSFMC_OK;
    // one device's share: enqueue its copies and kernels.  An error ends the enqueueing but NEVER
    // skips the synchronize loop below: earlier devices still have copies in flight out of `theta`
    // / `rows` and into `out`, which the caller may free as soon as this function returns.
    auto enqueue = [&](int r) -> int {
        int lo, hi;
        group_block(W, n, r, &lo, &hi);
        psfmc_ctx* c = g->ctx[r];
        const int w = hi - lo;
        if (w == 0) return PSFMC_OK;
        if (w > c->max_walkers) return fail(PSFMC_EINVAL, "W=%d exceeds the group's max_walkers", W);
        HIP_TRY(hipSetDevice(c->device));
        hipStream_t st = c->stream;
        if (rows) {
            HIP_TRY(hipMemcpyAsync(c->d_rows, rows + (size_t)lo * c->rlen, (size_t)w * c->rlen * sizeof(double),
                                   hipMemcpyHostToDevice, st));
            if (skip) HIP_TRY(hipMemcpyAsync(c->d_skip, skip + lo, (size_t)w, hipMemcpyHostToDevice, st));
            RC_TRY(eval_device(c, w, c->d_rows, skip ? c->d_skip : nullptr, c->d_like, st));
        } else {
            RC_TRY(check_theta_call(c, w, theta, out));
            const int P = c->layout.n_params;
            if (P) HIP_TRY(hipMemcpyAsync(c->d_theta, theta + (size_t)lo * P, (size_t)w * P * sizeof(double),
                                          hipMemcpyHostToDevice, st));
            if (extra) HIP_TRY(hipMemcpyAsync(c->d_extra, extra + lo, (size_t)w * sizeof(double),
                                              hipMemcpyHostToDevice, st));
            RC_TRY(eval_theta_device(c, w, c->d_theta, extra ? c->d_extra : nullptr, c->d_like, st));
        }
        HIP_TRY(hipMemcpyAsync(out + lo, c->d_like, (size_t)w * sizeof(double), hipMemcpyDeviceToHost, st));
        return PSFMC_OK;
    };
    std::string first_error;
    for (int r = 0; r < n && rc == PSFMC_OK; ++r) {          // enqueue everything ...
        rc = enqueue(r);
        if (rc != PSFMC_OK) first_error = psfmc_last_error();
    }
    for (int r = 0; r < n; ++r) {                             // ... then wait for every device
        psfmc_ctx* c = g->ctx[r];
        if (hipSetDevice(c->device) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess)
            if (rc == PSFMC_OK) rc = fail(PSFMC_EHIP, "device %d failed: %s", c->device,
                                          hipGetErrorString(hipGetLastError()));
    }
    if (!first_error.empty()) (void)fail(rc, "%s", first_error.c_str());   // the enqueue error is the one to report
    return rc;
}

extern "C" int psfmc_group_eval_batch(psfmc_group* g, int W, const double* rows, const uint8_t* skip,
                                      double* loglike) {
    if (W > 0 && !rows) return fail(PSFMC_EINVAL, "NULL rows");
    return group_eval(g, W, rows, skip, nullptr, nullptr, loglike);
}

extern "C" int psfmc_group_eval_theta(psfmc_group* g, int W, const double* theta, const double* extra_lnprior,
                                      double* lnprob) {
    if (W > 0 && !theta) return fail(PSFMC_EINVAL, "NULL theta");
    return group_eval(g, W, nullptr, nullptr, theta, extra_lnprior, lnprob);
}

// ---------------------------------------------------------------------------
// diagnostic: device elementary functions on host arrays
// ---------------------------------------------------------------------------
__global__ void k_debug_math(int op, int n, const double* __restrict__ in, double* __restrict__ out) {
    __shared__ __align__(16) double tab[4][kLogTabBytes / sizeof(double)];   // one copy per wave, as in k_rows_fwd
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (op >= 100) {
        // op = 100: x^p through the rasteriser's power tables, p = in[n]; the tables are built in `out + n`
        // (kPowTab doubles of scratch behind the n results) by this wave, as k_pow_tables builds them
        __shared__ __align__(16) double ptab[4][kRasterLdsDoubles];
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        const double p = in[n];
        double* g = out + n + (size_t)(blockIdx.x * 4 + wave) * kPowTab;
        build_pow_table(p, g, lane);
        __threadfence_block();
        wave_lds_sync();
        load_a_table(ptab[wave], lane);
        load_pow_table(ptab[wave], g, lane);
        wave_lds_sync();
        if (i < n) out[i] = fast_pow_tab(in[i], pow_poly(p), ptab[wave]);
        return;
    }
    load_log_table(tab[threadIdx.x >> 6], threadIdx.x & 63);
    wave_lds_sync();
    if (i >= n) return;
    const double x = in[i];
    out[i] = op == 0 ? fast_log2(x) : op == 1 ? fast_exp2(x) : op == 2 ? fast_rcp(x) : op == 3 ? fast_rcp1(x)
             : op == 4 ? fast_exp2_noclamp(x) : op == 5 ? fast_log2_tab(x, tab[threadIdx.x >> 6]) : fast_exp2_floor(x);
}

// ---------------------------------------------------------------------------
// Diagnostic: the plain memory sweeps the three kernels of a pass correspond to (no arithmetic),
// timed with HIP events -- what this GPU gives the data flow at best.  mode 0: every 16 bytes
// written once (k_rows_fwd), 1: read and written back in place (k_cols), 2: read once (k_rows_inv).
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_sweep(double2* __restrict__ buf, size_t n, int mode, double* __restrict__ sink) {
    const size_t stride = (size_t)gridDim.x * 1024;
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i + 768 < n; i += stride) {
        if (mode == 0) {
            const double2 v = {1.0, 2.0};
            buf[i] = v; buf[i + 256] = v; buf[i + 512] = v; buf[i + 768] = v;
        } else {
            double2 a = buf[i], b = buf[i + 256], c = buf[i + 512], d = buf[i + 768];
            if (mode == 1) {
                a.x += 1.0; b.x += 1.0; c.x += 1.0; d.x += 1.0;
                buf[i] = a; buf[i + 256] = b; buf[i + 512] = c; buf[i + 768] = d;
            } else {
                s += a.x + b.x + c.x + d.x;
            }
        }
    }
    if (s == 12345.678) sink[0] = s;            // keeps the loads of mode 2
}

extern "C" int psfmc_debug_sweep(int device, int mode, size_t nbytes, int reps, double* us_per_sweep) {
    if (mode < 0 || mode > 2 || nbytes < (1u << 20) || reps < 1 || !us_per_sweep) return fail(PSFMC_EINVAL, "bad argument");
    HIP_TRY(hipSetDevice(device));
    double2* buf = nullptr;
    double* sink = nullptr;
    HIP_TRY(hipMalloc(&buf, nbytes));
    int rc = PSFMC_OK;
    hipEvent_t a = nullptr, b = nullptr;
    if (hipMalloc(&sink, 64) != hipSuccess || hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
        rc = fail(PSFMC_ENOMEM, "hipMalloc / hipEventCreate");
    } else {
        (void)hipMemset(buf, 0, nbytes);
        const size_t n = nbytes / sizeof(double2);
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_sweep, dim3(2048), dim3(256), 0, 0, buf, n, mode, sink);
        (void)hipEventRecord(a, 0);
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_sweep, dim3(2048), dim3(256), 0, 0, buf, n, mode, sink);
        (void)hipEventRecord(b, 0);
        float ms = 0.f;
        if (hipEventSynchronize(b) != hipSuccess || hipEventElapsedTime(&ms, a, b) != hipSuccess)
            rc = fail(PSFMC_EHIP, "sweep failed: %s", hipGetErrorString(hipGetLastError()));
        *us_per_sweep = (double)ms * 1e3 / reps;
    }
    if (a) (void)hipEventDestroy(a);
    if (b) (void)hipEventDestroy(b);
    if (sink) (void)hipFree(sink);
    (void)hipFree(buf);
    return rc;
}

// ---------------------------------------------------------------------------
// diagnostic: the rate at which THIS chip issues fp64 vector instructions when every SIMD is busy -- the
// other ceiling of the path besides the memory sweeps (psfmc_debug_sweep): with two and four Sersic components
// the 512^2 / 1024^2 passes carry more VALU time than sweep time.  64 v_fma_f64 per loop iteration (8 chains
// x 8, inline asm), `waves` waves per SIMD on every CU, timed with HIP events: nanoseconds per wave-instruction
// per SIMD.  (tools/clock_probe.hip is the stand-alone version with the in-kernel clock.)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_valu_rate(double* __restrict__ out, int iters) {
    double a[8];
    for (int i = 0; i < 8; ++i) a[i] = 1.0 + 1e-9 * (threadIdx.x + i);
    double m = 1.0000001, c = 1e-12;
    asm volatile("" : "+v"(m), "+v"(c));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
    }
    double acc = 0.0;
    for (int i = 0; i < 8; ++i) acc += a[i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

extern "C" int psfmc_debug_valu_rate(int device, int waves_per_simd, int iters, double* ns_per_instruction) {
    if (waves_per_simd < 1 || waves_per_simd > 8 || iters < 1 || !ns_per_instruction) return fail(PSFMC_EINVAL, "bad argument");
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    const int blocks = prop.multiProcessorCount * waves_per_simd;        // 4 waves per block: one per SIMD
    double* out = nullptr;
    HIP_TRY(hipMalloc(&out, (size_t)blocks * 256 * sizeof(double)));
    int rc = PSFMC_OK;
    hipEvent_t a = nullptr, b = nullptr;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
        rc = fail(PSFMC_ENOMEM, "hipEventCreate");
    } else {
        hipLaunchKernelGGL(k_valu_rate, dim3(blocks), dim3(256), 0, 0, out, iters);          // warm-up (clocks settle)
        (void)hipEventRecord(a, 0);
        hipLaunchKernelGGL(k_valu_rate, dim3(blocks), dim3(256), 0, 0, out, iters);
        (void)hipEventRecord(b, 0);
        float ms = 0.f;
        if (hipEventSynchronize(b) != hipSuccess || hipEventElapsedTime(&ms, a, b) != hipSuccess)
            rc = fail(PSFMC_EHIP, "rate probe failed: %s", hipGetErrorString(hipGetLastError()));
        // every SIMD issued waves_per_simd x iters x 64 wave-instructions
        *ns_per_instruction = (double)ms * 1e6 / ((double)waves_per_simd * iters * 64.0);
    }
    if (a) (void)hipEventDestroy(a);
    if (b) (void)hipEventDestroy(b);
    (void)hipFree(out);
    return rc;
}

// op 100: in[n] holds the exponent p of x^p through the rasteriser's power tables (in has n + 1 values)
extern "C" int psfmc_debug_math(int device, int op, int n, const double* in, double* out) {
    const bool pw = op == 100;
    if (n < 0 || op < 0 || (op > 6 && !pw) || (n > 0 && (!in || !out))) return fail(PSFMC_EINVAL, "bad argument");
    if (n == 0) return PSFMC_OK;
    HIP_TRY(hipSetDevice(device));
    double *d_in = nullptr, *d_out = nullptr;
    const int blocks = (n + 255) / 256;
    const size_t n_in = (size_t)n + (pw ? 1 : 0), n_out = (size_t)n + (pw ? (size_t)blocks * 4 * kPowTab : 0);
    HIP_TRY(hipMalloc(&d_in, n_in * sizeof(double)));
    int rc = PSFMC_OK;
    if (hipMalloc(&d_out, n_out * sizeof(double)) != hipSuccess) rc = fail(PSFMC_ENOMEM, "hipMalloc");
    if (rc == PSFMC_OK) {
        (void)hipMemcpy(d_in, in, n_in * sizeof(double), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_debug_math, dim3(blocks), dim3(256), 0, 0, op, n, d_in, d_out);
        if (hipMemcpy(out, d_out, (size_t)n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
            rc = fail(PSFMC_EHIP, "debug_math copy failed: %s", hipGetErrorString(hipGetLastError()));
    }
    (void)hipFree(d_in);
    if (d_out) (void)hipFree(d_out);
    return rc;
}

#endif  // PSFMC_PART == 0
