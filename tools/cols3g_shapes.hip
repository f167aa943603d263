// cols3g_shapes.hip -- times (and checks against a direct DFT) the column kernel k_cols3g for chosen
// factorisations N = R1 (R2 R3) of a side: which shape psfmc_fft.h fft3g_pick should prefer is an empirical
// question (registers vs idle lanes vs idle stage slots).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -DSHAPES='X(600,6,10) X(600,5,12)' tools/cols3g_shapes.hip -o build/probe/cols3g_shapes
// (X(n, 0, 0) = the two-stage kernel k_cols<n> for comparison)
#include "../psfmc_amd/csrc/psfmc_fused_path.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace psfmc;
#ifndef NXH_LESS
#define NXH_LESS 0   /* 1: one kx column fewer (what a packed DC + Nyquist column would leave): the tail-round experiment */
#endif

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int N, int R2, int R3>
static void run_one(size_t t_budget_bytes) {
    constexpr int rg_log2 = 2;
    const int nxh = N / 2 + 1 - NXH_LESS, nyp = t_col_len(N, rg_log2), plen = 8;
    const size_t per_w = (size_t)nxh * 2 * nyp;
    int n_w = (int)(t_budget_bytes / (per_w * sizeof(cd)));
    if (n_w < 1) n_w = 1;
    std::vector<cd> hT(per_w), hK((size_t)nxh * 2 * N), tw(N);
    srand(7);
    for (auto& z : hT) z = cd{rand() / (double)RAND_MAX - 0.5, rand() / (double)RAND_MAX - 0.5};
    for (auto& z : hK) z = cd{rand() / (double)RAND_MAX - 0.5, rand() / (double)RAND_MAX - 0.5};
    for (int j = 0; j < N; ++j) tw[j] = cd{cos(2.0 * M_PI * j / N), -sin(2.0 * M_PI * j / N)};
    cd *dT, *dK, *dtw;
    double* dprep;
    CK(hipMalloc(&dT, per_w * n_w * sizeof(cd)));
    CK(hipMalloc(&dK, hK.size() * sizeof(cd)));
    CK(hipMalloc(&dtw, N * sizeof(cd)));
    CK(hipMalloc(&dprep, (size_t)n_w * plen * sizeof(double)));
    CK(hipMemset(dprep, 0, (size_t)n_w * plen * sizeof(double)));
    CK(hipMemcpy(dK, hK.data(), hK.size() * sizeof(cd), hipMemcpyHostToDevice));
    CK(hipMemcpy(dtw, tw.data(), N * sizeof(cd), hipMemcpyHostToDevice));
    auto fill = [&]() { for (int w = 0; w < n_w; ++w) CK(hipMemcpy(dT + per_w * w, hT.data(), per_w * sizeof(cd), hipMemcpyHostToDevice)); };
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int n_cols = n_w * 2 * nxh;
    auto launch = [&]() {
        if constexpr (R2 == -2) {                  // X(n, -2, 0): k_cols3f<n> (512, 1024, 1536, 2048)
            using S = Fft3gShape<N, 8, 8>;
            constexpr size_t lds = fused_col3f_lds_bytes<S>();
            CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cols3f<N, true, S>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            const int wpb = cols3f_waves<S>(), blocks = (n_cols + wpb - 1) / wpb, cap = 8 * prop.multiProcessorCount;
            hipLaunchKernelGGL((k_cols3f<N, true, S>), dim3(blocks < cap ? blocks : cap), dim3(cols3f_threads<S>()), lds, 0, dT, dK, dprep,
                               (const uint8_t*)nullptr, dtw, plen, nxh, n_w, rg_log2);
        } else if constexpr (R2 < 0) {             // X(n, -1, 0): k_cols3<n> (512, 1024)
            constexpr size_t lds = fused_col3_lds_bytes<N>();
            CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cols3<N, true, cd>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            const int blocks = (n_cols + 3) / 4, cap = 8 * prop.multiProcessorCount;
            hipLaunchKernelGGL((k_cols3<N, true, cd>), dim3(blocks < cap ? blocks : cap), dim3(kColThreads), lds, 0, dT, dK, dprep,
                               (const uint8_t*)nullptr, dtw, plen, nxh, n_w, rg_log2);
        } else if constexpr (R2 > 0) {
            using S = Fft3gShape<N, R2, R3>;
            constexpr size_t lds = fused_col3g_lds_bytes<S>();
            CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cols3g<N, true, S>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            const int wpb = cols3g_waves<S>(), blocks = (n_cols + wpb - 1) / wpb, cap = 8 * prop.multiProcessorCount;
            hipLaunchKernelGGL((k_cols3g<N, true, S>), dim3(blocks < cap ? blocks : cap), dim3(cols3g_threads<S>()), lds, 0, dT, dK, dprep,
                               (const uint8_t*)nullptr, dtw, plen, nxh, n_w, rg_log2);
        } else if constexpr (two_stage_side(N)) {
            constexpr size_t lds = fused_col_lds_bytes<N>();
            CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cols<N, true, cd>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            const int groups = (n_cols + col_ffts_per_block<N>() - 1) / col_ffts_per_block<N>(), cap = 2 * prop.multiProcessorCount;
            hipLaunchKernelGGL((k_cols<N, true, cd>), dim3(groups < cap ? groups : cap), dim3(kColThreads), lds, 0, dT, dK, dprep,
                               (const uint8_t*)nullptr, dtw, plen, nxh, n_w, rg_log2);
        }
    };
    // correctness: one launch, columns (kx, c) = (3, 1) of the last walker against a direct DFT in long double
    fill();
    launch();
    CK(hipDeviceSynchronize());
    std::vector<cd> got(per_w);
    CK(hipMemcpy(got.data(), dT + per_w * (n_w - 1), per_w * sizeof(cd), hipMemcpyDeviceToHost));
    double worst = 0.0, scale = 0.0;
    for (int kx : {0, 3, nxh - 1}) for (int c = 0; c < 2; ++c) {
        auto off = [&](int y) { return (size_t)kx * 2 * nyp + (size_t)((((y >> rg_log2) * 2 + c) << rg_log2) + (y & ((1 << rg_log2) - 1))); };
        std::vector<long double> xr(N), xi(N), fr(N), fi(N);
        for (int y = 0; y < N; ++y) { xr[y] = hT[off(y)].x; xi[y] = hT[off(y)].y; }
        for (int k = 0; k < N; ++k) {
            long double sr = 0, si = 0;
            for (int y = 0; y < N; ++y) {
                const long double a = -2.0L * M_PIl * ((long long)y * k % N) / N, cr = cosl(a), ci = sinl(a);
                sr += xr[y] * cr - xi[y] * ci; si += xr[y] * ci + xi[y] * cr;
            }
            const cd kk = hK[((size_t)kx * 2 + c) * N + k];
            fr[k] = sr * kk.x - si * kk.y; fi[k] = sr * kk.y + si * kk.x;
        }
        for (int y = 0; y < N; ++y) {
            long double sr = 0, si = 0;
            for (int k = 0; k < N; ++k) {
                const long double a = 2.0L * M_PIl * ((long long)y * k % N) / N, cr = cosl(a), ci = sinl(a);
                sr += fr[k] * cr - fi[k] * ci; si += fr[k] * ci + fi[k] * cr;
            }
            worst = fmax(worst, fmax(fabs((double)(sr - got[off(y)].x)), fabs((double)(si - got[off(y)].y))));
            scale = fmax(scale, fmax(fabs((double)sr), fabs((double)si)));
        }
    }
    // timing
    for (int i = 0; i < 3; ++i) launch();
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 20;
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps, gbs = 2.0 * per_w * n_w * sizeof(cd) / (us * 1e-6) / 1e9;
    if (R2 == -2)
        printf("N %4d   k_cols3f                          walkers %4d  %7.1f us  %6.0f GB/s   max err %.2e of %.2e %s\n", N, n_w, us, gbs,
               worst, scale, worst < 1e-9 * scale ? "ok" : "WRONG");
    else if (R2 < 0)
        printf("N %4d   k_cols3 (waves %d, w2 in LDS %d)  walkers %4d  %7.1f us  %6.0f GB/s   max err %.2e of %.2e %s\n", N, PSFMC_COLS3_WAVES,
               PSFMC_COLS3_W2_LDS, n_w, us, gbs, worst, scale, worst < 1e-9 * scale ? "ok" : "WRONG");
    else if (R2 > 0)
        printf("N %4d = %2d x (%2d x %2d)  walkers %4d  %7.1f us  %6.0f GB/s   max err %.2e of %.2e %s\n", N, N / (R2 * R3 ? R2 * R3 : 1), R2, R3,
               n_w, us, gbs, worst, scale, worst < 1e-9 * scale ? "ok" : "WRONG");
    else if constexpr (two_stage_side(N))
        printf("N %4d   two-stage %2d x %2d    walkers %4d  %7.1f us  %6.0f GB/s   max err %.2e of %.2e %s\n", N, FftShape<N>::P, FftShape<N>::T,
               n_w, us, gbs, worst, scale, worst < 1e-9 * scale ? "ok" : "WRONG");
    fflush(stdout);
    CK(hipFree(dT)); CK(hipFree(dK)); CK(hipFree(dtw)); CK(hipFree(dprep));
}

int main(int argc, char** argv) {
    const size_t budget = (size_t)(argc > 1 ? atof(argv[1]) : 110.0) * 1000000;
#define X(n, r2, r3) run_one<n, r2, r3>(budget);
    SHAPES
#undef X
    return 0;
}
