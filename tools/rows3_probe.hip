// rows3_probe.hip -- the one-row-per-wave three-stage row kernels (psfmc_rows3_path.h) against the two-stage row
// kernels (psfmc_fused_path.h) on the same prep records: values (forward: every element of T; inverse: every
// walker's chi^2 sum) and time per launch at a pass-sized batch.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -DSIDES='X(1024,6,4) X(512,24,2)' tools/rows3_probe.hip -o build/probe/rows3_probe
//        X(side, walkers per launch, Sersic components)
#include "../psfmc_amd/csrc/psfmc_rows3_path.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace psfmc;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void k_probe_tables(double* prep, int n_pairs, int n_ps, int n_sersic) {
    const int pair = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (pair >= n_pairs) return;
    const int w = pair / n_sersic, k = pair - w * n_sersic;
    double* rec = prep + (size_t)w * prep_len(n_ps, n_sersic);
    const double p = rec[kPrepHead + kPrepPs * n_ps + kPrepSersic * k + 7];
    build_pow_table(p, rec + prep_rec_len(n_ps, n_sersic) + (size_t)k * kPowTab, lane);
}

static double urand() { return rand() / (double)RAND_MAX; }

template <int NX>
static void run_side(int n_w, int n_sersic, int reps) {
    using S3 = typename Rows3<NX>::S;
    constexpr int NY = NX, NXH = NX / 2 + 1;
    constexpr bool FAST = FftShape<NX>::kPlain;             // the unguarded power-of-two kernels, or the general ones
    constexpr int rgl2_old = layout_rg_log2<NX, FAST>();
    const int n_ps = 1, plen = prep_len(n_ps, n_sersic);
    std::vector<double> hprep((size_t)n_w * plen, 0.0);
    srand(11);
    for (int w = 0; w < n_w; ++w) {
        double* r = &hprep[(size_t)w * plen];
        r[0] = 0.01 * urand();
        r[kPrepPsfIdx] = 0.0;
        r[kPrepMu] = 0.25;
        r[kPrepInvLambda] = 2.0;
        double* ps = r + kPrepHead;
        ps[0] = NY / 2 - 3 + (int)(4 * urand()); ps[1] = 7; ps[2] = NX / 2 - 3 + (int)(4 * urand()); ps[3] = 7;
        for (int j = 0; j < 7; ++j) { ps[4 + j] = urand() - 0.3; ps[4 + kTaps + j] = 3.0 * (urand() - 0.3); }
        double* se = ps + kPrepPs;
        for (int k = 0; k < n_sersic; ++k, se += kPrepSersic) {
            const double n = 0.6 + 5.0 * urand(), reff = 5.0 + 40.0 * urand(), reff_b = reff * (0.3 + 0.6 * urand());
            const double th = M_PI * urand();
            se[0] = NX / 2 + 16.0 * (urand() - 0.5); se[1] = NY / 2 + 16.0 * (urand() - 0.5);
            se[2] = cos(th) / reff; se[3] = sin(th) / reff; se[4] = -sin(th) / reff_b; se[5] = cos(th) / reff_b;
            se[6] = 2.0 * n - 1.0 / 3.0; se[7] = 0.5 / n; se[8] = 0.5 + urand();
        }
    }
    std::vector<cd> tw(NX);
    for (int j = 0; j < NX; ++j) tw[j] = cd{(double)cosl(2.0L * M_PIl * j / NX), (double)-sinl(2.0L * M_PIl * j / NX)};
    std::vector<double> sci((size_t)NY * NX), var((size_t)NY * NX);
    std::vector<uint8_t> bad((size_t)NY * NX, 0);
    for (size_t i = 0; i < sci.size(); ++i) { sci[i] = urand(); var[i] = 0.5 + urand(); bad[i] = urand() < 0.001; }

    const int nyp_old = t_col_len(NY, rgl2_old), nyp_new = t_col_len(NY, rows3_rg_log2(NX));
    const size_t per_old = (size_t)2 * NXH * nyp_old, per_new = (size_t)2 * NXH * nyp_new;
    double *dprep, *dsci, *dvar, *dpart_old, *dpart_new;
    uint8_t* dbad;
    cd *dtw, *dT_old, *dT_new;
    FieldPx *dfield_old, *dfield_new;
    const int nblk_old = (NY + row_group<NX>() - 1) / row_group<NX>();
    CK(hipMalloc(&dprep, hprep.size() * sizeof(double)));
    CK(hipMemcpy(dprep, hprep.data(), hprep.size() * sizeof(double), hipMemcpyHostToDevice));
    CK(hipMalloc(&dtw, NX * sizeof(cd)));
    CK(hipMemcpy(dtw, tw.data(), NX * sizeof(cd), hipMemcpyHostToDevice));
    CK(hipMalloc(&dT_old, per_old * n_w * sizeof(cd)));
    CK(hipMalloc(&dT_new, per_new * n_w * sizeof(cd)));
    CK(hipMemset(dT_old, 0, per_old * n_w * sizeof(cd)));
    CK(hipMemset(dT_new, 0, per_new * n_w * sizeof(cd)));
    CK(hipMalloc(&dsci, sci.size() * 8)); CK(hipMalloc(&dvar, var.size() * 8)); CK(hipMalloc(&dbad, bad.size()));
    CK(hipMemcpy(dsci, sci.data(), sci.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dvar, var.data(), var.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dbad, bad.data(), bad.size(), hipMemcpyHostToDevice));
    CK(hipMalloc(&dfield_old, fused_field_len<NX>(NY) * sizeof(FieldPx)));
    CK(hipMalloc(&dfield_new, rows3_field_len<NX>(NY) * sizeof(FieldPx)));
    CK(hipMalloc(&dpart_old, (size_t)n_w * nblk_old * 8));
    CK(hipMalloc(&dpart_new, (size_t)n_w * NY * 8));
    if (n_sersic > 0) hipLaunchKernelGGL(k_probe_tables, dim3((n_w * n_sersic + 3) / 4), dim3(256), 0, 0, dprep, n_w * n_sersic, n_ps, n_sersic);
    hipLaunchKernelGGL((k_pack_field<NX>), dim3(256), dim3(256), 0, 0, dsci, dvar, dbad, dfield_old, NY);
    hipLaunchKernelGGL((k_pack_field3<NX>), dim3(256), dim3(256), 0, 0, dsci, dvar, dbad, dfield_new, NY);
    CK(hipDeviceSynchronize());

    constexpr size_t lds_old = fused_row_lds_bytes<NX, FAST>(), lds_new = rows3_lds_bytes<S3>();
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rows_fwd<NX, false, cd, FAST, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_old));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rows_inv<NX, cd, FAST, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_old));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rows3_fwd<NX, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_new));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rows3_inv<NX, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_new));
    constexpr int waves_old = row_waves<NX, FAST>();
    const WrapDesc wr{0, 0, 0, 0, 0, 0};
    auto fwd_old = [&]() {
        hipLaunchKernelGGL((k_rows_fwd<NX, false, cd, FAST, false>), dim3((nblk_old + waves_old - 1) / waves_old, n_w), dim3(64 * waves_old), lds_old, 0,
                           dprep, (const uint8_t*)nullptr, dtw, dT_old, n_ps, n_sersic, NY, 0, (const double*)nullptr,
                           (const double*)nullptr, (double*)nullptr, wr, kPowTabsBuilt);
    };
    auto fwd_new = [&]() {
        hipLaunchKernelGGL((k_rows3_fwd<NX, false, false>), dim3((NY + rows3_waves(NX) - 1) / rows3_waves(NX), n_w), dim3(rows3_threads(NX)), lds_new, 0, dprep,
                           (const uint8_t*)nullptr, dtw, dT_new, n_ps, n_sersic, NY, 0, (const double*)nullptr,
                           (const double*)nullptr, (double*)nullptr, wr, kPowTabsBuilt);
    };
    auto inv_old = [&]() {
        hipLaunchKernelGGL((k_rows_inv<NX, cd, FAST, false>), dim3((nblk_old + waves_old - 1) / waves_old, n_w), dim3(64 * waves_old), lds_old, 0,
                           dT_old, (const uint8_t*)nullptr, dtw, dfield_old, dpart_old, NY, dprep, plen, (double*)nullptr,
                           (double*)nullptr, 0, 0u);
    };
    auto inv_new = [&]() {
        hipLaunchKernelGGL((k_rows3_inv<NX, false>), dim3((NY + rows3_waves(NX) - 1) / rows3_waves(NX), n_w), dim3(rows3_threads(NX)), lds_new, 0, dT_new,
                           (const uint8_t*)nullptr, dtw, dfield_new, dpart_new, NY, dprep, plen, (double*)nullptr,
                           (double*)nullptr, 0, 0u);
    };
    fwd_old(); fwd_new();
    CK(hipDeviceSynchronize());
    CK(hipGetLastError());
    // forward values
    std::vector<cd> a(per_old * n_w), b(per_new * n_w);
    CK(hipMemcpy(a.data(), dT_old, a.size() * sizeof(cd), hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), dT_new, b.size() * sizeof(cd), hipMemcpyDeviceToHost));
    double worst = 0.0, scale = 0.0;
    auto el = [](int y, int c, int rgl2) { return (size_t)((((y >> rgl2) * 2 + c) << rgl2) + (y & ((1 << rgl2) - 1))); };
    for (int w = 0; w < n_w; ++w)
        for (int kx = 0; kx < NXH; ++kx)
            for (int y = 0; y < NY; ++y)
                for (int c = 0; c < 2; ++c) {
                    const cd p = a[per_old * w + (size_t)kx * 2 * nyp_old + el(y, c, rgl2_old)];
                    const cd q = b[per_new * w + (size_t)kx * 2 * nyp_new + el(y, c, rows3_rg_log2(NX))];
                    worst = fmax(worst, fmax(fabs(p.x - q.x), fabs(p.y - q.y)));
                    scale = fmax(scale, fmax(fabs(p.x), fabs(p.y)));
                }
    printf("side %4d  fwd: max |two-stage - three-stage| %.3e of %.3e  %s\n", NX, worst, scale,
           worst <= 1e-12 * scale && scale > 0 ? "ok" : "WRONG");
    // inverse values (on the forward output as it is: any spectrum will do)
    inv_old(); inv_new();
    CK(hipDeviceSynchronize());
    CK(hipGetLastError());
    std::vector<double> po((size_t)n_w * nblk_old), pn((size_t)n_w * NY);
    CK(hipMemcpy(po.data(), dpart_old, po.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(pn.data(), dpart_new, pn.size() * 8, hipMemcpyDeviceToHost));
    double worst_i = 0.0;
    for (int w = 0; w < n_w; ++w) {
        long double so = 0, sn = 0;
        for (int i = 0; i < nblk_old; ++i) so += po[(size_t)w * nblk_old + i];
        for (int i = 0; i < NY; ++i) sn += pn[(size_t)w * NY + i];
        worst_i = fmax(worst_i, fabs((double)((so - sn) / so)));
    }
    printf("side %4d  inv: max relative difference of the walkers' chi^2 sums %.3e  %s\n", NX, worst_i,
           worst_i <= 1e-11 ? "ok" : "WRONG");
    // timing
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time_it = [&](auto&& fn) {
        for (int i = 0; i < 3; ++i) fn();
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) fn();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        return ms * 1e3 / reps;
    };
    const double t_fo = time_it(fwd_old), t_fn = time_it(fwd_new), t_io = time_it(inv_old), t_in = time_it(inv_new);
    printf("side %4d  %2d walkers, %d Sersic:  fwd two-stage %6.1f us  three-stage %6.1f us   inv two-stage %6.1f us  three-stage %6.1f us\n",
           NX, n_w, n_sersic, t_fo, t_fn, t_io, t_in);
    fflush(stdout);
    CK(hipFree(dprep)); CK(hipFree(dtw)); CK(hipFree(dT_old)); CK(hipFree(dT_new)); CK(hipFree(dsci)); CK(hipFree(dvar));
    CK(hipFree(dbad)); CK(hipFree(dfield_old)); CK(hipFree(dfield_new)); CK(hipFree(dpart_old)); CK(hipFree(dpart_new));
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 20;
#define X(n, w, s) run_side<n>(w, s, reps);
    SIDES
#undef X
    return 0;
}
