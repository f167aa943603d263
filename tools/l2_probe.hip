// Probe: bandwidth a kernel sees when every XCD re-reads (and re-writes) a private region
// small enough for its 4 MiB L2, vs Infinity-Cache- and HBM-sized regions.  Workgroup b uses
// region b % 8 (workgroups are dealt to the 8 XCDs round-robin).
// build: hipcc --offload-arch=gfx950 -O2 -o _ab/l2_probe tools/l2_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(256) void k_read(const double2* __restrict__ buf, size_t region_elems, int reps,
                                              double* out, int* xcc) {
    const int xcd = blockIdx.x & 7, lw = blockIdx.x >> 3, nlw = gridDim.x >> 3;
    const double2* r = buf + (size_t)xcd * region_elems;
    const size_t slice = region_elems / nlw;
    double s = 0;
    for (int rep = 0; rep < reps; ++rep) {
        const double2* p = r + (size_t)((lw + rep) % nlw) * slice;
        for (size_t i = threadIdx.x; i < slice; i += 256 * 4) {
            double2 a = p[i], b = p[i + 256], c = p[i + 512], d = p[i + 768];
            s += a.x + b.y + c.x + d.y;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        xcc[blockIdx.x] = id & 0xf;
    }
}

__global__ __launch_bounds__(256) void k_rw(double2* __restrict__ buf, size_t region_elems, int reps, double* out) {
    const int xcd = blockIdx.x & 7, lw = blockIdx.x >> 3, nlw = gridDim.x >> 3;
    double2* r = buf + (size_t)xcd * region_elems;
    const size_t slice = region_elems / nlw;
    for (int rep = 0; rep < reps; ++rep) {
        double2* p = r + (size_t)((lw + rep) % nlw) * slice;   // races between workgroups do not matter here
        for (size_t i = threadIdx.x; i < slice; i += 256 * 4) {
            double2 a = p[i], b = p[i + 256], c = p[i + 512], d = p[i + 768];
            a.x += 1.0; b.x += 1.0; c.x += 1.0; d.x += 1.0;
            p[i] = a; p[i + 256] = b; p[i + 512] = c; p[i + 768] = d;
        }
    }
    if (threadIdx.x == 0) out[blockIdx.x] = 0;
}

int main() {
    const size_t max_bytes = (size_t)8 << 30;
    double2* buf; double* out; int* xcc;
    if (hipMalloc(&buf, max_bytes) != hipSuccess) return 1;
    hipMemset(buf, 0, max_bytes);
    hipMalloc(&out, 1 << 24); hipMalloc(&xcc, 1 << 16);
    const size_t region_mb[] = {1, 2, 3, 4, 8, 24, 64, 512};
    for (int wg_per_cu = 1; wg_per_cu <= 8; wg_per_cu *= 2) {
        const int grid = 256 * wg_per_cu;
        for (size_t mb : region_mb) {
            const size_t region_elems = (mb << 20) / 16;
            if (region_elems / (grid / 8) < 1024) continue;
            int reps = (int)(4096 / mb); if (reps < 4) reps = 4; if (reps > 400) reps = 400;
            for (int mode = 0; mode < 2; ++mode) {
                hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
                if (mode == 0) k_read<<<grid, 256>>>(buf, region_elems, 2, out, xcc);
                else k_rw<<<grid, 256>>>(buf, region_elems, 2, out);
                hipDeviceSynchronize();
                hipEventRecord(a);
                if (mode == 0) k_read<<<grid, 256>>>(buf, region_elems, reps, out, xcc);
                else k_rw<<<grid, 256>>>(buf, region_elems, reps, out);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                const double bytes = (double)reps * (mb << 20) * 8 * (mode ? 2 : 1);
                printf("wg/CU %d  region %4zu MiB per XCD  %s  reps %3d  %8.3f ms  %8.0f GB/s\n", wg_per_cu, mb,
                       mode ? "read+write" : "read      ", reps, ms, bytes / ms / 1e6);
            }
        }
    }
    int h[64];
    hipMemcpy(h, xcc, sizeof h, hipMemcpyDeviceToHost);
    printf("XCC_ID of workgroups 0..31:");
    for (int i = 0; i < 32; ++i) printf(" %d", h[i]);
    printf("\n");
    return 0;
}
