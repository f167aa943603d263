// Probe: the rate of an in-place read-modify-write sweep (every byte read once and written once per
// launch, like the column kernel over T) against the buffer size -- Infinity-Cache-sized and larger.
// build: hipcc --offload-arch=gfx950 -O2 -o _ab/mall_probe tools/mall_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void k_rmw(double2* __restrict__ buf, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256 * 4;
    for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i + 768 < n; i += stride) {
        double2 a = buf[i], b = buf[i + 256], c = buf[i + 512], d = buf[i + 768];
        a.x += 1.0; b.x += 1.0; c.x += 1.0; d.x += 1.0;
        buf[i] = a; buf[i + 256] = b; buf[i + 512] = c; buf[i + 768] = d;
    }
}
__global__ __launch_bounds__(256) void k_read(const double2* __restrict__ buf, size_t n, double* out) {
    const size_t stride = (size_t)gridDim.x * 256 * 4;
    double s = 0;
    for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i + 768 < n; i += stride) {
        double2 a = buf[i], b = buf[i + 256], c = buf[i + 512], d = buf[i + 768];
        s += a.x + b.x + c.x + d.x;
    }
    if (s == 12345.678) out[0] = s;
}
__global__ __launch_bounds__(256) void k_write(double2* __restrict__ buf, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256 * 4;
    for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i + 768 < n; i += stride) {
        const double2 v = {1.0, 2.0};
        buf[i] = v; buf[i + 256] = v; buf[i + 512] = v; buf[i + 768] = v;
    }
}

int main() {
    const size_t max_bytes = (size_t)4 << 30;
    double2* buf; double* out;
    if (hipMalloc(&buf, max_bytes) != hipSuccess) return 1;
    hipMalloc(&out, 64);
    hipMemset(buf, 0, max_bytes);
    const size_t mbs[] = {32, 64, 112, 160, 224, 320, 448, 1024, 4096};
    for (int grid : {512, 1024, 2048, 4096}) {
        for (size_t mb : mbs) {
            const size_t n = (mb << 20) / 16;
            for (int mode = 0; mode < 3; ++mode) {
                hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
                const int reps = mb <= 448 ? 20 : 5;
                auto run = [&] {
                    if (mode == 0) k_rmw<<<grid, 256>>>(buf, n);
                    else if (mode == 1) k_read<<<grid, 256>>>(buf, n, out);
                    else k_write<<<grid, 256>>>(buf, n);
                };
                run(); run();
                hipDeviceSynchronize();
                hipEventRecord(a);
                for (int r = 0; r < reps; ++r) run();
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                const double bytes = (double)reps * (mb << 20) * (mode == 0 ? 2 : 1);
                printf("grid %4d  %5zu MiB  %s  %7.1f us/launch  %7.0f GB/s\n", grid, mb,
                       mode == 0 ? "read+write" : mode == 1 ? "read      " : "write     ", ms * 1e3 / reps, bytes / ms / 1e6);
            }
        }
    }
    return 0;
}
