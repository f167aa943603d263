// clock_probe.hip -- the shader clock the chip HOLDS under a dense fp64 VALU load, measured inside the
// kernel: delta s_memtime (shader cycles) / delta s_memrealtime (100 MHz), MI355X_MICROARCH.md "DVFS
// give-back" (6).  Build: hipcc -O3 --offload-arch=gfx950 tools/clock_probe.hip -o /tmp/clock_probe
// usage: clock_probe [waves_per_simd=2] [iters=200000] [mode: 0 registers only, 1 + a memory stream] [op: 0 fma, 1 mul, 2 add, 3 mul/add]
//        [blocks: 0 = every CU, else that many workgroups]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

// 64 independent-enough fp64 instructions per loop iteration (8 chains x 8, inline asm so that the compiler
// neither folds nor reorders them; constants in VGPRs): the loop's branch is ~3 % of the body
template <int OP>
__global__ void k_load(double* out, unsigned long long* stamps, int iters, const double* stream, size_t n_stream) {
    double a[8];
    for (int i = 0; i < 8; ++i) a[i] = 1.0 + 1e-9 * (threadIdx.x + i);
    double m = 1.0000001, c = 1e-12;
    asm volatile("" : "+v"(m), "+v"(c));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    double s = 0.0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
                else if (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                else if (OP == 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                else if (i & 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                else asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            }
        if (stream) s += stream[((size_t)blockIdx.x * blockDim.x + threadIdx.x + (size_t)it * 4096) % n_stream];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double acc = s;
    for (int i = 0; i < 8; ++i) acc += a[i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) {
        const size_t w = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / 64;
        stamps[2 * w] = t1 - t0;
        stamps[2 * w + 1] = r1 - r0;
    }
}

int main(int argc, char** argv) {
    const int wps = argc > 1 ? atoi(argv[1]) : 2, iters = argc > 2 ? atoi(argv[2]) : 200000, mode = argc > 3 ? atoi(argv[3]) : 0;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int nblk_arg = argc > 5 ? atoi(argv[5]) : 0;
    const int blocks = nblk_arg > 0 ? nblk_arg : prop.multiProcessorCount * wps, threads = 256;   // 4 waves per block: wps waves per SIMD
    double *d_out, *d_stream = nullptr;
    unsigned long long* d_st;
    const size_t n_waves = (size_t)blocks * threads / 64, n_stream = (size_t)1 << 27;
    hipMalloc(&d_out, (size_t)blocks * threads * sizeof(double));
    hipMalloc(&d_st, n_waves * 2 * sizeof(unsigned long long));
    if (mode == 1) { hipMalloc(&d_stream, n_stream * sizeof(double)); hipMemset(d_stream, 0, n_stream * sizeof(double)); }
    for (int rep = 0; rep < 3; ++rep) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        const int op = argc > 4 ? atoi(argv[4]) : 0;
        if (op == 0) hipLaunchKernelGGL(k_load<0>, dim3(blocks), dim3(threads), 0, 0, d_out, d_st, iters, d_stream, n_stream);
        else if (op == 1) hipLaunchKernelGGL(k_load<1>, dim3(blocks), dim3(threads), 0, 0, d_out, d_st, iters, d_stream, n_stream);
        else if (op == 2) hipLaunchKernelGGL(k_load<2>, dim3(blocks), dim3(threads), 0, 0, d_out, d_st, iters, d_stream, n_stream);
        else hipLaunchKernelGGL(k_load<3>, dim3(blocks), dim3(threads), 0, 0, d_out, d_st, iters, d_stream, n_stream);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> st(n_waves * 2);
        hipMemcpy(st.data(), d_st, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        std::vector<double> ghz(n_waves);
        for (size_t w = 0; w < n_waves; ++w) ghz[w] = (double)st[2 * w] / (double)st[2 * w + 1] * 0.1;
        std::sort(ghz.begin(), ghz.end());
        const double instr = (double)iters * 64;                                   // fp64 FMAs per wave
        std::vector<double> cyc(n_waves);
        for (size_t w = 0; w < n_waves; ++w) cyc[w] = (double)st[2 * w];
        std::sort(cyc.begin(), cyc.end());
        const double cyc_per = cyc.back() / instr;                        // the wave that took longest
        printf("waves/SIMD %d  mode %d  kernel %.3f ms  in-kernel clock median %.3f GHz (min %.3f max %.3f)  longest wave: cycles per instruction %.2f  -> chip rate %.1f TFLOP/s fp64\n",
               wps, mode, ms, ghz[n_waves / 2], ghz.front(), ghz.back(), cyc_per,
               (double)n_waves * instr * 128.0 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
