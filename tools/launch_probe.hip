// launch_probe.hip -- what a dependent kernel boundary costs on this chip, against a grid-wide barrier inside
// one kernel: the floor under the small-ensemble half-step (five dependent launches).
// Build: hipcc -O3 --offload-arch=gfx950 tools/launch_probe.hip -o /tmp/launch_probe
// usage: launch_probe [blocks=352] [threads=64]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

__global__ void k_tiny(double* p) {
    if (threadIdx.x == 0) p[blockIdx.x] += 1.0;       // a dependent read-modify-write: the boundary must order it
}

// `rounds` grid-wide barriers inside one launch: every workgroup bumps a device-scope counter and waits for
// all of them (the grid is sized to be co-resident, so every wave reaches the exit)
__global__ void k_barriers(double* p, unsigned* counter, int rounds) {
    const unsigned n = gridDim.x;
    for (int r = 0; r < rounds; ++r) {
        if (threadIdx.x == 0) p[blockIdx.x] += 1.0;
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            atomicAdd(counter, 1u);
            const unsigned want = n * (unsigned)(r + 1);
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(1);
            __threadfence();
        }
        __syncthreads();
    }
}

static double now() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 352, threads = argc > 2 ? atoi(argv[2]) : 64;
    double* d;
    unsigned* cnt;
    hipMalloc(&d, blocks * sizeof(double));
    hipMemset(d, 0, blocks * sizeof(double));
    hipMalloc(&cnt, sizeof(unsigned));
    hipStream_t st;
    hipStreamCreate(&st);
    const int chain = 5, reps = 400;
    // (a) plain stream launches
    for (int pass = 0; pass < 2; ++pass) {
        hipStreamSynchronize(st);
        const double t0 = now();
        for (int r = 0; r < reps; ++r)
            for (int k = 0; k < chain; ++k) hipLaunchKernelGGL(k_tiny, dim3(blocks), dim3(threads), 0, st, d);
        hipStreamSynchronize(st);
        if (pass) printf("stream launches : %.2f us per dependent kernel (%d blocks x %d threads)\n",
                         (now() - t0) / (reps * chain) * 1e6, blocks, threads);
    }
    // (b) the same chain in a graph
    hipGraph_t g;
    hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int k = 0; k < chain; ++k) hipLaunchKernelGGL(k_tiny, dim3(blocks), dim3(threads), 0, st, d);
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int pass = 0; pass < 2; ++pass) {
        hipStreamSynchronize(st);
        const double t0 = now();
        for (int r = 0; r < reps; ++r) hipGraphLaunch(ge, st);
        hipStreamSynchronize(st);
        if (pass) printf("graph launches  : %.2f us per dependent kernel\n", (now() - t0) / (reps * chain) * 1e6);
    }
    // (c) one launch, grid-wide barriers between the stages
    for (int pass = 0; pass < 2; ++pass) {
        const int rounds = 200;
        hipMemsetAsync(cnt, 0, sizeof(unsigned), st);
        hipStreamSynchronize(st);
        const double t0 = now();
        hipLaunchKernelGGL(k_barriers, dim3(blocks), dim3(threads), 0, st, d, cnt, rounds);
        hipStreamSynchronize(st);
        if (pass) printf("grid barriers   : %.2f us per barrier (one launch, %d rounds)\n", (now() - t0) / rounds * 1e6, rounds);
    }
    // (d) event-timed single tiny kernel
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, st);
    for (int r = 0; r < 100; ++r) hipLaunchKernelGGL(k_tiny, dim3(blocks), dim3(threads), 0, st, d);
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("event-timed     : %.2f us per kernel\n", ms * 10.0);
    return 0;
}
