// store_pattern_probe.hip -- what the store phase of a forward row kernel costs as a function of HOW a wave's 16-byte
// lane stores fall on the 128-byte lines of T (round 4: the one-row-per-wave kernels' store phase measured twice
// the two-stage kernels' for the same bytes; this isolates the pattern from everything else).
// Every mode writes the same [walker][kx][y-group][c][r] volume once (ny = nx = N, kx <= N/2, 16-byte elements):
//   0  one row per wave, lanes = 64 consecutive kx (psfmc_rows3_path.h): 64 lines per instruction, 16 bytes each
//   1  two rows per wave, lanes t and t + 32 = the two rows of a group (psfmc_fused_path.h at nx = 1024):
//      32 lines per instruction, 32 bytes each (from lanes 32 apart)
//   2  one row per wave, ADJACENT lanes write adjacent 16-byte pieces (what a [kx][y][c] layout with a lane-pair
//      exchange would give): 32 lines per instruction, 32 bytes each (from neighbouring lanes)
//   3  four rows per wave, lanes t, t + 16, t + 32, t + 48 = the four rows of a group (nx = 512's pattern): 64 bytes
//   4  plain contiguous stores (the write sweep)
//   5  one row per wave, a lane's two elements ADJACENT ([kx][y][c]) but written by two instructions (16 bytes each):
//      does anything merge the two half-sector writes of consecutive instructions?
//   6  the column kernel's side of such a layout: a wave per (kx, c) column, 64 lanes = 64 consecutive y, elements 32
//      bytes apart (the other component's wave fills the gaps) -- against 7, today's 64-byte runs (4 rows of one c)
//   8  one row per wave, TODAY's layout, a lane pair shares a line: lane 2j writes (c = 0, r), lane 2j + 1 writes
//      (c = 1, r) of the SAME kx in one instruction (16 bytes each, 64 bytes apart): 32 lines per instruction
// Build: hipcc -O3 --offload-arch=gfx950 tools/store_pattern_probe.hip -o build/probe/store_pattern_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int MODE>
__global__ void __launch_bounds__(256) k_store(double2* __restrict__ T, int N, int n_w) {
    const int nxh = N / 2 + 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int w = blockIdx.y;
    const size_t kstride = (size_t)2 * N;                       // elements between kx columns (ny = N, groups of 4 rows)
    double2* base = T + (size_t)w * nxh * kstride;
    const double2 val = {1.0 + lane, 2.0};
    if (MODE == 4) {
        const size_t per_w = (size_t)nxh * kstride, n_thr = (size_t)gridDim.x * 256;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < per_w; i += n_thr) base[i] = val;
        return;
    }
    constexpr int ROWS = MODE == 1 ? 2 : MODE == 3 ? 4 : 1, LANES = 64 / ROWS;
    const int r0 = (blockIdx.x * 4 + wave) * ROWS;              // first row of this wave
    if (r0 >= N) return;
    const int f = lane / LANES, t = lane % LANES;
    const int y = r0 + f;
    const size_t row_el = (size_t)((y >> 2) * 8 + (y & 3));     // c = 0 element of row y inside a kx column
    if (MODE == 5) {
        const size_t row2 = (size_t)y * 2;
        for (int e = 0; e * 64 < nxh; ++e) {
            const int kx = t + 64 * e;
            if (kx < nxh) {
                double2* o = base + kx * kstride + row2;
                o[0] = val;
                o[1] = val;
            }
        }
        return;
    }
    if (MODE == 6 || MODE == 7) {
        // column waves: wave id -> (kx, c); lane -> y = lane + 64 a
        const int col = (blockIdx.x * 4 + wave);
        if (col >= 2 * nxh) return;
        const int kx = col >> 1, c = col & 1;
        double2* cb = base + kx * kstride;
        for (int a = 0; a * 64 < N; ++a) {
            const int yy = lane + 64 * a;
            const size_t el = MODE == 6 ? (size_t)yy * 2 + c : (size_t)((yy >> 2) * 8 + c * 4 + (yy & 3));
            cb[el] = val;
        }
        return;
    }
    if (MODE == 8) {
        for (int e = 0; e * 64 < nxh; ++e) {
            const int kx_a = (t & ~1) + 64 * e, kx_b = (t | 1) + 64 * e;
            if (kx_a < nxh) base[kx_a * kstride + row_el + 4 * (t & 1)] = val;
            if (kx_b < nxh) base[kx_b * kstride + row_el + 4 * (t & 1)] = val;
        }
        return;
    }
    if (MODE == 2) {
        // adjacent lanes write adjacent pieces: layout [kx][y][c]
        const size_t row2 = (size_t)y * 2;
        for (int e = 0; e * 64 < nxh; ++e) {
            const int kx_a = (t & ~1) + 64 * e, kx_b = (t | 1) + 64 * e;
            if (kx_a < nxh) base[kx_a * kstride + row2 + (t & 1)] = val;
            if (kx_b < nxh) base[kx_b * kstride + row2 + (t & 1)] = val;
        }
        return;
    }
    for (int e = 0; e * LANES < nxh; ++e) {
        const int kx = t + LANES * e;
        if (kx < nxh) {
            double2* o = base + kx * kstride + row_el;
            o[0] = val;
            o[4] = val;
        }
    }
}

template <int MODE> static void run(double2* T, int N, int n_w, const char* what) {
    constexpr int ROWS = MODE == 1 ? 2 : MODE == 3 ? 4 : 1;
    const dim3 grid(MODE == 4 ? 2048 / n_w + 1 : (MODE == 6 || MODE == 7) ? (2 * (N / 2 + 1) + 3) / 4 : (N / ROWS + 3) / 4, n_w);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_store<MODE>), grid, dim3(256), 0, 0, T, N, n_w);
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int reps = 20;
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_store<MODE>), grid, dim3(256), 0, 0, T, N, n_w);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    const double bytes = (double)n_w * (N / 2 + 1) * 2.0 * N * 16.0;
    printf("N %4d  %2d walkers  %-58s %7.1f us  %6.0f GB/s\n", N, n_w, what, ms * 1e3 / reps, bytes / (ms * 1e-3 / reps) / 1e9);
}

int main() {
    for (int cfg = 0; cfg < 2; ++cfg) {
        const int N = cfg ? 512 : 1024, n_w = cfg ? 24 : 6;
        const size_t bytes = (size_t)n_w * (N / 2 + 1) * 2 * N * 16;
        double2* T;
        CK(hipMalloc(&T, bytes));
        CK(hipMemset(T, 0, bytes));
        run<0>(T, N, n_w, "one row per wave: 16-byte pieces");
        run<1>(T, N, n_w, "two rows per wave: 32 bytes from lanes 32 apart");
        run<2>(T, N, n_w, "one row per wave: 32 bytes from neighbouring lanes");
        run<3>(T, N, n_w, "four rows per wave: 64 bytes from lanes 16 apart");
        run<4>(T, N, n_w, "contiguous");
        run<5>(T, N, n_w, "one row per wave: 2 x 16 bytes of one sector, two instructions");
        run<6>(T, N, n_w, "column waves: 16 bytes every 32 (layout [kx][y][c])");
        run<7>(T, N, n_w, "column waves: 64-byte runs (today's layout)");
        run<8>(T, N, n_w, "one row per wave: a lane pair writes c = 0 and c = 1 of one kx");
        CK(hipFree(T));
    }
    return 0;
}
