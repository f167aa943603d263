// Feasibility probe for an XCD-resident pipeline: teams of 64 single-wave workgroups on one XCD
// pass a 1 MiB "T slot" through three phases (write rows / read-modify-write columns / read rows)
// with flag barriers between them, the data staying in that XCD's L2.  Reports walkers/s, the
// barrier cost, and checks every value read (stale lines would show as mismatches).
// build: hipcc --offload-arch=gfx950 -O2 -o _ab/xcd_team_probe tools/xcd_team_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

constexpr int kTeam = 64;                 // waves per team
constexpr int kMaxTeams = 8;              // per XCD
constexpr long long kSpinCap = 1 << 20;

struct Ctl {
    int arrive[8];
    int bar[8 * kMaxTeams];
    int next_walker;
    int abort_flag;
    int teams_seen[8];
    long long mismatches;
    long long spin_polls;
};

__device__ __forceinline__ int xcc_id() {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    return (int)(id & 0xf);
}

// all lanes of the wave call; returns false when the spin cap was hit
__device__ __forceinline__ bool team_barrier(int* bar, int target, Ctl* ctl, long long& polls) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    bool ok = true;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(bar, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        long long n = 0;
        while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (++n > kSpinCap || __hip_atomic_load(&ctl->abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(&ctl->abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = false;
                break;
            }
        }
        polls += n;
    }
    ok = __shfl((int)ok, 0, 64) != 0;
    asm volatile("buffer_inv sc0" ::: "memory");           // drop this CU's L1 lines
    return ok;
}

template <int N>
__device__ __forceinline__ double filler(double x) {
    double a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3;
#pragma unroll 8
    for (int i = 0; i < N / 4; ++i) {
        a0 = __builtin_fma(a0, 1.0000001, 1e-9); a1 = __builtin_fma(a1, 1.0000001, 1e-9);
        a2 = __builtin_fma(a2, 1.0000001, 1e-9); a3 = __builtin_fma(a3, 1.0000001, 1e-9);
    }
    return a0 + a1 + a2 + a3;
}

template <bool FILL>
__global__ __launch_bounds__(64) void k_team(double2* __restrict__ slots, Ctl* ctl, int n_walkers, int teams_per_xcd,
                                             double* sink) {
    extern __shared__ double pad[];
    const int xcd = xcc_id();
    if (xcd >= 8) return;
    int slot = 0;
    if (threadIdx.x == 0) slot = __hip_atomic_fetch_add(&ctl->arrive[xcd], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    slot = __shfl(slot, 0, 64);
    const int team = slot / kTeam, rank = slot % kTeam;
    if (team >= teams_per_xcd) return;
    const int lane = threadIdx.x;
    int* bar = &ctl->bar[xcd * kMaxTeams + team];
    double2* T = slots + (size_t)(xcd * kMaxTeams + team) * 65536;          // 1 MiB = 65536 double2
    const int n_teams = 8 * teams_per_xcd;
    const int tg = team * 8 + xcd;                                          // global team id
    long long polls = 0, bad = 0;
    double acc = 0;
    int gen = 0;
    if (rank == 0 && lane == 0) atomicAdd(&ctl->teams_seen[xcd], 1);
    for (int w = tg; w < n_walkers; w += n_teams) {
        // phase 1: wave `rank` writes its contiguous 16 KiB (1024 double2): 16 per lane
        if (FILL) acc += filler<1700>(acc + lane);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int i = rank * 1024 + k * 64 + lane;
            T[i] = double2{(double)w, (double)i};
        }
        if (!team_barrier(bar, kTeam * (++gen), ctl, polls)) return;
        // phase 2: wave `rank` read-modify-writes a "column": 16 elements of every wave's chunk
        double2 v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int src_wave = k * 4 + (lane >> 4);                       // 0..63
            const int i = src_wave * 1024 + rank * 16 + (lane & 15);
            v[k] = T[i];
        }
        if (FILL) acc += filler<840>(acc);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int src_wave = k * 4 + (lane >> 4);
            const int i = src_wave * 1024 + rank * 16 + (lane & 15);
            bad += (v[k].x != (double)w) | (v[k].y != (double)i);
            T[i] = double2{v[k].x + 1.0, v[k].y};
        }
        if (!team_barrier(bar, kTeam * (++gen), ctl, polls)) return;
        // phase 3: wave `rank` reads its contiguous chunk back
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int i = rank * 1024 + k * 64 + lane;
            const double2 u = T[i];
            bad += (u.x != (double)w + 1.0) | (u.y != (double)i);
            acc += u.x;
        }
        if (FILL) acc += filler<770>(acc);
        // no barrier here: the next walker's phase 1 writes only this wave's own chunk
    }
    for (int o = 32; o > 0; o >>= 1) bad += __shfl_down(bad, o, 64);
    if (lane == 0) {
        if (bad) atomicAdd((unsigned long long*)&ctl->mismatches, (unsigned long long)bad);
        atomicAdd((unsigned long long*)&ctl->spin_polls, (unsigned long long)polls);
    }
    if (acc == 1.2345) sink[0] = acc;
}

int main(int argc, char** argv) {
    double2* slots; Ctl* ctl; double* sink;
    hipMalloc(&slots, (size_t)8 * kMaxTeams * 65536 * sizeof(double2));
    hipMalloc(&ctl, sizeof(Ctl)); hipMalloc(&sink, 64);
    const int n_walkers = 4096;
    for (int fill = 0; fill < 2; ++fill)
        for (int tpx : {1, 2, 3, 4, 6, 8}) {
            for (int lds_kb : {20, 13}) {
                if (lds_kb == 13 && tpx < 6) continue;
                float best = 1e9; Ctl h;
                for (int rep = 0; rep < 3; ++rep) {
                    hipMemset(ctl, 0, sizeof(Ctl));
                    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
                    hipDeviceSynchronize();
                    hipEventRecord(a);
                    const int grid = 8 * tpx * kTeam;
                    if (fill) k_team<true><<<grid, 64, lds_kb * 1000>>>(slots, ctl, n_walkers, tpx, sink);
                    else k_team<false><<<grid, 64, lds_kb * 1000>>>(slots, ctl, n_walkers, tpx, sink);
                    hipEventRecord(b);
                    if (hipEventSynchronize(b) != hipSuccess) { printf("launch failed\n"); return 1; }
                    float ms; hipEventElapsedTime(&ms, a, b);
                    if (ms < best) best = ms;
                    hipMemcpy(&h, ctl, sizeof h, hipMemcpyDeviceToHost);
                }
                int teams = 0; for (int x = 0; x < 8; ++x) teams += h.teams_seen[x];
                printf("filler %d  teams/XCD %d (LDS %2d KB/wave)  %8.3f ms for %d walkers = %8.0f walkers/s  | teams formed %d of %d, abort %d, mismatches %lld, polls/wave/barrier %.1f\n",
                       fill, tpx, lds_kb, best, n_walkers, n_walkers / (best * 1e-3), teams, 8 * tpx, h.abort_flag, h.mismatches,
                       (double)h.spin_polls / ((double)teams * kTeam * 2.0 * n_walkers / (8.0 * tpx)));
            }
        }
    return 0;
}
