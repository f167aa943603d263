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

// VAR: how a consumer makes sure it sees the team's stores -- 0: plain loads after `buffer_inv sc0`,
// 1: plain loads after `buffer_inv sc1`, 2: loads with the sc1 (agent-scope) bit, no invalidate;
// SLEEP: s_sleep argument between polls; LOCAL: poll with a workgroup-scope RMW (performed at the L2)
// all lanes of the wave call; returns false when the spin cap was hit
template <int VAR, int SLEEP, bool LOCAL>
__device__ __forceinline__ bool team_barrier(int* bar, int target, Ctl* ctl, long long& polls) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    bool ok = true;
    if (threadIdx.x == 0) {
        if (LOCAL) __hip_atomic_fetch_add(bar, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else __hip_atomic_fetch_add(bar, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        long long n = 0;
        while ((LOCAL ? __hip_atomic_fetch_add(bar, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
                      : __hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < target) {
            __builtin_amdgcn_s_sleep(SLEEP);
            if (++n > kSpinCap || __hip_atomic_load(&ctl->abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(&ctl->abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = false;
                break;
            }
        }
        polls += n;
    }
    ok = __shfl((int)ok, 0, 64) != 0;
    if (VAR == 0) asm volatile("buffer_inv sc0" ::: "memory");
    if (VAR == 1) asm volatile("buffer_inv sc1" ::: "memory");
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

template <int VAR>
__device__ __forceinline__ double2 ld(const double2* p) {
    if (VAR != 2) return *p;
    double2 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

template <bool FILL, int VAR, int SLEEP, bool LOCAL>
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
        if (!team_barrier<VAR, SLEEP, LOCAL>(bar, kTeam * (++gen), ctl, polls)) return;
        // phase 2: wave `rank` read-modify-writes a "column": 16 elements of every wave's chunk
        double2 v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int src_wave = k * 4 + (lane >> 4);                       // 0..63
            const int i = src_wave * 1024 + rank * 16 + (lane & 15);
            v[k] = ld<VAR>(T + i);
        }
        if (FILL) acc += filler<840>(acc);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int src_wave = k * 4 + (lane >> 4);
            const int i = src_wave * 1024 + rank * 16 + (lane & 15);
            bad += (v[k].x != (double)w) | (v[k].y != (double)i);
            T[i] = double2{v[k].x + 1.0, v[k].y};
        }
        if (!team_barrier<VAR, SLEEP, LOCAL>(bar, kTeam * (++gen), ctl, polls)) return;
        // phase 3: wave `rank` reads its contiguous chunk back
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int i = rank * 1024 + k * 64 + lane;
            const double2 u = ld<VAR>(T + i);
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

template <bool FILL, int VAR, int SLEEP, bool LOCAL>
static void run(const char* label, double2* slots, Ctl* ctl, double* sink, int n_walkers) {
    for (int tpx : {1, 2, 4}) {
        float best = 1e9; Ctl h;
        for (int rep = 0; rep < 2; ++rep) {
            hipMemset(ctl, 0, sizeof(Ctl));
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipDeviceSynchronize();
            hipEventRecord(a);
            k_team<FILL, VAR, SLEEP, LOCAL><<<8 * tpx * kTeam, 64, 20000>>>(slots, ctl, n_walkers, tpx, sink);
            hipEventRecord(b);
            if (hipEventSynchronize(b) != hipSuccess) { printf("launch failed\n"); exit(1); }
            float ms; hipEventElapsedTime(&ms, a, b);
            if (ms < best) best = ms;
            hipMemcpy(&h, ctl, sizeof h, hipMemcpyDeviceToHost);
        }
        int teams = 0; for (int x = 0; x < 8; ++x) teams += h.teams_seen[x];
        printf("%-46s filler %d teams/XCD %d  %8.3f ms = %8.0f walkers/s | teams %d/%d abort %d mismatches %lld polls/barrier %.1f\n",
               label, (int)FILL, tpx, best, n_walkers / (best * 1e-3), teams, 8 * tpx, h.abort_flag, h.mismatches,
               (double)h.spin_polls / ((double)teams * kTeam * 2.0 * n_walkers / (8.0 * tpx)));
    }
}

int main(int argc, char** argv) {
    double2* slots; Ctl* ctl; double* sink;
    hipMalloc(&slots, (size_t)8 * kMaxTeams * 65536 * sizeof(double2));
    hipMalloc(&ctl, sizeof(Ctl)); hipMalloc(&sink, 64);
    const int n = 2048;
    run<false, 0, 2, false>("plain loads + buffer_inv sc0, agent polls", slots, ctl, sink, n);
    run<false, 1, 2, false>("plain loads + buffer_inv sc1, agent polls", slots, ctl, sink, n);
    run<false, 2, 2, false>("sc1 loads, agent polls", slots, ctl, sink, n);
    run<false, 1, 16, false>("buffer_inv sc1, agent polls, sleep 16", slots, ctl, sink, n);
    run<false, 1, 2, true>("buffer_inv sc1, L2-local RMW polls", slots, ctl, sink, n);
    run<false, 1, 16, true>("buffer_inv sc1, L2-local RMW polls, sleep 16", slots, ctl, sink, n);
    run<false, 2, 16, true>("sc1 loads, L2-local RMW polls, sleep 16", slots, ctl, sink, n);
    run<true, 1, 16, true>("buffer_inv sc1, L2-local RMW polls, sleep 16", slots, ctl, sink, n);
    run<true, 2, 16, true>("sc1 loads, L2-local RMW polls, sleep 16", slots, ctl, sink, n);
    return 0;
}
