// Issue cost of the fp64 VALU instructions the rasteriser and the transforms use, gfx950:
// one wave on one SIMD, 8 independent chains, cycles per instruction from s_memtime.
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_rates tools/valu_rates.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define ITER 20000

#define KERNEL(NAME, BODY)                                                                      \
    __global__ void NAME(double* out, long long* cyc, double b, double c, int ib) {             \
        double a[8];                                                                            \
        int n[8];                                                                               \
        for (int i = 0; i < 8; ++i) { a[i] = b + threadIdx.x * 1e-3 + i; n[i] = ib + i; }       \
        long long t0 = __builtin_readcyclecounter();                                            \
        for (int it = 0; it < ITER; ++it) {                                                     \
            REP8(BODY) REP8(BODY) REP8(BODY) REP8(BODY)                                         \
        }                                                                                       \
        long long t1 = __builtin_readcyclecounter();                                            \
        double s = 0; for (int i = 0; i < 8; ++i) s += a[i] + n[i];                             \
        out[threadIdx.x + blockIdx.x * blockDim.x] = s;                                         \
        if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                        \
    }

#define B_FMA(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define B_MUL(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define B_ADD(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define B_RCP(i) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[i]));
#define B_RSQ(i) asm volatile("v_rsq_f64 %0, %0" : "+v"(a[i]));
#define B_SQRT(i) asm volatile("v_sqrt_f64 %0, %0" : "+v"(a[i]));
#define B_LDEXP(i) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(a[i]) : "v"(n[i]));
#define B_FRMANT(i) asm volatile("v_frexp_mant_f64 %0, %0" : "+v"(a[i]));
#define B_FREXP(i) asm volatile("v_frexp_exp_i32_f64 %0, %1" : "=v"(n[i]) : "v"(a[i]));
#define B_RNDNE(i) asm volatile("v_rndne_f64 %0, %0" : "+v"(a[i]));
#define B_CVTI(i) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(n[i]) : "v"(a[i]));
#define B_CVTD(i) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a[i]) : "v"(n[i]));
#define B_MAX(i) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define B_CMP(i) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
#define B_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(n[i]) : "v"(ib) : "vcc");
#define B_ADDU(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(n[i]) : "v"(ib));
#define B_FMAF32(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(n[i]) : "v"(ib));
#define B_PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define B_EXPF32(i) asm volatile("v_exp_f32 %0, %0" : "+v"(n[i]));
#define B_DIVFIX(i) asm volatile("v_div_fixup_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define B_MOV64(i) asm volatile("v_mov_b32 %0, %1" : "=v"(n[i]) : "v"(ib));
#define B_DSREAD(i) asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(a[i]) : "v"(n[i] & 0xff8));

KERNEL(k_fma, B_FMA) KERNEL(k_mul, B_MUL) KERNEL(k_add, B_ADD) KERNEL(k_rcp, B_RCP) KERNEL(k_rsq, B_RSQ)
KERNEL(k_sqrt, B_SQRT) KERNEL(k_ldexp, B_LDEXP) KERNEL(k_frmant, B_FRMANT) KERNEL(k_frexp, B_FREXP)
KERNEL(k_rndne, B_RNDNE) KERNEL(k_cvti, B_CVTI) KERNEL(k_cvtd, B_CVTD) KERNEL(k_max, B_MAX)
KERNEL(k_cmp, B_CMP) KERNEL(k_cndmask, B_CNDMASK) KERNEL(k_addu, B_ADDU) KERNEL(k_fmaf32, B_FMAF32)
KERNEL(k_pkfma, B_PKFMA) KERNEL(k_expf32, B_EXPF32)

typedef void (*kern_t)(double*, long long*, double, double, int);
struct Entry { const char* name; kern_t k; };

int main() {
    Entry es[] = {{"v_fma_f64", k_fma}, {"v_mul_f64", k_mul}, {"v_add_f64", k_add}, {"v_rcp_f64", k_rcp},
                  {"v_rsq_f64", k_rsq}, {"v_sqrt_f64", k_sqrt}, {"v_ldexp_f64", k_ldexp},
                  {"v_frexp_mant_f64", k_frmant}, {"v_frexp_exp_i32_f64", k_frexp}, {"v_rndne_f64", k_rndne},
                  {"v_cvt_i32_f64", k_cvti}, {"v_cvt_f64_i32", k_cvtd}, {"v_max_f64", k_max},
                  {"v_cmp_lt_f64", k_cmp}, {"v_cndmask_b32", k_cndmask}, {"v_add_u32", k_addu},
                  {"v_fma_f32", k_fmaf32}, {"v_pk_fma_f32", k_pkfma}, {"v_exp_f32", k_expf32}};
    double* out; long long* cyc;
    hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 4096);
    // s_memtime counts at a constant 100 MHz on gfx9; convert with the wall clock of a known kernel
    for (int waves = 1; waves <= 4; waves *= 2) {
        printf("--- %d wave(s) per SIMD (block of %d threads on one CU)\n", waves, waves * 256);
        for (auto& e : es) {
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            e.k<<<1, waves * 256>>>(out, cyc, 1.0000001, 1e-9, 3);
            hipDeviceSynchronize();
            hipEventRecord(a);
            e.k<<<1, waves * 256>>>(out, cyc, 1.0000001, 1e-9, 3);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
            double n_inst = 32.0 * ITER * waves;           // per SIMD
            printf("%-22s  %8.3f ms  -> %.2f ns per wave-instruction per SIMD  (counter %lld)\n", e.name, ms,
                   (ms * 1e6 - 8000) / n_inst, h);
        }
    }
    return 0;
}
