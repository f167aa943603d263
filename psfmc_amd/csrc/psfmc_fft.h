// psfmc_fft.h -- register/LDS complex FFT engine for gfx950 (fp64).
//
// One length-N transform (N = P*T, power of two) is computed by T adjacent lanes
// of a wave, each holding P points in registers:
//     v[a] = x[T*a + t]          on entry  (t = lane within the group, a < P)
//     v[e] = X[t + T*e]          on exit   (natural order, same striding)
// so that global loads/stores of consecutive lanes touch consecutive addresses.
// Decimation in frequency, two stages with ONE exchange through LDS:
//   stage 1  radix-P DFT over a in registers, then the twiddle W_N^(t*c)
//   exchange y[t][c] -> LDS rows of P+1 complex (the +1 keeps ds_write_b128 of
//            the T lanes on distinct banks; reads are contiguous across lanes);
//            wave-local, so there is no workgroup barrier anywhere in a transform
//   stage 2  P/T radix-T DFTs over the T lanes' values, in registers
//   X[c + P*d] = sum_b W_N^(b c) W_T^(b d) sum_a x[T a + b] W_P^(a c)
// The in-register DFTs are fully unrolled radix-2 recursions with compile-time
// twiddles (trivial factors 1, -i, (1-i)/sqrt2 special-cased).  Direction is a
// template parameter: SIGN = -1 forward (numpy's convention), +1 inverse
// (unnormalised).  A 64-lane wave holds 64/T transforms side by side.
#pragma once
#include <hip/hip_runtime.h>

namespace psfmc {

struct cd {
    double x, y;
};

__device__ __forceinline__ cd cadd(cd a, cd b) { return cd{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cd csub(cd a, cd b) { return cd{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ cd cmul(cd a, cd b) {
    return cd{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}
__device__ __forceinline__ cd cconj(cd a) { return cd{a.x, -a.y}; }

// 16-byte load of a value that will not be read again by this kernel's launch
// (streamed once): the non-temporal policy keeps it from displacing data that other
// kernels of the pipeline are about to re-read from the caches.
#ifndef PSFMC_NT_LOADS
#define PSFMC_NT_LOADS 0      /* measured: hurts k_rows_inv (its mirrored re-reads want the cache), neutral elsewhere */
#endif
typedef double psfmc_v2d __attribute__((ext_vector_type(2)));
__device__ __forceinline__ cd load_stream(const cd* p) {
#if PSFMC_NT_LOADS
    const psfmc_v2d v = __builtin_nontemporal_load(reinterpret_cast<const psfmc_v2d*>(p));
    return cd{v.x, v.y};
#else
    return *p;
#endif
}

template <int N> struct FftShape;
template <> struct FftShape<64>   { static constexpr int P = 8,  T = 8;  };
template <> struct FftShape<128>  { static constexpr int P = 16, T = 8;  };
template <> struct FftShape<256>  { static constexpr int P = 16, T = 16; };
template <> struct FftShape<512>  { static constexpr int P = 32, T = 16; };
template <> struct FftShape<1024> { static constexpr int P = 32, T = 32; };

// LDS doubles one transform needs for its exchange (one component at a time)
template <int N> constexpr int fft_lds_elems() { return FftShape<N>::T * (FftShape<N>::P + 1); }

// cos(2 pi k / 32), k = 0..8
__device__ constexpr double kCos32[9] = {
    1.0,
    0.98078528040323044912618223613423903697393373089333609500291,
    0.92387953251128675612818318939678828682241662586364248611509,
    0.83146961230254523707878837761790575673856081198797241619098,
    0.70710678118654752440084436210484903928483593768847403658834,
    0.55557023301960222474283081394853287437493719075480404592415,
    0.38268343236508977172845998403039886676134456248562704143380,
    0.19509032201612826784828486847702224092769161775195480775450,
    0.0};

// real / imaginary part of exp(SIGN * 2 pi i * k / R), compile time, R | 32
template <int R, int K> __device__ constexpr double tw_cos() {
    constexpr int k = ((K % R) + R) % R * (32 / R);          // in 32nds of a turn
    return k <= 8 ? kCos32[k] : k <= 16 ? -kCos32[16 - k] : k <= 24 ? -kCos32[k - 16] : kCos32[32 - k];
}
template <int R, int K> __device__ constexpr double tw_sin() {   // sin(2 pi K / R)
    return tw_cos<R, K - R / 4>();
}

// t = v * exp(SIGN 2 pi i K / R)
template <int R, int K, int SIGN> __device__ __forceinline__ cd tw_mul(cd v) {
    constexpr int k = ((K % R) + R) % R;
    if constexpr (k == 0) {
        return v;
    } else if constexpr (4 * k == R) {            // exp(SIGN i pi/2) = SIGN i
        return SIGN < 0 ? cd{v.y, -v.x} : cd{-v.y, v.x};
    } else if constexpr (2 * k == R) {
        return cd{-v.x, -v.y};
    } else if constexpr (4 * k == 3 * R) {
        return SIGN < 0 ? cd{-v.y, v.x} : cd{v.y, -v.x};
    } else {
        constexpr double c = tw_cos<R, k>();
        constexpr double s = SIGN * tw_sin<R, k>();
        return cd{v.x * c - v.y * s, v.x * s + v.y * c};
    }
}

// in-register DFT of R points, natural order in and out
template <int R, int SIGN> struct Dft {
    static __device__ __forceinline__ void run(cd (&v)[R]) {
        cd ev[R / 2], od[R / 2];
#pragma unroll
        for (int i = 0; i < R / 2; ++i) {
            ev[i] = v[2 * i];
            od[i] = v[2 * i + 1];
        }
        Dft<R / 2, SIGN>::run(ev);
        Dft<R / 2, SIGN>::run(od);
        combine<0>(v, ev, od);
    }
    template <int K>
    static __device__ __forceinline__ void combine(cd (&v)[R], const cd (&ev)[R / 2], const cd (&od)[R / 2]) {
        if constexpr (K < R / 2) {
            if constexpr (K == 0 || 4 * K == R) {          // twiddle 1 or -+i: two adds per output
                const cd t = tw_mul<R, K, SIGN>(od[K]);
                v[K] = cadd(ev[K], t);
                v[K + R / 2] = csub(ev[K], t);
            } else {
                // ev + od w in two FMAs per component, and ev - od w = 2 ev - (ev + od w) in
                // one: 6 instructions per butterfly where twiddle-multiply-then-add/sub is 8
                // (the (+-1 +- i)/sqrt2 twiddles included).  The second output inherits the
                // first one's rounding: the error stays eps (|ev| + |od|) like the plain form.
                constexpr double c = tw_cos<R, K>();
                constexpr double s = SIGN * tw_sin<R, K>();
                const cd e = ev[K], o = od[K];
                const cd a = cd{__builtin_fma(o.x, c, __builtin_fma(-o.y, s, e.x)),
                                __builtin_fma(o.x, s, __builtin_fma(o.y, c, e.y))};
                v[K] = a;
                v[K + R / 2] = cd{__builtin_fma(2.0, e.x, -a.x), __builtin_fma(2.0, e.y, -a.y)};
            }
            combine<K + 1>(v, ev, od);
        }
    }
};
template <int SIGN> struct Dft<1, SIGN> {
    static __device__ __forceinline__ void run(cd (&)[1]) {}
};
template <int SIGN> struct Dft<2, SIGN> {
    static __device__ __forceinline__ void run(cd (&v)[2]) {
        const cd a = v[0], b = v[1];
        v[0] = cadd(a, b);
        v[1] = csub(a, b);
    }
};

// LDS hand-off between lanes of ONE wave.  A wave's DS instructions execute in
// program order, so a ds_read issued after a ds_write of the same wave observes
// it; all that is needed is that the compiler keeps that order (the fences) and
// that the wave is converged here.  No s_barrier: waves never wait for each other.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Per-lane inter-stage twiddles W_N^(t*c), c < P, from the table tw[k] =
// exp(-2 pi i k/N).  Three homes (PSFMC_TW_MODE):
//   0  registers for the whole kernel (P <= 16 only): fastest per transform, but
//      4 P VGPRs that cap the waves a SIMD can hold
//   1  a per-wave LDS table twl[c][t] (lane-contiguous, conflict-free reads), filled
//      once per wave: a few extra ds_read_b128 per transform, 60 VGPRs back
//   2  re-read from the global table at every use (always for P = 32)
// Measured at 256^2 (bench.py, MI355X): mode 0 1.085 M evals/s at 2-3 waves/SIMD, mode 1
// 1.06 M at 4 waves/SIMD -- the row kernels are not occupancy-limited, so their default is 0.
// The column kernel (PSFMC_TW_MODE_COLS) uses mode 1: alone it is neutral (47.3 vs 46.6 us),
// but the 60 VGPRs pay for a second register set that pipelines its loads (43.3 us).
#ifndef PSFMC_TW_MODE
#define PSFMC_TW_MODE 0
#endif
#ifndef PSFMC_TW_MODE_COLS
#define PSFMC_TW_MODE_COLS 1     /* the column kernel: LDS table, the registers go to its load pipeline */
#endif
template <int N, int TWM = PSFMC_TW_MODE> constexpr int fft_tw_mode() { return FftShape<N>::P > 16 ? 2 : TWM; }
template <int N, int TWM = PSFMC_TW_MODE> constexpr bool fft_tw_in_regs() { return fft_tw_mode<N, TWM>() == 0; }
template <int N, int TWM = PSFMC_TW_MODE> struct TwRegs { static constexpr int value = (FftShape<N>::P > 16 ? 2 : TWM) == 0 ? FftShape<N>::P : 1; };
template <int N, int TWM = PSFMC_TW_MODE> constexpr int fft_tw_regs() { return TwRegs<N, TWM>::value; }
// LDS complex elements of the per-wave twiddle table (mode 1)
template <int N, int TWM = PSFMC_TW_MODE> constexpr int fft_tw_lds_elems() {
    return fft_tw_mode<N, TWM>() == 1 ? FftShape<N>::P * FftShape<N>::T : 0;
}

// `twl`: this wave's LDS table (mode 1) -- every lane of the wave must call.
template <int N, int TWM = PSFMC_TW_MODE>
__device__ __forceinline__ void load_twiddles(cd* w /* [TwRegs<N, TWM>::value] */, const cd* __restrict__ table, int t,
                                              cd* __restrict__ twl, int lane) {
    constexpr int P = FftShape<N>::P, T = FftShape<N>::T;
    if constexpr (fft_tw_mode<N, TWM>() == 0) {
#pragma unroll
        for (int c = 0; c < P; ++c) w[c] = table[t * c];
    } else {
        w[0] = cd{1.0, 0.0};
        if constexpr (fft_tw_mode<N, TWM>() == 1) {
#pragma unroll
            for (int i = lane; i < P * T; i += 64) twl[i] = table[(i / T) * (i % T)];   // twl[c][t] = W^(t c)
            wave_lds_sync();
        }
    }
}

// The cooperative transform.  `xbuf` = this transform's private LDS region of
// fft_lds_elems<N>() DOUBLES; the T lanes of a transform sit in one wave (T <= 32),
// `t` in [0,T).  `w` from load_twiddles (forward table).  Converged call only.
// The exchange goes through LDS one component at a time (real parts, then
// imaginary parts): half the LDS footprint per wave, which is what bounds how many
// waves a CU can hold, for the same number of LDS bytes moved.
template <int N, int SIGN, int TWM = PSFMC_TW_MODE>
__device__ __forceinline__ void fft_wave(cd (&v)[FftShape<N>::P], const cd* w /* [TwRegs<N, TWM>::value] */,
                                         const cd* __restrict__ table, int t, double* __restrict__ xbuf,
                                         const cd* __restrict__ twl) {
    constexpr int P = FftShape<N>::P, T = FftShape<N>::T;
    Dft<P, SIGN>::run(v);
#pragma unroll
    for (int c = 1; c < P; ++c) {
        cd wc;
        if constexpr (fft_tw_mode<N, TWM>() == 0) wc = w[c];
        else if constexpr (fft_tw_mode<N, TWM>() == 1) wc = twl[c * T + t];
        else wc = table[t * c];
        v[c] = cmul(v[c], SIGN < 0 ? wc : cconj(wc));
    }
    double* row = xbuf + t * (P + 1);
    cd z[P / T][T];
#pragma unroll
    for (int c = 0; c < P; ++c) row[c] = v[c].x;
    wave_lds_sync();
#pragma unroll
    for (int h = 0; h < P / T; ++h)
#pragma unroll
        for (int b = 0; b < T; ++b) z[h][b].x = xbuf[b * (P + 1) + t + T * h];
    wave_lds_sync();
#pragma unroll
    for (int c = 0; c < P; ++c) row[c] = v[c].y;
    wave_lds_sync();
#pragma unroll
    for (int h = 0; h < P / T; ++h)
#pragma unroll
        for (int b = 0; b < T; ++b) z[h][b].y = xbuf[b * (P + 1) + t + T * h];
    wave_lds_sync();
#pragma unroll
    for (int h = 0; h < P / T; ++h) {
        Dft<T, SIGN>::run(z[h]);
#pragma unroll
        for (int d = 0; d < T; ++d) v[h + (P / T) * d] = z[h][d];
    }
}

// ---------------------------------------------------------------------------
// Wave-wide three-stage transform for long columns (N = 512, 1024): all 64 lanes of
// a wave work on ONE transform, N = R1 * 8 * 8, lane t = 8 n2 + n3 holds
//     v[a] = x[64 a + t]   on entry,   v[e] = X[t + 64 e]   on exit   (a, e < R1)
// so every load / store instruction of the column kernel is one contiguous run and a
// lane needs only R1 = 8 or 16 complex registers (the two-stage engine needs 32 for
// these lengths, which left the column kernel at one wave per SIMD).
//   stage 1  radix-R1 over a (registers), twiddle W_N^(t k1)
//   exchange E1[k1][t]                                  (row stride 72 doubles)
//   stage 2  R1/8 radix-8 DFTs over n2 per lane, twiddle W_64^(n3 k2)
//   exchange E2[k2][n3][k1]                             (row stride R1+1 doubles)
//   stage 3  R1/8 radix-8 DFTs over n3 per lane
//   X[k1 + R1 k2 + 8 R1 k3]: lane t = (k1 + R1 k2) mod 64, e = (k1 + R1 k2)/64 + (R1/8) k3
// Exchanges go through LDS one component at a time, wave-local (no barriers).
// ---------------------------------------------------------------------------
template <int N> struct Fft3Shape { static constexpr int R1 = N / 64; };
template <int N> constexpr int fft3_lds_doubles() {
    constexpr int R1 = Fft3Shape<N>::R1;
    return (R1 * 72 > 64 * (R1 + 1)) ? R1 * 72 : 64 * (R1 + 1);
}

// per-lane twiddles: w1[k1] = W_N^(t k1) (k1 < R1), w2[k2] = W_N^(R1 n3 k2) (k2 < 8).
// For R1 = 16 the stage-1 set would cost 64 VGPRs: it comes from a table the workgroup
// shares in LDS (fft_wave3's w1_lds) or is re-read from the global table at each use.
template <int N> constexpr int fft3_w1_regs() { return Fft3Shape<N>::R1 <= 8 ? Fft3Shape<N>::R1 : 1; }
template <int N>
__device__ __forceinline__ void load_twiddles3(cd (&w1)[fft3_w1_regs<N>()], cd (&w2)[8],
                                               const cd* __restrict__ table, int t) {
    constexpr int R1 = Fft3Shape<N>::R1;
    if constexpr (R1 <= 8) {
#pragma unroll
        for (int k = 0; k < R1; ++k) w1[k] = table[t * k];
    } else {
        w1[0] = cd{1.0, 0.0};
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) w2[k] = table[R1 * (t & 7) * k];
}

// `w1_lds` (R1 = 16): the workgroup's shared stage-1 table [k1][t] = W_N^(t k1) in LDS,
// or nullptr to re-read the global table.
template <int N, int SIGN>
__device__ __forceinline__ void fft_wave3(cd (&v)[Fft3Shape<N>::R1], const cd (&w1)[fft3_w1_regs<N>()],
                                          const cd (&w2)[8], const cd* __restrict__ table, int t,
                                          double* __restrict__ lds, const cd* __restrict__ w1_lds = nullptr) {
    constexpr int R1 = Fft3Shape<N>::R1, NB = R1 / 8, S1 = 72, S2 = R1 + 1;
    const int n3 = t & 7, g = t >> 3;
    // stage 1
    Dft<R1, SIGN>::run(v);
#pragma unroll
    for (int k = 1; k < R1; ++k) {
        cd wk;
        if constexpr (R1 <= 8) wk = w1[k];
        else wk = w1_lds ? w1_lds[k * 64 + t] : table[t * k];
        v[k] = cmul(v[k], SIGN < 0 ? wk : cconj(wk));
    }
    // exchange 1 + stage 2
    cd z[NB][8];
#pragma unroll
    for (int k = 0; k < R1; ++k) lds[k * S1 + t] = v[k].x;
    wave_lds_sync();
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int n2 = 0; n2 < 8; ++n2) z[i][n2].x = lds[(g + 8 * i) * S1 + n2 * 8 + n3];
    wave_lds_sync();
#pragma unroll
    for (int k = 0; k < R1; ++k) lds[k * S1 + t] = v[k].y;
    wave_lds_sync();
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int n2 = 0; n2 < 8; ++n2) z[i][n2].y = lds[(g + 8 * i) * S1 + n2 * 8 + n3];
    wave_lds_sync();
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        Dft<8, SIGN>::run(z[i]);
#pragma unroll
        for (int k2 = 1; k2 < 8; ++k2) z[i][k2] = cmul(z[i][k2], SIGN < 0 ? w2[k2] : cconj(w2[k2]));
    }
    // exchange 2 + stage 3: E2[(k2*8 + n3)][k1], k1 = g + 8 i
    cd y[NB][8];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2) lds[(k2 * 8 + n3) * S2 + g + 8 * i] = z[i][k2].x;
    wave_lds_sync();
#pragma unroll
    for (int q = 0; q < NB; ++q) {
        const int c = t + 64 * q, k1 = c % R1, k2 = c / R1;
#pragma unroll
        for (int m = 0; m < 8; ++m) y[q][m].x = lds[(k2 * 8 + m) * S2 + k1];
    }
    wave_lds_sync();
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2) lds[(k2 * 8 + n3) * S2 + g + 8 * i] = z[i][k2].y;
    wave_lds_sync();
#pragma unroll
    for (int q = 0; q < NB; ++q) {
        const int c = t + 64 * q, k1 = c % R1, k2 = c / R1;
#pragma unroll
        for (int m = 0; m < 8; ++m) y[q][m].y = lds[(k2 * 8 + m) * S2 + k1];
    }
    wave_lds_sync();
#pragma unroll
    for (int q = 0; q < NB; ++q) {
        Dft<8, SIGN>::run(y[q]);
#pragma unroll
        for (int k3 = 0; k3 < 8; ++k3) v[q + NB * k3] = y[q][k3];
    }
}

}  // namespace psfmc
