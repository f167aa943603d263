// psfmc_fft.h -- register/LDS complex FFT engine for gfx950 (fp64).
//
// One length-N transform (N = P*T; N a product of 2s, 3s and 5s) is computed by T adjacent
// lanes of a wave (T <= 32, a wave holds 64/T transforms; with T not a power of two the last
// 64 - T*(64/T) lanes idle), each holding P points in registers:
//     v[a] = x[T*a + t]              on entry  (t = lane within the group, a < P)
//     v[e] = X[fft_k_of(t, e)]       on exit:  e = h + H*d, H = ceil(P/T),
//                                    k = (t + T*h) + P*d, valid iff t + T*h < P
// When T divides P (every power-of-two shape) that is k = t + T*e: natural order with the same
// striding.  Either way consecutive lanes touch consecutive addresses in global loads/stores.
// Decimation in frequency, two stages with ONE exchange through LDS:
//   stage 1  radix-P DFT over a in registers, then the twiddle W_N^(t*c)
//   exchange y[t][c] -> LDS rows of P|1 doubles (an odd stride keeps the writes of
//            the T lanes on distinct banks; reads are contiguous across lanes);
//            wave-local, so there is no workgroup barrier anywhere in a transform
//   stage 2  P/T radix-T DFTs over the T lanes' values, in registers
//   X[c + P*d] = sum_b W_N^(b c) W_T^(b d) sum_a x[T a + b] W_P^(a c)
// The in-register DFTs are fully unrolled decimation-in-time recursions with compile-time
// twiddles: radix 5 and 3 butterflies for those factors, radix 2 for the rest (a butterfly with a
// non-trivial twiddle is 6 FMAs; twiddles 1 and -+i cost only the additions).  Direction is a
// template parameter: SIGN = -1 forward (numpy's convention), +1 inverse
// (unnormalised).  A 64-lane wave holds 64/T transforms side by side.
#pragma once
#include <hip/hip_runtime.h>

#include "psfmc_trig_table.h"
#include "psfmc_device.h"

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
// Optional single-precision STORAGE of the intermediate half-spectra (arithmetic stays fp64):
// a T element as two floats.  Converts implicitly on store, explicitly on load.
struct cf {
    float x, y;
    cf() = default;
    __device__ __forceinline__ cf(cd v) : x((float)v.x), y((float)v.y) {}
};
__device__ __forceinline__ cd load_stream(const cf* p) {
    const cf v = *p;
    return cd{(double)v.x, (double)v.y};
}

typedef double psfmc_v2d __attribute__((ext_vector_type(2)));
__device__ __forceinline__ cd load_stream(const cd* p) {
#if PSFMC_NT_LOADS
    const psfmc_v2d v = __builtin_nontemporal_load(reinterpret_cast<const psfmc_v2d*>(p));
    return cd{v.x, v.y};
#else
    return *p;
#endif
}

// LDS row stride (doubles) of a shape's exchange: odd, so that the T lanes writing one register
// hit distinct banks, and with T (stride - 1) a multiple of 32 where that costs at most half as
// much LDS again, so that the transforms sharing a wave (region = T * stride doubles apart) do
// not collide on the reads either (k_cols<300>: two thirds of its LDS cycles were conflicts).
constexpr int fft_lds_stride(int P, int T) {
    for (int ls = P | 1; ls <= P + P / 2 + 8; ls += 2)
        if ((T * (ls - 1)) % 32 == 0) return ls;
    return P | 1;
}

template <int N> struct FftShape;
#define PSFMC_FFT_SHAPE(N_, P_, T_)                                                                 \
    template <> struct FftShape<N_> {                                                               \
        static constexpr int P = P_, T = T_;                                                        \
        static_assert(P_ * T_ == N_ && T_ <= 32 && P_ <= 32, "shape");                              \
        static constexpr int H = (P_ + T_ - 1) / T_;        /* stage-2 transforms per lane */       \
        static constexpr int R = H * T_;                    /* registers (complex) per lane, >= P */ \
        static constexpr int TPW = 64 / T_;                 /* transforms per wave */               \
        static constexpr int LS = fft_lds_stride(P_, T_);   /* LDS row stride (doubles) */          \
        static constexpr bool kExact = (P_ % T_ == 0);      /* k = t + T e on exit */               \
        static constexpr bool kFull = (TPW * T_ == 64);     /* every lane of a wave works */        \
        static constexpr bool kPlain = kExact && kFull;     /* the power-of-two shapes */           \
    };
PSFMC_FFT_SHAPE(64, 8, 8)
PSFMC_FFT_SHAPE(128, 16, 8)
PSFMC_FFT_SHAPE(256, 16, 16)
PSFMC_FFT_SHAPE(512, 32, 16)
PSFMC_FFT_SHAPE(1024, 32, 32)
// Round 4, a survey of the shapes that leave many lanes idle (tools/side_costs.py with the alternative shape against
// this table, same box; profiles/r4_shape_survey*_{base,alt}.jsonl): where the COLUMNS of a side run on the three-stage
// engine anyway (fft3g_pick) the two-stage shape only serves the row kernels, and more lanes beat fewer registers --
// 250: 10 x 25 (50 lanes) -> 25 x 10 (60): +10 % whole step; 264: 12 x 22 (44) -> 22 x 12 (60): +19 %; 286: 13 x 22 -> 22 x 13:
// +19 %; 312: 13 x 24 -> 24 x 13: +15 %; 330: 15 x 22 -> 22 x 15: +9 %; 350: 14 x 25 -> 25 x 14: +9 %; 352: 16 x 22 -> 22 x 16: +17 %;
// 416: 16 x 26 -> 26 x 16: +15 %; 288 (once its columns had moved to the three-stage engine, late in round 4): 16 x 18 (54
// lanes) -> 24 x 12 (60): +4.6 %, 18 x 16 (64 lanes, 32 registers): +0 % (profiles/r4_shape_288.txt).  Measured and left alone: 384, 390, 448, 480, 504, 576, 600, 672 (-8 ... +3 %), every small
// side tried (110 ... 288: -1 ... -44 %: their two-stage COLUMN kernel pays for the registers) and the R = 40 shapes of 440,
// 500, 520, 560 (-13 ... -14 %).
// sides with factors 3 and 5 (any even side of this list runs on the fused kernels).  The
// shapes keep R = max(P, ceil(P/T) T) small -- registers, hence waves per SIMD, are what the
// memory-bound column kernel lives on -- at the price of a few idle lanes (T = 10, 15, 20, 30
// use 60 of a wave's 64 lanes, T = 18 54, T = 24 48, T = 25 50) or idle stage-2 slots (P < T).
PSFMC_FFT_SHAPE(96, 12, 8)
PSFMC_FFT_SHAPE(100, 10, 10)
PSFMC_FFT_SHAPE(120, 15, 8)
PSFMC_FFT_SHAPE(144, 12, 12)
PSFMC_FFT_SHAPE(150, 10, 15)
PSFMC_FFT_SHAPE(160, 10, 16)
PSFMC_FFT_SHAPE(180, 12, 15)
PSFMC_FFT_SHAPE(192, 12, 16)
PSFMC_FFT_SHAPE(200, 20, 10)
PSFMC_FFT_SHAPE(240, 15, 16)
PSFMC_FFT_SHAPE(250, 25, 10)
PSFMC_FFT_SHAPE(288, 24, 12)
PSFMC_FFT_SHAPE(300, 15, 20)
PSFMC_FFT_SHAPE(320, 16, 20)
PSFMC_FFT_SHAPE(360, 18, 20)
PSFMC_FFT_SHAPE(384, 16, 24)
PSFMC_FFT_SHAPE(400, 20, 20)
PSFMC_FFT_SHAPE(480, 20, 24)
PSFMC_FFT_SHAPE(500, 20, 25)
PSFMC_FFT_SHAPE(576, 24, 24)
PSFMC_FFT_SHAPE(600, 24, 25)
PSFMC_FFT_SHAPE(640, 20, 32)
PSFMC_FFT_SHAPE(720, 24, 30)
PSFMC_FFT_SHAPE(768, 24, 32)
PSFMC_FFT_SHAPE(800, 25, 32)
PSFMC_FFT_SHAPE(900, 30, 30)
PSFMC_FFT_SHAPE(960, 30, 32)
// sides with a factor 7 (round 2): the same rules
PSFMC_FFT_SHAPE(84, 7, 12)
PSFMC_FFT_SHAPE(98, 7, 14)
PSFMC_FFT_SHAPE(112, 14, 8)
PSFMC_FFT_SHAPE(126, 9, 14)
PSFMC_FFT_SHAPE(140, 10, 14)
PSFMC_FFT_SHAPE(168, 12, 14)
PSFMC_FFT_SHAPE(196, 14, 14)
PSFMC_FFT_SHAPE(210, 14, 15)
PSFMC_FFT_SHAPE(224, 14, 16)
PSFMC_FFT_SHAPE(252, 14, 18)
PSFMC_FFT_SHAPE(280, 14, 20)
PSFMC_FFT_SHAPE(294, 14, 21)
PSFMC_FFT_SHAPE(336, 16, 21)
PSFMC_FFT_SHAPE(350, 25, 14)
PSFMC_FFT_SHAPE(392, 14, 28)
PSFMC_FFT_SHAPE(420, 20, 21)
PSFMC_FFT_SHAPE(448, 16, 28)
PSFMC_FFT_SHAPE(504, 21, 24)
PSFMC_FFT_SHAPE(560, 20, 28)
PSFMC_FFT_SHAPE(630, 21, 30)
PSFMC_FFT_SHAPE(672, 24, 28)
PSFMC_FFT_SHAPE(700, 25, 28)
PSFMC_FFT_SHAPE(784, 28, 28)
PSFMC_FFT_SHAPE(840, 28, 30)
PSFMC_FFT_SHAPE(896, 28, 32)
// sides with a factor 11 or 13 (round 2): the same rules
PSFMC_FFT_SHAPE(88, 11, 8)
PSFMC_FFT_SHAPE(104, 13, 8)
PSFMC_FFT_SHAPE(110, 10, 11)
PSFMC_FFT_SHAPE(130, 10, 13)
PSFMC_FFT_SHAPE(132, 11, 12)
PSFMC_FFT_SHAPE(156, 12, 13)
PSFMC_FFT_SHAPE(176, 11, 16)
PSFMC_FFT_SHAPE(208, 13, 16)
PSFMC_FFT_SHAPE(220, 11, 20)
PSFMC_FFT_SHAPE(260, 13, 20)
PSFMC_FFT_SHAPE(264, 22, 12)
PSFMC_FFT_SHAPE(286, 22, 13)
PSFMC_FFT_SHAPE(308, 11, 28)
PSFMC_FFT_SHAPE(312, 24, 13)
PSFMC_FFT_SHAPE(330, 22, 15)
PSFMC_FFT_SHAPE(352, 22, 16)
PSFMC_FFT_SHAPE(364, 13, 28)
PSFMC_FFT_SHAPE(390, 15, 26)
PSFMC_FFT_SHAPE(416, 26, 16)
PSFMC_FFT_SHAPE(440, 20, 22)
PSFMC_FFT_SHAPE(484, 22, 22)
PSFMC_FFT_SHAPE(520, 20, 26)
PSFMC_FFT_SHAPE(528, 22, 24)
PSFMC_FFT_SHAPE(572, 22, 26)
PSFMC_FFT_SHAPE(616, 22, 28)
PSFMC_FFT_SHAPE(624, 24, 26)
PSFMC_FFT_SHAPE(650, 25, 26)
PSFMC_FFT_SHAPE(660, 22, 30)
PSFMC_FFT_SHAPE(676, 26, 26)
PSFMC_FFT_SHAPE(704, 22, 32)
PSFMC_FFT_SHAPE(728, 26, 28)
PSFMC_FFT_SHAPE(780, 26, 30)
PSFMC_FFT_SHAPE(832, 26, 32)
#undef PSFMC_FFT_SHAPE

// the output index lane t holds in register e, and whether that register holds one at all
template <int N> __device__ __forceinline__ constexpr int fft_k_of(int t, int e) {
    using S = FftShape<N>;
    return (t + S::T * (e % S::H)) + S::P * (e / S::H);
}
template <int N> __device__ __forceinline__ constexpr bool fft_slot_valid(int t, int e) {
    using S = FftShape<N>;
    return S::kExact || t + S::T * (e % S::H) < S::P;
}

// LDS doubles one transform needs for its exchange (one component at a time)
template <int N> constexpr int fft_lds_elems() { return FftShape<N>::T * FftShape<N>::LS; }

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

// real / imaginary part of exp(2 pi i * K / R), compile time: R | 32 from the 32nd-turn
// constants above, any other R <= 48 from the generated table
template <int R, int K> __device__ constexpr double tw_cos() {
    if constexpr (32 % R == 0) {
        constexpr int k = ((K % R) + R) % R * (32 / R);          // in 32nds of a turn
        return k <= 8 ? kCos32[k] : k <= 16 ? -kCos32[16 - k] : k <= 24 ? -kCos32[k - 16] : kCos32[32 - k];
    } else {
        static_assert(R <= kTrigMaxDen, "twiddle denominator outside the generated table");
        return kCosTab[R][((K % R) + R) % R];
    }
}
template <int R, int K> __device__ constexpr double tw_sin() {   // sin(2 pi K / R)
    if constexpr (32 % R == 0) return tw_cos<R, K - R / 4>();
    else return kSinTab[R][((K % R) + R) % R];
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

// the radix a length-R codelet splits off first: 13, 11, 7, then 5, then 3, then 2
template <int R> constexpr int dft_radix() { return R % 13 == 0 ? 13 : R % 11 == 0 ? 11 : R % 7 == 0 ? 7 : R % 5 == 0 ? 5 : R % 3 == 0 ? 3 : 2; }

// in-register DFT of R points, natural order in and out
template <int R, int SIGN, int RADIX = dft_radix<R>()> struct Dft;

// radix 3 / 5: decimation in time -- RADIX interleaved sub-transforms of length R / RADIX, then
// for each k the twiddled values go through one RADIX-point butterfly:
//   X[k + M q] = sum_j W_RADIX^(j q) (W_R^(j k) sub_j[k]),  M = R / RADIX
template <int R, int SIGN> struct Dft<R, SIGN, 3> {
    static constexpr int M = R / 3;
    static __device__ __forceinline__ void run(cd (&v)[R]) {
        cd s0[M], s1[M], s2[M];
#pragma unroll
        for (int i = 0; i < M; ++i) {
            s0[i] = v[3 * i];
            s1[i] = v[3 * i + 1];
            s2[i] = v[3 * i + 2];
        }
        Dft<M, SIGN>::run(s0);
        Dft<M, SIGN>::run(s1);
        Dft<M, SIGN>::run(s2);
        combine<0>(v, s0, s1, s2);
    }
    template <int K>
    static __device__ __forceinline__ void combine(cd (&v)[R], const cd (&s0)[M], const cd (&s1)[M],
                                                   const cd (&s2)[M]) {
        if constexpr (K < M) {
            constexpr double kS3 = SIGN * 0.86602540378443864676372317075293618347;   // sin(2 pi / 3)
            const cd a0 = s0[K], a1 = tw_mul<R, K, SIGN>(s1[K]), a2 = tw_mul<R, 2 * K, SIGN>(s2[K]);
            const cd s = cadd(a1, a2), d = csub(a1, a2);
            const cd m = cd{__builtin_fma(-0.5, s.x, a0.x), __builtin_fma(-0.5, s.y, a0.y)};
            v[K] = cadd(a0, s);
            v[K + M] = cd{__builtin_fma(-kS3, d.y, m.x), __builtin_fma(kS3, d.x, m.y)};        // m + i kS3 d
            v[K + 2 * M] = cd{__builtin_fma(kS3, d.y, m.x), __builtin_fma(-kS3, d.x, m.y)};    // m - i kS3 d
            combine<K + 1>(v, s0, s1, s2);
        }
    }
};

template <int R, int SIGN> struct Dft<R, SIGN, 5> {
    static constexpr int M = R / 5;
    static __device__ __forceinline__ void run(cd (&v)[R]) {
        cd s0[M], s1[M], s2[M], s3[M], s4[M];
#pragma unroll
        for (int i = 0; i < M; ++i) {
            s0[i] = v[5 * i];
            s1[i] = v[5 * i + 1];
            s2[i] = v[5 * i + 2];
            s3[i] = v[5 * i + 3];
            s4[i] = v[5 * i + 4];
        }
        Dft<M, SIGN>::run(s0);
        Dft<M, SIGN>::run(s1);
        Dft<M, SIGN>::run(s2);
        Dft<M, SIGN>::run(s3);
        Dft<M, SIGN>::run(s4);
        combine<0>(v, s0, s1, s2, s3, s4);
    }
    template <int K>
    static __device__ __forceinline__ void combine(cd (&v)[R], const cd (&s0)[M], const cd (&s1)[M],
                                                   const cd (&s2)[M], const cd (&s3)[M], const cd (&s4)[M]) {
        if constexpr (K < M) {
            constexpr double c1 = 0.30901699437494742410229341718281905886;    // cos(2 pi / 5)
            constexpr double c2 = -0.80901699437494742410229341718281905886;   // cos(4 pi / 5)
            constexpr double q1 = SIGN * 0.95105651629515357211643933337938214340;   // sin(2 pi / 5)
            constexpr double q2 = SIGN * 0.58778525229247312916870595463907276860;   // sin(4 pi / 5)
            const cd a0 = s0[K], a1 = tw_mul<R, K, SIGN>(s1[K]), a2 = tw_mul<R, 2 * K, SIGN>(s2[K]),
                     a3 = tw_mul<R, 3 * K, SIGN>(s3[K]), a4 = tw_mul<R, 4 * K, SIGN>(s4[K]);
            const cd p1 = cadd(a1, a4), p2 = cadd(a2, a3), d1 = csub(a1, a4), d2 = csub(a2, a3);
            v[K] = cadd(a0, cadd(p1, p2));
            const cd m1 = cd{__builtin_fma(c2, p2.x, __builtin_fma(c1, p1.x, a0.x)),
                             __builtin_fma(c2, p2.y, __builtin_fma(c1, p1.y, a0.y))};
            const cd m2 = cd{__builtin_fma(c1, p2.x, __builtin_fma(c2, p1.x, a0.x)),
                             __builtin_fma(c1, p2.y, __builtin_fma(c2, p1.y, a0.y))};
            const cd n1 = cd{__builtin_fma(q2, d2.x, q1 * d1.x), __builtin_fma(q2, d2.y, q1 * d1.y)};
            const cd n2 = cd{__builtin_fma(-q1, d2.x, q2 * d1.x), __builtin_fma(-q1, d2.y, q2 * d1.y)};
            v[K + M] = cd{m1.x - n1.y, m1.y + n1.x};              // m1 + i n1
            v[K + 4 * M] = cd{m1.x + n1.y, m1.y - n1.x};
            v[K + 2 * M] = cd{m2.x - n2.y, m2.y + n2.x};          // m2 + i n2
            v[K + 3 * M] = cd{m2.x + n2.y, m2.y - n2.x};
            combine<K + 1>(v, s0, s1, s2, s3, s4);
        }
    }
};

// radix 7: the same scheme; the six twiddled values pair up into sums p_j = a_j + a_(7-j) and
// differences d_j = a_j - a_(7-j), and X[k + M q], X[k + M (7 - q)] = m_q +- i n_q with
//   m_q = a_0 + sum_j cos(2 pi j q / 7) p_j,   n_q = SIGN sum_j sin(2 pi j q / 7) d_j
template <int R, int SIGN> struct Dft<R, SIGN, 7> {
    static constexpr int M = R / 7;
    static __device__ __forceinline__ void run(cd (&v)[R]) {
        cd s[7][M];
#pragma unroll
        for (int j = 0; j < 7; ++j)
#pragma unroll
            for (int i = 0; i < M; ++i) s[j][i] = v[7 * i + j];
#pragma unroll
        for (int j = 0; j < 7; ++j) Dft<M, SIGN>::run(s[j]);
        combine<0>(v, s);
    }
    template <int K>
    static __device__ __forceinline__ void combine(cd (&v)[R], const cd (&s)[7][M]) {
        if constexpr (K < M) {
            constexpr double c1 = 0.62348980185873353052500488400423981063;     // cos(2 pi / 7)
            constexpr double c2 = -0.22252093395631440428890256449679475947;    // cos(4 pi / 7)
            constexpr double c3 = -0.90096886790241912623610231950744505116;    // cos(6 pi / 7)
            constexpr double q1 = SIGN * 0.78183148246802980870844452667405775023;   // sin(2 pi / 7)
            constexpr double q2 = SIGN * 0.97492791218182360701813168299393121723;   // sin(4 pi / 7)
            constexpr double q3 = SIGN * 0.43388373911755812047576833284835875461;   // sin(6 pi / 7)
            const cd a0 = s[0][K], a1 = tw_mul<R, K, SIGN>(s[1][K]), a2 = tw_mul<R, 2 * K, SIGN>(s[2][K]),
                     a3 = tw_mul<R, 3 * K, SIGN>(s[3][K]), a4 = tw_mul<R, 4 * K, SIGN>(s[4][K]),
                     a5 = tw_mul<R, 5 * K, SIGN>(s[5][K]), a6 = tw_mul<R, 6 * K, SIGN>(s[6][K]);
            const cd p1 = cadd(a1, a6), p2 = cadd(a2, a5), p3 = cadd(a3, a4);
            const cd d1 = csub(a1, a6), d2 = csub(a2, a5), d3 = csub(a3, a4);
            v[K] = cadd(cadd(a0, p1), cadd(p2, p3));
            auto mix = [](cd base, double x1, cd u1, double x2, cd u2, double x3, cd u3) {
                return cd{__builtin_fma(x3, u3.x, __builtin_fma(x2, u2.x, __builtin_fma(x1, u1.x, base.x))),
                          __builtin_fma(x3, u3.y, __builtin_fma(x2, u2.y, __builtin_fma(x1, u1.y, base.y)))};
            };
            const cd zero = cd{0.0, 0.0};
            const cd m1 = mix(a0, c1, p1, c2, p2, c3, p3), n1 = mix(zero, q1, d1, q2, d2, q3, d3);
            const cd m2 = mix(a0, c2, p1, c3, p2, c1, p3), n2 = mix(zero, q2, d1, -q3, d2, -q1, d3);
            const cd m3 = mix(a0, c3, p1, c1, p2, c2, p3), n3 = mix(zero, q3, d1, -q1, d2, q2, d3);
            v[K + M] = cd{m1.x - n1.y, m1.y + n1.x};              // m1 + i n1
            v[K + 6 * M] = cd{m1.x + n1.y, m1.y - n1.x};
            v[K + 2 * M] = cd{m2.x - n2.y, m2.y + n2.x};
            v[K + 5 * M] = cd{m2.x + n2.y, m2.y - n2.x};
            v[K + 3 * M] = cd{m3.x - n3.y, m3.y + n3.x};
            v[K + 4 * M] = cd{m3.x + n3.y, m3.y - n3.x};
            combine<K + 1>(v, s);
        }
    }
};

// radix 11 and 13: the radix-7 scheme written once for any odd prime PR, its cosines and sines taken
// from the generated table at compile time (H = (PR - 1) / 2 pairs of sums and differences)
template <int R, int SIGN, int PR> struct DftPrime {
    static constexpr int M = R / PR, H = (PR - 1) / 2;
    static __device__ __forceinline__ void run(cd (&v)[R]) {
        cd s[PR][M];
#pragma unroll
        for (int j = 0; j < PR; ++j)
#pragma unroll
            for (int i = 0; i < M; ++i) s[j][i] = v[PR * i + j];
#pragma unroll
        for (int j = 0; j < PR; ++j) Dft<M, SIGN>::run(s[j]);
        combine<0>(v, s);
    }
    template <int K, int J>
    static __device__ __forceinline__ void pairs(const cd (&s)[PR][M], cd (&p)[H], cd (&d)[H]) {
        if constexpr (J <= H) {
            const cd a = tw_mul<R, J * K, SIGN>(s[J][K]), b = tw_mul<R, (PR - J) * K, SIGN>(s[PR - J][K]);
            p[J - 1] = cadd(a, b);
            d[J - 1] = csub(a, b);
            pairs<K, J + 1>(s, p, d);
        }
    }
    template <int Q, int J>
    static __device__ __forceinline__ void mix(cd& m, cd& n, const cd (&p)[H], const cd (&d)[H]) {
        if constexpr (J <= H) {
            constexpr double c = tw_cos<PR, (J * Q) % PR>();
            constexpr double q = SIGN * tw_sin<PR, (J * Q) % PR>();
            m = cd{__builtin_fma(c, p[J - 1].x, m.x), __builtin_fma(c, p[J - 1].y, m.y)};
            n = cd{__builtin_fma(q, d[J - 1].x, n.x), __builtin_fma(q, d[J - 1].y, n.y)};
            mix<Q, J + 1>(m, n, p, d);
        }
    }
    template <int K, int Q>
    static __device__ __forceinline__ void outputs(cd (&v)[R], cd a0, const cd (&p)[H], const cd (&d)[H]) {
        if constexpr (Q <= H) {
            cd m = a0, n = cd{0.0, 0.0};
            mix<Q, 1>(m, n, p, d);
            v[K + Q * M] = cd{m.x - n.y, m.y + n.x};               // m + i n
            v[K + (PR - Q) * M] = cd{m.x + n.y, m.y - n.x};
            outputs<K, Q + 1>(v, a0, p, d);
        }
    }
    template <int K>
    static __device__ __forceinline__ void combine(cd (&v)[R], const cd (&s)[PR][M]) {
        if constexpr (K < M) {
            cd p[H], d[H];
            pairs<K, 1>(s, p, d);
            const cd a0 = s[0][K];
            cd sum = a0;
#pragma unroll
            for (int j = 0; j < H; ++j) sum = cadd(sum, p[j]);
            v[K] = sum;
            outputs<K, 1>(v, a0, p, d);
            combine<K + 1>(v, s);
        }
    }
};
template <int R, int SIGN> struct Dft<R, SIGN, 11> : DftPrime<R, SIGN, 11> {};
template <int R, int SIGN> struct Dft<R, SIGN, 13> : DftPrime<R, SIGN, 13> {};

// radix 2 (what is left once the 13s, 11s, 7s, 5s and 3s are split off: the power-of-two codelets)
template <int R, int SIGN> struct Dft<R, SIGN, 2> {
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
template <int SIGN> struct Dft<1, SIGN, 2> {
    static __device__ __forceinline__ void run(cd (&)[1]) {}
};
template <int SIGN> struct Dft<2, SIGN, 2> {
    static __device__ __forceinline__ void run(cd (&v)[2]) {
        const cd a = v[0], b = v[1];
        v[0] = cadd(a, b);
        v[1] = csub(a, b);
    }
};

// wave_lds_sync (the wave-local LDS hand-off the transforms use between stages): psfmc_device.h

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
// `t` in [0,T).  `w` from load_twiddles (forward table).  Converged call only: every lane of
// the wave calls; `active` = false for the idle tail lanes of a wave whose T does not divide
// 64 (they compute on whatever they hold and must not write LDS).
// The exchange goes through LDS one component at a time (real parts, then
// imaginary parts): half the LDS footprint per wave, which is what bounds how many
// waves a CU can hold, for the same number of LDS bytes moved.
template <int N, int SIGN, int TWM = PSFMC_TW_MODE>
__device__ __forceinline__ void fft_wave(cd (&v)[FftShape<N>::R], const cd* w /* [TwRegs<N, TWM>::value] */,
                                         const cd* __restrict__ table, int t, double* __restrict__ xbuf,
                                         const cd* __restrict__ twl, bool active = true) {
    using S = FftShape<N>;
    constexpr int P = S::P, T = S::T, H = S::H;
    if constexpr (S::R == P) {
        Dft<P, SIGN>::run(v);
    } else {                                        // the first P registers hold the input
        cd a[P];
#pragma unroll
        for (int c = 0; c < P; ++c) a[c] = v[c];
        Dft<P, SIGN>::run(a);
#pragma unroll
        for (int c = 0; c < P; ++c) v[c] = a[c];
    }
#pragma unroll
    for (int c = 1; c < P; ++c) {
        cd wc;
        if constexpr (fft_tw_mode<N, TWM>() == 0) wc = w[c];
        else if constexpr (fft_tw_mode<N, TWM>() == 1) wc = twl[c * T + t];
        else wc = table[t * c];
        v[c] = cmul(v[c], SIGN < 0 ? wc : cconj(wc));
    }
    double* row = xbuf + t * S::LS;
    // lane t's stage-2 transforms work on c = t + T h; a c >= P (only when T does not divide P)
    // reads a valid column instead and its results are never used
    int col[H];
#pragma unroll
    for (int h = 0; h < H; ++h) col[h] = (S::kExact || t + T * h < P) ? t + T * h : 0;
    cd z[H][T];
    if (S::kFull || active) {
#pragma unroll
        for (int c = 0; c < P; ++c) row[c] = v[c].x;
    }
    wave_lds_sync();
#pragma unroll
    for (int h = 0; h < H; ++h)
#pragma unroll
        for (int b = 0; b < T; ++b) z[h][b].x = xbuf[b * S::LS + col[h]];
    wave_lds_sync();
    if (S::kFull || active) {
#pragma unroll
        for (int c = 0; c < P; ++c) row[c] = v[c].y;
    }
    wave_lds_sync();
#pragma unroll
    for (int h = 0; h < H; ++h)
#pragma unroll
        for (int b = 0; b < T; ++b) z[h][b].y = xbuf[b * S::LS + col[h]];
    wave_lds_sync();
#pragma unroll
    for (int h = 0; h < H; ++h) {
        Dft<T, SIGN>::run(z[h]);
#pragma unroll
        for (int d = 0; d < T; ++d) v[h + H * d] = z[h][d];
    }
}

// Output order -> input order for a shape whose T does not divide P: register e of lane t
// holds X[fft_k_of(t, e)]; afterwards register a holds X[T a + t] (what a following transform
// of the same data expects).  Through the transform's LDS region, one component at a time
// (T (P+1) doubles >= N).  Converged call; idle tail lanes pass active = false.
template <int N>
__device__ __forceinline__ void fft_regroup(cd (&v)[FftShape<N>::R], int t, double* __restrict__ xbuf,
                                            bool active) {
    using S = FftShape<N>;
    constexpr int P = S::P, T = S::T, R = S::R;
    double re[P];
#pragma unroll
    for (int e = 0; e < R; ++e)
        if ((S::kFull || active) && fft_slot_valid<N>(t, e)) xbuf[fft_k_of<N>(t, e)] = v[e].x;
    wave_lds_sync();
#pragma unroll
    for (int a = 0; a < P; ++a) re[a] = xbuf[T * a + t];
    wave_lds_sync();
#pragma unroll
    for (int e = 0; e < R; ++e)
        if ((S::kFull || active) && fft_slot_valid<N>(t, e)) xbuf[fft_k_of<N>(t, e)] = v[e].y;
    wave_lds_sync();
#pragma unroll
    for (int a = 0; a < P; ++a) v[a] = cd{re[a], xbuf[T * a + t]};
#pragma unroll
    for (int a = P; a < R; ++a) v[a] = cd{0.0, 0.0};
    wave_lds_sync();
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
// `w2_lds` (optional): the stage-2 twiddles [k2][n3] = W_N^(R1 n3 k2) from a 64-entry LDS table instead of the
// eight per-lane registers w2 (32 VGPRs: what keeps the R1 = 16 column kernel from a third wave per SIMD).
template <int N, int SIGN>
__device__ __forceinline__ void fft_wave3(cd (&v)[Fft3Shape<N>::R1], const cd (&w1)[fft3_w1_regs<N>()],
                                          const cd (&w2)[8], const cd* __restrict__ table, int t,
                                          double* __restrict__ lds, const cd* __restrict__ w1_lds = nullptr,
                                          const cd* __restrict__ w2_lds = nullptr) {
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
        for (int k2 = 1; k2 < 8; ++k2) {
            const cd wk = w2_lds ? w2_lds[k2 * 8 + n3] : w2[k2];
            z[i][k2] = cmul(z[i][k2], SIGN < 0 ? wk : cconj(wk));
        }
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

// ---------------------------------------------------------------------------
// The wave-wide three-stage transform for N = R1 * R2 * R3 on L = R2 * R3 <= 64 lanes of a wave (round 3: the
// columns of every built side above 512, which ran on the two-stage engine at 20 ... 32 complex registers per
// lane and one wave per SIMD).  Same scheme as fft_wave3: lane t = R3 n2 + n3 (t < L) holds v[a] = x[L a + t]
// (a < R1) on entry;
//   stage 1  radix-R1 over a, twiddle W_N^(t k1);        exchange E1[k1][t]
//   stage 2  R1 R3 radix-R2 transforms over n2: lane (g = t / R3, n3) takes k1 = g + R2 i, i < NB2 =
//            ceil(R1 / R2), where k1 < R1; twiddle W_L^(n3 k2);   exchange E2[k2][n3][k1]
//   stage 3  R1 R2 radix-R3 transforms over n3 on all 64 lanes: lane t takes c = t + 64 q = k1 + R1 k2,
//            q < NB3 = ceil(R1 R2 / 64), where c < R1 R2
// and on exit   o[q][k3] = X[(t + 64 q) + R1 R2 k3]   (valid iff t + 64 q < R1 R2).
// Its inverse is the mirror image below (fft_wave3g_inv), which takes exactly this output layout.
// ---------------------------------------------------------------------------
struct Fft3gPick { int r2, r3; };
#ifndef PSFMC_COLS3G_ROUND4_SIDES
#define PSFMC_COLS3G_ROUND4_SIDES 1
#endif
// The shape of a side, {0, 0} = none (its columns stay on the two-stage engine).  Empirical: every candidate
// factorisation was timed against the two-stage kernel on an MI355X (tools/cols3g_shapes.hip;
// profiles/r3_cols3g_shapes.txt).  What wins: R2 = 4 (stage 2 is then NB2 = R1 / 4 cheap radix-4 passes and stage 3
// one radix-R3 pass, R3 <= 15: the widest stage holds max(R1, R3) complex registers) or 8 x 8 on all 64 lanes;
// R3 = 16 and the 5 x 10 / 10 x 5 splits of 50 lanes at R1 >= 13 (650, 700, 800) lose to the two-stage kernel, as
// does every side whose two-stage shape has T <= 21 lanes per transform (the small sides); 250, 294, 330, whose
// two-stage kernels hold 21 ... 25 complex registers at one wave per SIMD, win 14 ... 25 % on it.
constexpr Fft3gPick fft3g_pick(int n) {
    switch (n) {
        // Round 4 re-surveyed the shapes (profiles/r4_cols3g_resurvey.txt: 233 variants of the 47 sides) once the column's
        // addresses sat on a scalar base: round 3's ranking no longer held at thirteen sides.  Alone / whole step
        // (r4_cols3g_resurvey_step.txt, same box, both libraries twice):
        //   384 (4, 12) -> (8, 8) -17 % / +5.8 %     392 (4, 14) -> (7, 8) -31 % / +11.8 %    448 (4, 16) -> (8, 7) -17 % / +9.0 %
        //   480 (4, 15) -> (6, 10) -11 % / +3.7 %    504 (4, 14) -> (7, 8) -6 % / +4.2 %      600 (5, 12) -> (6, 10) -12 % / +2.9 %
        //   640, 704 (8, 8) -> (4, 16) -5 % / +2.7 %, +2.2 %    728, 784 (4, 14) -> (7, 8) -4 % / +1.4 %, +2.0 %
        //   780 (4, 13) -> (6, 10) -8 % / +4.0 %
        // taken; 676 (4, 13) -> (13, 4) and 768 (8, 8) -> (4, 16), -5 % alone, move the step by -0.8 % / -0.2 %: not taken.
        case 264: case 308: case 352: case 484: return {4, 11};
        case 528: case 576: return {4, 12};
        case 312: case 364: case 416: case 520: case 572: case 624: case 676: return {4, 13};
        case 560: case 616: case 672: case 840: case 896: return {4, 14};
        case 900: return {4, 15};
        case 640: case 704: return {4, 16};
        case 330: case 440: return {5, 11};
        case 250: case 500: return {5, 10};
        case 294: return {7, 7};
        case 660: case 720: return {5, 12};
        case 480: case 600: case 780: return {6, 10};
        case 392: case 504: case 728: case 784: return {7, 8};
        case 448: return {8, 7};
        case 384: case 512: case 768: case 832: case 960: case 1024: return {8, 8};
        // round 4: sides above 1024 -- no two-stage shape (P, T <= 32) reaches them; all 64 lanes, R1 = 18 ... 32
        case 1152: case 1280: case 1536: case 2048: return {8, 8};
#if PSFMC_COLS3G_ROUND4_SIDES
        // round 4, second survey (profiles/r4_cols3g_more_sides.txt): with the column's addresses on a scalar base the
        // three-stage kernel also wins at sides with 5 ... 10 elements per lane, where round 3 measured it slower:
        // kernel alone -5 ... -25 %, whole step (profiles/r4_cols3g_round4_sides_step.txt) 280 +2.2 %, 288 +1.6 %,
        // 300 +2.9 %, 336 +5.0 %, 350 +3.3 %, 360 +0.6 %, 630 +2.7 %.  320 = 5 x (8 x 8) and 420 = 7 x (5 x 12) were
        // faster alone (-3 %, -11 %) and slower in the step (-1.2 %, -1.7 %): not taken.
        case 280: case 336: return {7, 8};
        case 288: return {6, 8};
        case 300: return {5, 12};
        case 350: return {5, 10};
        case 360: return {6, 10};
        case 630: return {7, 9};
#endif
        default: return {0, 0};
    }
}
template <int N, int R2_ = fft3g_pick(N).r2, int R3_ = fft3g_pick(N).r3> struct Fft3gShape {
    static constexpr bool kBuilt = R2_ > 0;
    static constexpr int kN = N;
    static constexpr int R2 = kBuilt ? R2_ : 1, R3 = kBuilt ? R3_ : 1, L = R2 * R3, R1 = kBuilt ? N / L : 1;
    static_assert(!kBuilt || (R1 * L == N && L <= 64 && R1 <= 32 && R1 >= 4), "N = R1 R2 R3");
    static constexpr int NB2 = (R1 + R2 - 1) / R2, NB3 = (R1 * R2 + 63) / 64;
    static constexpr int S1 = 64 + R3;                      // >= 64 and = R3 (mod 32): stage 2's reads are conflict-free
    static constexpr int S2 = R1 | 1;
};
template <class S> constexpr int fft3g_lds_doubles() {
    constexpr int a = S::R1 * S::S1, b = S::L * S::S2;
    return a > b ? a : b;
}
template <class S> __device__ __forceinline__ bool fft3g_valid(int t, int q) { return t + 64 * q < S::R1 * S::R2; }
template <class S> __device__ __forceinline__ int fft3g_index(int t, int q, int k3) {
    return t + 64 * q + S::R1 * S::R2 * k3;
}

// `w1_lds`: the workgroup's stage-1 twiddle table [k1][t] = W_N^(t k1) in LDS (row stride 64);
// w2[k2] = W_N^(R1 n3 k2).  Lanes t >= L carry no input; they work in stage 3 only.
// HALF1: the table holds k1 < R1 / 2 only and `w1h` = W_N^(t R1 / 2) is the lane's factor for the other half,
// W_N^(t (k1 + R1 / 2)) = W_N^(t k1) w1h -- R1 / 2 more complex multiplies per transform for half the table's LDS
// (what lets eight row waves of 2048 share a CU, psfmc_rows3_path.h).
template <class S, int SIGN, bool HALF1 = false>
__device__ __forceinline__ void fft_wave3g(cd (&v)[S::R1], cd (&o)[S::NB3][S::R3], const cd (&w2)[S::R2], int t,
                                           double* __restrict__ lds, const cd* __restrict__ w1_lds,
                                           cd w1h = cd{1.0, 0.0}) {
    constexpr int R1 = S::R1, R2 = S::R2, R3 = S::R3, L = S::L, NB2 = S::NB2, NB3 = S::NB3, S1 = S::S1, S2 = S::S2;
    const bool lane_in = L == 64 || t < L;
    const int tl = lane_in ? t : 0;
    const int n3 = tl % R3, g = tl / R3;
    Dft<R1, SIGN>::run(v);
    if constexpr (HALF1) {
        static_assert(R1 % 2 == 0, "half table");
        constexpr int H = R1 / 2;
        const cd wh = SIGN < 0 ? w1h : cconj(w1h);
        v[H] = cmul(v[H], wh);
#pragma unroll
        for (int k = 1; k < H; ++k) {
            const cd w0 = w1_lds[k * 64 + t];
            const cd wk = SIGN < 0 ? w0 : cconj(w0);
            v[k] = cmul(v[k], wk);
            v[k + H] = cmul(v[k + H], cmul(wk, wh));
        }
    } else {
#pragma unroll
    for (int k = 1; k < R1; ++k) {
        const cd wk = w1_lds[k * 64 + t];
        v[k] = cmul(v[k], SIGN < 0 ? wk : cconj(wk));
    }
    }
    cd z[NB2][R2];
    int k1s[NB2];
    bool ok2[NB2];
#pragma unroll
    for (int i = 0; i < NB2; ++i) {
        ok2[i] = lane_in && g + R2 * i < R1;
        k1s[i] = ok2[i] ? g + R2 * i : 0;                      // an idle slot works on a valid row and is never stored
    }
#pragma unroll
    for (int k = 0; k < R1; ++k) lds[k * S1 + t] = v[k].x;     // S1 >= 64: lanes t >= L write columns nobody reads
    wave_lds_sync();
#pragma unroll
    for (int i = 0; i < NB2; ++i)
#pragma unroll
        for (int n2 = 0; n2 < R2; ++n2) z[i][n2].x = lds[k1s[i] * S1 + n2 * R3 + n3];
    wave_lds_sync();
#pragma unroll
    for (int k = 0; k < R1; ++k) lds[k * S1 + t] = v[k].y;
    wave_lds_sync();
#pragma unroll
    for (int i = 0; i < NB2; ++i)
#pragma unroll
        for (int n2 = 0; n2 < R2; ++n2) z[i][n2].y = lds[k1s[i] * S1 + n2 * R3 + n3];
    wave_lds_sync();
#pragma unroll
    for (int i = 0; i < NB2; ++i) {
        Dft<R2, SIGN>::run(z[i]);
#pragma unroll
        for (int k2 = 1; k2 < R2; ++k2) z[i][k2] = cmul(z[i][k2], SIGN < 0 ? w2[k2] : cconj(w2[k2]));
    }
    int k1o[NB3], k2o[NB3];
#pragma unroll
    for (int q = 0; q < NB3; ++q) {
        const int c = fft3g_valid<S>(t, q) ? t + 64 * q : 0;
        k1o[q] = c % R1;
        k2o[q] = c / R1;
    }
#pragma unroll
    for (int i = 0; i < NB2; ++i)
        if (ok2[i]) {
#pragma unroll
            for (int k2 = 0; k2 < R2; ++k2) lds[(k2 * R3 + n3) * S2 + k1s[i]] = z[i][k2].x;
        }
    wave_lds_sync();
#pragma unroll
    for (int q = 0; q < NB3; ++q)
#pragma unroll
        for (int m = 0; m < R3; ++m) o[q][m].x = lds[(k2o[q] * R3 + m) * S2 + k1o[q]];
    wave_lds_sync();
#pragma unroll
    for (int i = 0; i < NB2; ++i)
        if (ok2[i]) {
#pragma unroll
            for (int k2 = 0; k2 < R2; ++k2) lds[(k2 * R3 + n3) * S2 + k1s[i]] = z[i][k2].y;
        }
    wave_lds_sync();
#pragma unroll
    for (int q = 0; q < NB3; ++q)
#pragma unroll
        for (int m = 0; m < R3; ++m) o[q][m].y = lds[(k2o[q] * R3 + m) * S2 + k1o[q]];
    wave_lds_sync();
#pragma unroll
    for (int q = 0; q < NB3; ++q) Dft<R3, SIGN>::run(o[q]);
}

// The inverse of fft_wave3g as its mirror image (decimation in time against the forward transform's decimation
// in frequency): it takes the forward transform's OUTPUT layout o[q][k3] = Y[(t + 64 q) + R1 R2 k3] and leaves
// v[a] = y[L a + t], the forward transform's INPUT layout -- a convolution needs no regrouping between the two
// (an LDS round trip of the whole column and four barriers, 15 % of the column kernel's time at 512 and 1024).
//   stage A  radix-R3 over k3 in place, twiddle conj W_N^(n3 c), c = t + 64 q = k1 + R1 k2   (table wB[n3][c])
//   exchange E2 backwards;  stage B  radix-R2 over k2, twiddle conj W_N^(R3 n2 k1) = conj w1[k1][R3 n2]
//   exchange E1 backwards;  stage C  radix-R1 over k1.          Unnormalised, like the forward transform.
template <class S>
__device__ __forceinline__ void fft_wave3g_inv(cd (&o)[S::NB3][S::R3], cd (&v)[S::R1], int t, double* __restrict__ lds,
                                               const cd* __restrict__ w1_lds, const cd* __restrict__ wB_lds) {
    constexpr int R1 = S::R1, R2 = S::R2, R3 = S::R3, L = S::L, NB2 = S::NB2, NB3 = S::NB3, S1 = S::S1, S2 = S::S2;
    constexpr int M = R1 * R2;
    const bool lane_in = L == 64 || t < L;
    const int tl = lane_in ? t : 0;
    const int n3 = tl % R3, g = tl / R3;
    int k1o[NB3], k2o[NB3];
    bool ok3[NB3];
#pragma unroll
    for (int q = 0; q < NB3; ++q) {
        ok3[q] = fft3g_valid<S>(t, q);
        const int c = ok3[q] ? t + 64 * q : 0;
        k1o[q] = c % R1;
        k2o[q] = c / R1;
        Dft<R3, +1>::run(o[q]);
#pragma unroll
        for (int m = 1; m < R3; ++m) o[q][m] = cmul(o[q][m], cconj(wB_lds[m * M + c]));
    }
    cd z[NB2][R2];
    int k1s[NB2];
    bool ok2[NB2];
#pragma unroll
    for (int i = 0; i < NB2; ++i) {
        ok2[i] = lane_in && g + R2 * i < R1;
        k1s[i] = ok2[i] ? g + R2 * i : 0;
    }
#pragma unroll
    for (int q = 0; q < NB3; ++q)
        if (ok3[q]) {
#pragma unroll
            for (int m = 0; m < R3; ++m) lds[(k2o[q] * R3 + m) * S2 + k1o[q]] = o[q][m].x;
        }
    wave_lds_sync();
#pragma unroll
    for (int i = 0; i < NB2; ++i)
#pragma unroll
        for (int k2 = 0; k2 < R2; ++k2) z[i][k2].x = lds[(k2 * R3 + n3) * S2 + k1s[i]];
    wave_lds_sync();
#pragma unroll
    for (int q = 0; q < NB3; ++q)
        if (ok3[q]) {
#pragma unroll
            for (int m = 0; m < R3; ++m) lds[(k2o[q] * R3 + m) * S2 + k1o[q]] = o[q][m].y;
        }
    wave_lds_sync();
#pragma unroll
    for (int i = 0; i < NB2; ++i)
#pragma unroll
        for (int k2 = 0; k2 < R2; ++k2) z[i][k2].y = lds[(k2 * R3 + n3) * S2 + k1s[i]];
    wave_lds_sync();
#pragma unroll
    for (int i = 0; i < NB2; ++i) {
        Dft<R2, +1>::run(z[i]);
#pragma unroll
        for (int n2 = 1; n2 < R2; ++n2) z[i][n2] = cmul(z[i][n2], cconj(w1_lds[k1s[i] * 64 + R3 * n2]));
    }
#pragma unroll
    for (int i = 0; i < NB2; ++i)
        if (ok2[i]) {
#pragma unroll
            for (int n2 = 0; n2 < R2; ++n2) lds[k1s[i] * S1 + n2 * R3 + n3] = z[i][n2].x;
        }
    wave_lds_sync();
#pragma unroll
    for (int k = 0; k < R1; ++k) v[k].x = lds[k * S1 + tl];
    wave_lds_sync();
#pragma unroll
    for (int i = 0; i < NB2; ++i)
        if (ok2[i]) {
#pragma unroll
            for (int n2 = 0; n2 < R2; ++n2) lds[k1s[i] * S1 + n2 * R3 + n3] = z[i][n2].y;
        }
    wave_lds_sync();
#pragma unroll
    for (int k = 0; k < R1; ++k) v[k].y = lds[k * S1 + tl];
    wave_lds_sync();
    Dft<R1, +1>::run(v);
}

}  // namespace psfmc
