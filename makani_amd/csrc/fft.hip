// Longitudinal real FFT kernels (K1 / K4 of SURVEY.md section 2a) for gfx950.
//
// One workgroup transforms G rows (G consecutive channels at one latitude) held in
// LDS: the real row of length N is packed as a complex row of H = N/2 points,
// transformed by in-LDS Stockham passes (radices fixed at compile time per H), and
// split / merged by the usual real-FFT post / pre-processing step.  Only the
// first `mmax` modes are written (analysis) or read (synthesis, implicit zero
// padding).  The Fourier side uses the private layout xf[m][k][bc] so that the
// Legendre GEMM reads row-major [K x 2BC] panels per m; the G rows of a workgroup
// make G*8-byte contiguous segments per (m, k).
//
// HBM-bound by design: each input element is read once, each kept mode written once.
#include "common.h"
#include "../../include/makani_amd.h"

#include <hip/hip_bf16.h>
#include <cstdlib>
#include <type_traits>
#include <utility>

namespace {

// Layout of the Fourier rows xf: element (m, k, bc) at ((m * sm + k * sk) * BC + bc) complex numbers.
//   [M][K][BC] (default): sm = K, sk = 1.     [K][M][BC] (latitude major, distributed SHT): sm = 1, sk = M.
// Peer-major (Cp > 0, split kernels only): the channels c = bc % C are cut into blocks of Cp (one per rank of the
// latitude group) and the block index is the OUTERMOST axis, [P][K][M][B][Cp]: element at
//   p * pstride + ((m * sm + k * sk) * (B * Cp) + b * Cp + c % Cp),  p = c / Cp, b = bc / C, sm = 1, sk = M
// -- the send / receive buffer of the channel <-> latitude all-to-all as it is, no pack or concatenate copy.
struct XfLayout {
    int sm, sk;
    int C, Cp, Bn;        // Cp = 0: plain layouts
    long long pstride;    // K * M * Bn * Cp
};
static thread_local XfLayout g_xl = {0, 0, 0, 0, 0, 0};   // set by the C entry points before they launch
static thread_local double* g_rowsums = nullptr;          // mk_irfft_sums: per-row (sum, sum of squares) accumulators of the output

// per-tile channel addressing: all G rows of a split-kernel tile lie in one (batch item, channel block) when Cp % G == 0
struct XfChan {
    size_t pbase;
    int bcx0, BCx;
};
__device__ __forceinline__ XfChan xf_chan(const XfLayout& xl, int bc0, int BC) {
    XfChan c{0, bc0, BC};
    if (xl.Cp) {
        const int b = bc0 / xl.C, ch = bc0 - b * xl.C, pp = ch / xl.Cp;
        c.pbase = (size_t)pp * (size_t)xl.pstride;
        c.bcx0 = b * xl.Cp + (ch - pp * xl.Cp);
        c.BCx = xl.Bn * xl.Cp;
    }
    return c;
}


// ---------------------------------------------------------------------------
// compile-time helpers
// ---------------------------------------------------------------------------
template <int I>
using IC = std::integral_constant<int, I>;

template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
    (f(IC<Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

constexpr double kPi = 3.14159265358979323846264338327950288;

constexpr double cx_reduce(double x) {  // to [-pi, pi]
    while (x > kPi) x -= 2.0 * kPi;
    while (x < -kPi) x += 2.0 * kPi;
    return x;
}
constexpr double cx_sin(double x) {
    x = cx_reduce(x);
    double term = x, sum = x;
    for (int n = 1; n < 20; ++n) {
        term *= -x * x / ((2.0 * n) * (2.0 * n + 1.0));
        sum += term;
    }
    return sum;
}
constexpr double cx_cos(double x) {
    x = cx_reduce(x);
    double term = 1.0, sum = 1.0;
    for (int n = 1; n < 20; ++n) {
        term *= -x * x / ((2.0 * n - 1.0) * (2.0 * n));
        sum += term;
    }
    return sum;
}

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// multiply by -i
__device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }

// ---------------------------------------------------------------------------
// small forward DFTs in registers (sign -1)
// ---------------------------------------------------------------------------
template <int R>
struct Dft;

template <>
struct Dft<2> {
    static __device__ __forceinline__ void run(float2 (&v)[2]) {
        const float2 a = v[0], b = v[1];
        v[0] = cadd(a, b);
        v[1] = csub(a, b);
    }
};

template <>
struct Dft<3> {
    static __device__ __forceinline__ void run(float2 (&v)[3]) {
        constexpr float s = 0.86602540378443864676f;
        const float2 t = cadd(v[1], v[2]);
        const float2 d = csub(v[1], v[2]);
        const float2 m = make_float2(v[0].x - 0.5f * t.x, v[0].y - 0.5f * t.y);
        const float2 q = mul_mi(make_float2(s * d.x, s * d.y));  // -i s d
        v[0] = cadd(v[0], t);
        v[1] = cadd(m, q);
        v[2] = csub(m, q);
    }
};

template <>
struct Dft<4> {
    static __device__ __forceinline__ void run(float2 (&v)[4]) {
        const float2 a = cadd(v[0], v[2]), b = csub(v[0], v[2]);
        const float2 c = cadd(v[1], v[3]), d = mul_mi(csub(v[1], v[3]));
        v[0] = cadd(a, c);
        v[1] = cadd(b, d);
        v[2] = csub(a, c);
        v[3] = csub(b, d);
    }
};

template <>
struct Dft<5> {
    static __device__ __forceinline__ void run(float2 (&v)[5]) {
        constexpr float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;
        constexpr float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;
        const float2 t1 = cadd(v[1], v[4]), t2 = cadd(v[2], v[3]);
        const float2 t3 = csub(v[1], v[4]), t4 = csub(v[2], v[3]);
        const float2 a1 = make_float2(v[0].x + c1 * t1.x + c2 * t2.x, v[0].y + c1 * t1.y + c2 * t2.y);
        const float2 a2 = make_float2(v[0].x + c2 * t1.x + c1 * t2.x, v[0].y + c2 * t1.y + c1 * t2.y);
        const float2 b1 = mul_mi(make_float2(s1 * t3.x + s2 * t4.x, s1 * t3.y + s2 * t4.y));
        const float2 b2 = mul_mi(make_float2(s2 * t3.x - s1 * t4.x, s2 * t3.y - s1 * t4.y));
        v[0] = cadd(v[0], cadd(t1, t2));
        v[1] = cadd(a1, b1);
        v[4] = csub(a1, b1);
        v[2] = cadd(a2, b2);
        v[3] = csub(a2, b2);
    }
};

// Cooley-Tukey composite R = R1*R2 with compile-time inner twiddles:
// n = R2*n1 + n2, k = k1 + R1*k2.
template <int R1, int R2>
struct DftCT {
    static constexpr int R = R1 * R2;
    static __device__ __forceinline__ void run(float2 (&v)[R]) {
        float2 t[R2][R1];
        static_for<R2>([&](auto n2c) {
            constexpr int n2 = decltype(n2c)::value;
            float2 u[R1];
            static_for<R1>([&](auto n1c) {
                constexpr int n1 = decltype(n1c)::value;
                u[n1] = v[R2 * n1 + n2];
            });
            Dft<R1>::run(u);
            static_for<R1>([&](auto k1c) {
                constexpr int k1 = decltype(k1c)::value;
                constexpr int p = (n2 * k1) % R;
                if constexpr (p == 0) {
                    t[n2][k1] = u[k1];
                } else {
                    constexpr float wr = (float)cx_cos(2.0 * kPi * p / R);
                    constexpr float wi = (float)(-cx_sin(2.0 * kPi * p / R));
                    t[n2][k1] = cmul(u[k1], make_float2(wr, wi));
                }
            });
        });
        static_for<R1>([&](auto k1c) {
            constexpr int k1 = decltype(k1c)::value;
            float2 u[R2];
            static_for<R2>([&](auto n2c) {
                constexpr int n2 = decltype(n2c)::value;
                u[n2] = t[n2][k1];
            });
            Dft<R2>::run(u);
            static_for<R2>([&](auto k2c) {
                constexpr int k2 = decltype(k2c)::value;
                v[k1 + R1 * k2] = u[k2];
            });
        });
    }
};
template <> struct Dft<6> : DftCT<3, 2> {};
template <> struct Dft<8> : DftCT<4, 2> {};
template <> struct Dft<9> : DftCT<3, 3> {};
template <> struct Dft<10> : DftCT<5, 2> {};
template <> struct Dft<15> : DftCT<5, 3> {};
template <> struct Dft<16> : DftCT<4, 4> {};

// ---------------------------------------------------------------------------
// plans: radices per half-length H
// ---------------------------------------------------------------------------
template <int H> struct Plan;
#define MK_PLAN(H_, ...)                                        \
    template <> struct Plan<H_> {                               \
        static constexpr int radix[] = {__VA_ARGS__};           \
        static constexpr int npass = sizeof(radix) / sizeof(int); \
    };
MK_PLAN(8, 8)
MK_PLAN(16, 16)
MK_PLAN(32, 8, 4)
MK_PLAN(45, 9, 5)
MK_PLAN(48, 8, 6)
MK_PLAN(64, 8, 8)
MK_PLAN(90, 10, 9)
MK_PLAN(120, 10, 4, 3)
MK_PLAN(128, 16, 8)
MK_PLAN(180, 10, 6, 3)
MK_PLAN(240, 16, 15)
MK_PLAN(256, 16, 16)
MK_PLAN(360, 10, 6, 6)
MK_PLAN(720, 10, 9, 8)
#undef MK_PLAN

template <int H, int P>
constexpr int plan_ns() {  // product of radices before pass P
    int ns = 1;
    for (int i = 0; i < P; ++i) ns *= Plan<H>::radix[i];
    return ns;
}

constexpr int kThreads = 256;

// One Stockham pass over G rows of H points, in place (read all -> barrier -> write all).
template <int H, int G, int P>
__device__ __forceinline__ void stockham_pass(float2* lds, const float2* __restrict__ tw, int tid) {
    constexpr int R = Plan<H>::radix[P];
    constexpr int Ns = plan_ns<H, P>();
    constexpr int HR = H / R;
    constexpr int HP = H + 1;
    constexpr int NB = G * HR;  // butterflies per workgroup
    constexpr int ITERS = (NB + kThreads - 1) / kThreads;
    constexpr int STEP = H / (Ns * R);
    float2 v[ITERS][R];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int b = tid + it * kThreads;
        if (b < NB) {
            const int row = b / HR, j = b - row * HR;
            const float2* src = lds + row * HP + j;
#pragma unroll
            for (int r = 0; r < R; ++r) v[it][r] = src[r * HR];
            if constexpr (Ns > 1) {
                const int k = j % Ns;
#pragma unroll
                for (int r = 1; r < R; ++r) v[it][r] = cmul(v[it][r], tw[k * r * STEP]);
            }
            Dft<R>::run(v[it]);
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int b = tid + it * kThreads;
        if (b < NB) {
            const int row = b / HR, j = b - row * HR;
            const int k = j % Ns;
            float2* dst = lds + row * HP + (j / Ns) * (Ns * R) + k;
#pragma unroll
            for (int r = 0; r < R; ++r) dst[r * Ns] = v[it][r];
        }
    }
    __syncthreads();
}

template <int H, int G>
__device__ __forceinline__ void stockham_all(float2* lds, const float2* __restrict__ tw, int tid) {
    static_for<Plan<H>::npass>([&](auto pc) { stockham_pass<H, G, decltype(pc)::value>(lds, tw, tid); });
}

__device__ __forceinline__ float2 load_pair(const float* p) { return *reinterpret_cast<const float2*>(p); }
__device__ __forceinline__ float2 load_pair(const __hip_bfloat16* p) {
    const unsigned int u = *reinterpret_cast<const unsigned int*>(p);
    return make_float2(__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u));
}

// ---------------------------------------------------------------------------
// analysis: x[bc][k][n] -> xf[m][k][bc]
// ---------------------------------------------------------------------------
template <int H, int G, typename TIn>
__global__ __launch_bounds__(kThreads) void rfft_kernel(const TIn* __restrict__ x, float2* __restrict__ xf,
                                                        const float2* __restrict__ tw, int BC, int K, int M,
                                                        float scale0, float scale_m, float scale_h, XfLayout xl) {
    constexpr int N = 2 * H, HP = H + 1;
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    const int tid = threadIdx.x;
    const int ntile = (BC + G - 1) / G;
    const int k = blockIdx.x / ntile;
    const int bc0 = (blockIdx.x - k * ntile) * G;

    for (int idx = tid; idx < G * H; idx += kThreads) {
        const int row = idx / H, j = idx - row * H;
        const int bc = bc0 + row;
        float2 z = make_float2(0.f, 0.f);
        if (bc < BC) z = load_pair(x + ((size_t)bc * K + k) * N + 2 * j);
        lds[row * HP + j] = z;
    }
    __syncthreads();
    stockham_all<H, G>(lds, tw, tid);

    const float2* tw2 = tw + H;  // exp(-2 pi i m / N)
    for (int idx = tid; idx < G * M; idx += kThreads) {
        const int m = idx / G, row = idx - m * G;
        const int bc = bc0 + row;
        if (bc >= BC) continue;
        const int i0 = (m == H) ? 0 : m;
        const int i1 = (m == 0 || m == H) ? 0 : H - m;
        const float2 a = lds[row * HP + i0];
        float2 b = lds[row * HP + i1];
        b.y = -b.y;  // conj
        const float2 e = make_float2(0.5f * (a.x + b.x), 0.5f * (a.y + b.y));
        const float2 d = make_float2(0.5f * (a.x - b.x), 0.5f * (a.y - b.y));
        const float2 o = mul_mi(cmul(tw2[m], d));
        const float s = (m == 0) ? scale0 : ((m == H) ? scale_h : scale_m);
        xf[((size_t)m * xl.sm + (size_t)k * xl.sk) * BC + bc] = make_float2(s * (e.x + o.x), s * (e.y + o.y));
    }
}

// ---------------------------------------------------------------------------
// synthesis: xf[m][k][bc] (m < M, zero above) -> x[bc][k][n]
// ---------------------------------------------------------------------------
template <int H, int G>
__global__ __launch_bounds__(kThreads) void irfft_kernel(const float2* __restrict__ xf, float* __restrict__ x,
                                                         const float2* __restrict__ tw, int BC, int K, int M,
                                                         float scale0, float scale_m, float scale_h, XfLayout xl) {
    constexpr int N = 2 * H, HP = H + 1;
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    const int tid = threadIdx.x;
    const int ntile = (BC + G - 1) / G;
    const int k = blockIdx.x / ntile;
    const int bc0 = (blockIdx.x - k * ntile) * G;
    const float2* tw2 = tw + H;

    // Z[j] = (X[j] + conj X[H-j]) + i e^{+2 pi i j/N} (X[j] - conj X[H-j]); store conj(Z)
    for (int idx = tid; idx < G * H; idx += kThreads) {
        const int j = idx / G, row = idx - j * G;
        const int bc = bc0 + row;
        float2 a = make_float2(0.f, 0.f), b = make_float2(0.f, 0.f);
        if (bc < BC) {
            if (j < M) {
                a = xf[((size_t)j * xl.sm + (size_t)k * xl.sk) * BC + bc];
                const float s = (j == 0) ? scale0 : scale_m;
                a.x *= s;
                a.y *= s;
                if (j == 0) a.y = 0.f;  // irfft ignores Im of the zero mode
            }
            const int jm = H - j;  // 1..H
            if (jm < M) {
                b = xf[((size_t)jm * xl.sm + (size_t)k * xl.sk) * BC + bc];
                const float sb = (jm == H) ? scale_h : scale_m;
                b.x *= sb;
                b.y *= sb;
                if (jm == H) b.y = 0.f;  // Nyquist taken real
                b.y = -b.y;              // conj
            }
        }
        const float2 e = cadd(a, b), d = csub(a, b);
        float2 w = tw2[j];
        w.y = -w.y;                                // e^{+2 pi i j/N}
        const float2 t = cmul(w, d);               // w d
        const float2 z = make_float2(e.x - t.y, e.y + t.x);  // e + i t
        lds[row * HP + j] = make_float2(z.x, -z.y);
    }
    __syncthreads();
    stockham_all<H, G>(lds, tw, tid);

    for (int idx = tid; idx < G * H; idx += kThreads) {
        const int row = idx / H, j = idx - row * H;
        const int bc = bc0 + row;
        if (bc >= BC) continue;
        const float2 z = lds[row * HP + j];
        *reinterpret_cast<float2*>(x + ((size_t)bc * K + k) * N + 2 * j) = make_float2(z.x, -z.y);
    }
}

// ---------------------------------------------------------------------------
// generic lengths: direct DFT (correct for any even nlon, not tuned)
// ---------------------------------------------------------------------------
template <typename TIn>
__global__ __launch_bounds__(kThreads) void rdft_generic_kernel(const TIn* __restrict__ x, float2* __restrict__ xf,
                                                                const float2* __restrict__ tw, int BC, int K, int N,
                                                                int M, float scale0, float scale_m, float scale_h, XfLayout xl) {
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    float* row = reinterpret_cast<float*>(lds);  // N floats
    const int H = N / 2;
    const float2* tw2 = tw + H;  // exp(-2 pi i m/N), m <= H
    const int bc = blockIdx.x % BC, k = blockIdx.x / BC;
    for (int n = threadIdx.x; n < N; n += kThreads) {
        if constexpr (std::is_same<TIn, float>::value)
            row[n] = x[((size_t)bc * K + k) * N + n];
        else
            row[n] = __bfloat162float(x[((size_t)bc * K + k) * N + n]);
    }
    __syncthreads();
    for (int m = threadIdx.x; m < M; m += kThreads) {
        float re = 0.f, im = 0.f;
        for (int n = 0; n < N; ++n) {
            int p = (int)(((long long)m * n) % N);
            float2 w = (p <= H) ? tw2[p] : make_float2(tw2[N - p].x, -tw2[N - p].y);
            re += row[n] * w.x;
            im += row[n] * w.y;
        }
        const float s = (m == 0) ? scale0 : ((m == H) ? scale_h : scale_m);
        xf[((size_t)m * xl.sm + (size_t)k * xl.sk) * BC + bc] = make_float2(s * re, s * im);
    }
}

__global__ __launch_bounds__(kThreads) void irdft_generic_kernel(const float2* __restrict__ xf, float* __restrict__ x,
                                                                 const float2* __restrict__ tw, int BC, int K, int N,
                                                                 int M, float scale0, float scale_m, float scale_h, XfLayout xl) {
    extern __shared__ __attribute__((aligned(16))) float2 lds[];  // M modes
    const int H = N / 2;
    const float2* tw2 = tw + H;
    const int bc = blockIdx.x % BC, k = blockIdx.x / BC;
    for (int m = threadIdx.x; m < M; m += kThreads) {
        float2 a = xf[((size_t)m * xl.sm + (size_t)k * xl.sk) * BC + bc];
        const float s = (m == 0) ? scale0 : ((m == H) ? scale_h : 2.f * scale_m);
        a.x *= s;
        a.y *= s;
        if (m == 0 || m == H) a.y = 0.f;
        lds[m] = a;
    }
    __syncthreads();
    for (int n = threadIdx.x; n < N; n += kThreads) {
        float acc = 0.f;
        for (int m = 0; m < M; ++m) {
            int p = (int)(((long long)m * n) % N);
            float2 w = (p <= H) ? tw2[p] : make_float2(tw2[N - p].x, -tw2[N - p].y);
            // Re(X e^{+i a}) with w = e^{-i a}
            acc += lds[m].x * w.x + lds[m].y * w.y;
        }
        x[((size_t)bc * K + k) * N + n] = acc;
    }
}

template <int H, int G>
int launch_rfft(const void* x, int x_dtype, float* xf, const float* tw, int bc, int nlat, int mmax, float s0,
                float sm, float sh, hipStream_t st) {
    const int ntile = mk::ceil_div(bc, G);
    const dim3 grid((unsigned)(ntile * nlat));
    const size_t lds = sizeof(float2) * G * (H + 1);
    if (x_dtype == 0)
        hipLaunchKernelGGL((rfft_kernel<H, G, float>), grid, dim3(kThreads), lds, st, (const float*)x, (float2*)xf,
                           (const float2*)tw, bc, nlat, mmax, s0, sm, sh, g_xl);
    else
        hipLaunchKernelGGL((rfft_kernel<H, G, __hip_bfloat16>), grid, dim3(kThreads), lds, st,
                           (const __hip_bfloat16*)x, (float2*)xf, (const float2*)tw, bc, nlat, mmax, s0, sm, sh, g_xl);
    return 0;
}

template <int H, int G>
int launch_irfft(const float* xf, float* x, const float* tw, int bc, int nlat, int mmax, float s0, float sm,
                 float sh, hipStream_t st) {
    const int ntile = mk::ceil_div(bc, G);
    const dim3 grid((unsigned)(ntile * nlat));
    const size_t lds = sizeof(float2) * G * (H + 1);
    hipLaunchKernelGGL((irfft_kernel<H, G>), grid, dim3(kThreads), lds, st, (const float2*)xf, x, (const float2*)tw,
                       bc, nlat, mmax, s0, sm, sh, g_xl);
    return 0;
}

#include "fft_split.h"

}  // namespace

#define MK_FFT_SIZES(X) \
    X(8, 16) X(16, 16) X(32, 16) X(45, 16) X(48, 16) X(64, 16) X(90, 16) X(120, 16) X(128, 16) X(180, 16) \
    X(240, 16) X(256, 16) X(360, 8) X(720, 8)

extern "C" int mk_rfft(const void* x, int x_dtype, float* xf, const float* twiddles, int bc, int nlat, int nlon,
                       int mmax, float scale0, float scale_m, float scale_h, void* stream) {
    return mk_rfft_ex(x, x_dtype, xf, twiddles, bc, nlat, nlon, mmax, scale0, scale_m, scale_h, 0, stream);
}

extern "C" int mk_rfft_ex(const void* x, int x_dtype, float* xf, const float* twiddles, int bc, int nlat, int nlon,
                          int mmax, float scale0, float scale_m, float scale_h, int xf_layout, void* stream) {
    MK_REQUIRE(xf_layout == 0 || xf_layout == 1, "xf_layout must be 0 ([M][K][BC]) or 1 ([K][M][BC])");
    g_xl = xf_layout ? XfLayout{1, mmax, 0, 0, 0, 0} : XfLayout{nlat, 1, 0, 0, 0, 0};
    MK_REQUIRE(x && xf && twiddles, "null pointer");
    MK_REQUIRE(bc > 0 && nlat > 0 && nlon >= 2 && nlon % 2 == 0, "bad sizes (nlon must be even)");
    MK_REQUIRE(mmax >= 1 && mmax <= nlon / 2 + 1, "mmax out of range");
    MK_REQUIRE(x_dtype == 0 || x_dtype == 1, "x_dtype must be 0 (fp32) or 1 (bf16)");
    MK_REQUIRE((long long)nlat * mk::ceil_div(bc, 8) < 2147483647LL, "grid too large");
    hipStream_t st = (hipStream_t)stream;
    const bool split = !fft_legacy() && mmax <= 241 && (nlon == 480 || nlon == 1440);
    if (split && nlon == 480) {
        launch_rfft_split<1>(x, x_dtype, xf, twiddles, bc, nlat, mmax, scale0, scale_m, scale_h, st);
    } else if (split) {
        launch_rfft_split<3>(x, x_dtype, xf, twiddles, bc, nlat, mmax, scale0, scale_m, scale_h, st);
    } else
    switch (nlon / 2) {
#define X(H, G) \
    case H:     \
        launch_rfft<H, G>(x, x_dtype, xf, twiddles, bc, nlat, mmax, scale0, scale_m, scale_h, st); \
        break;
        MK_FFT_SIZES(X)
#undef X
        default: {
            MK_REQUIRE((long long)nlat * bc < 2147483647LL, "grid too large");
            const dim3 grid((unsigned)(nlat * bc));
            const size_t lds = sizeof(float) * nlon;
            if (x_dtype == 0)
                hipLaunchKernelGGL((rdft_generic_kernel<float>), grid, dim3(kThreads), lds, st, (const float*)x,
                                   (float2*)xf, (const float2*)twiddles, bc, nlat, nlon, mmax, scale0, scale_m, scale_h, g_xl);
            else
                hipLaunchKernelGGL((rdft_generic_kernel<__hip_bfloat16>), grid, dim3(kThreads), lds, st,
                                   (const __hip_bfloat16*)x, (float2*)xf, (const float2*)twiddles, bc, nlat, nlon,
                                   mmax, scale0, scale_m, scale_h, g_xl);
        }
    }
    MK_LAUNCH_CHECK();
    return 0;
}

extern "C" int mk_irfft(const float* xf, void* xout, int x_dtype, const float* twiddles, int bc, int nlat, int nlon,
                        int mmax, float scale0, float scale_m, float scale_h, void* stream) {
    return mk_irfft_ex(xf, xout, x_dtype, twiddles, bc, nlat, nlon, mmax, scale0, scale_m, scale_h, 0, stream);
}

extern "C" int mk_irfft_ex(const float* xf, void* xout, int x_dtype, const float* twiddles, int bc, int nlat, int nlon,
                           int mmax, float scale0, float scale_m, float scale_h, int xf_layout, void* stream) {
    MK_REQUIRE(xf_layout == 0 || xf_layout == 1, "xf_layout must be 0 ([M][K][BC]) or 1 ([K][M][BC])");
    g_xl = xf_layout ? XfLayout{1, mmax, 0, 0, 0, 0} : XfLayout{nlat, 1, 0, 0, 0, 0};
    float* x = (float*)xout;
    MK_REQUIRE(x_dtype == 0 || x_dtype == 1, "x_dtype must be 0 (fp32) or 1 (bf16)");
    MK_REQUIRE(x_dtype == 0 || (!fft_legacy() && mmax <= 241 && (nlon == 480 || nlon == 1440)),
               "bf16 output rows are built for the production lengths only (nlon 480 / 1440, mmax <= 241)");
    MK_REQUIRE(x && xf && twiddles, "null pointer");
    MK_REQUIRE(bc > 0 && nlat > 0 && nlon >= 2 && nlon % 2 == 0, "bad sizes (nlon must be even)");
    MK_REQUIRE(mmax >= 1 && mmax <= nlon / 2 + 1, "mmax out of range");
    MK_REQUIRE((long long)nlat * mk::ceil_div(bc, 8) < 2147483647LL, "grid too large");
    hipStream_t st = (hipStream_t)stream;
    const bool split = !fft_legacy() && mmax <= 241 && (nlon == 480 || nlon == 1440);
    if (split && nlon == 480) {
        launch_irfft_split<1>(xf, xout, x_dtype, twiddles, bc, nlat, mmax, scale0, scale_m, scale_h, st);
    } else if (split) {
        launch_irfft_split<3>(xf, xout, x_dtype, twiddles, bc, nlat, mmax, scale0, scale_m, scale_h, st);
    } else
    switch (nlon / 2) {
#define X(H, G) \
    case H:     \
        launch_irfft<H, G>(xf, x, twiddles, bc, nlat, mmax, scale0, scale_m, scale_h, st); \
        break;
        MK_FFT_SIZES(X)
#undef X
        default: {
            MK_REQUIRE((long long)nlat * bc < 2147483647LL, "grid too large");
            const dim3 grid((unsigned)(nlat * bc));
            const size_t lds = sizeof(float2) * mmax;
            hipLaunchKernelGGL(irdft_generic_kernel, grid, dim3(kThreads), lds, st, (const float2*)xf, x,
                               (const float2*)twiddles, bc, nlat, nlon, mmax, scale0, scale_m, scale_h, g_xl);
        }
    }
    MK_LAUNCH_CHECK();
    return 0;
}

// Peer-major Fourier rows (see XfLayout): bc = batch * chans rows, channel blocks of chans_per_peer.
static int fft_pm_setup(int bc, int nlat, int nlon, int mmax, int chans, int chans_per_peer) {
    MK_REQUIRE(chans > 0 && chans_per_peer > 0 && bc % chans == 0 && chans % chans_per_peer == 0, "bad channel blocking");
    MK_REQUIRE(chans_per_peer % 24 == 0, "channel blocks must be multiples of 24 (tiles of 8 / 24 rows must not straddle them)");
    MK_REQUIRE(!fft_legacy() && mmax <= 241 && (nlon == 480 || nlon == 1440), "peer-major rows exist for the split kernels only");
    const int Bn = bc / chans;
    g_xl = XfLayout{1, mmax, chans, chans_per_peer, Bn, (long long)nlat * mmax * Bn * chans_per_peer};
    return 0;
}

extern "C" int mk_rfft_pm(const void* x, int x_dtype, float* xf, const float* twiddles, int bc, int nlat, int nlon, int mmax,
                          float scale0, float scale_m, float scale_h, int chans, int chans_per_peer, void* stream) {
    MK_REQUIRE(x && xf && twiddles, "null pointer");
    MK_REQUIRE(bc > 0 && nlat > 0 && mmax >= 1, "bad sizes");
    MK_REQUIRE(x_dtype == 0 || x_dtype == 1, "x_dtype must be 0 (fp32) or 1 (bf16)");
    if (int e = fft_pm_setup(bc, nlat, nlon, mmax, chans, chans_per_peer)) return e;
    if (nlon == 480)
        launch_rfft_split<1>(x, x_dtype, xf, twiddles, bc, nlat, mmax, scale0, scale_m, scale_h, (hipStream_t)stream);
    else
        launch_rfft_split<3>(x, x_dtype, xf, twiddles, bc, nlat, mmax, scale0, scale_m, scale_h, (hipStream_t)stream);
    MK_LAUNCH_CHECK();
    return 0;
}

extern "C" int mk_irfft_pm(const float* xf, void* x, int x_dtype, const float* twiddles, int bc, int nlat, int nlon, int mmax,
                           float scale0, float scale_m, float scale_h, int chans, int chans_per_peer, void* stream) {
    MK_REQUIRE(x && xf && twiddles, "null pointer");
    MK_REQUIRE(bc > 0 && nlat > 0 && mmax >= 1, "bad sizes");
    MK_REQUIRE(x_dtype == 0 || x_dtype == 1, "x_dtype must be 0 (fp32) or 1 (bf16)");
    if (int e = fft_pm_setup(bc, nlat, nlon, mmax, chans, chans_per_peer)) return e;
    if (nlon == 480)
        launch_irfft_split<1>(xf, x, x_dtype, twiddles, bc, nlat, mmax, scale0, scale_m, scale_h, (hipStream_t)stream);
    else
        launch_irfft_split<3>(xf, x, x_dtype, twiddles, bc, nlat, mmax, scale0, scale_m, scale_h, (hipStream_t)stream);
    MK_LAUNCH_CHECK();
    return 0;
}

// Inverse transform that also delivers the statistics of its output: rowsums[2 r], rowsums[2 r + 1] += sum / sum of squares of
// output row r = b * C + c over this call's latitudes and all longitudes (fp64 accumulators the caller has zeroed), taken on
// the values as stored -- the first pass of the instance norm that follows the inverse SHT in every block (sfnonet.py:239-253)
// without reading the field again.  Split kernels only (nlon 480 / 1440, mmax <= 241); chans_per_peer > 0 selects the peer-major
// layout of mk_irfft_pm, else xf_layout as in mk_irfft_ex.
extern "C" int mk_irfft_sums(const float* xf, void* x, int x_dtype, const float* twiddles, int bc, int nlat, int nlon, int mmax,
                             float scale0, float scale_m, float scale_h, int xf_layout, int chans, int chans_per_peer,
                             double* rowsums, void* stream) {
    MK_REQUIRE(rowsums != nullptr, "null pointer");
    MK_REQUIRE(!fft_legacy() && mmax <= 241 && (nlon == 480 || nlon == 1440), "row statistics come from the split kernels only");
    g_rowsums = rowsums;
    const int rc = chans_per_peer > 0
        ? mk_irfft_pm(xf, x, x_dtype, twiddles, bc, nlat, nlon, mmax, scale0, scale_m, scale_h, chans, chans_per_peer, stream)
        : mk_irfft_ex(xf, x, x_dtype, twiddles, bc, nlat, nlon, mmax, scale0, scale_m, scale_h, xf_layout, stream);
    g_rowsums = nullptr;
    return rc;
}
