// Split real FFT for the two production longitudes, N = 480 * S (S = 1: 480, S = 3: 1440).
// Included inside the anonymous namespace of fft.hip (uses Dft<>, cmul, ...).
//
// A real row of N = 480*S points is S interleaved sub-sequences x[S*n + s] of 480 reals.  Each is
// transformed as a 240-point complex FFT (480 reals packed in pairs) with the radix plan 16 x 15,
// and because only modes m <= 240 are kept the radix-S decimation-in-time combine
//     X[m] = sum_s W_N^{s m} F_s[m],   m <= 240
// needs exactly the 241 non-redundant modes of every F_s: nothing is computed that is thrown away.
//
// Workgroup = 384 threads = 24 sub-rows x 16 lanes (S = 3: 8 rows; S = 1: 24 rows).  A sub-row is
// owned by 16 lanes of ONE wave, pass A (radix 16) and pass B (radix 15, in place) run back to back
// with no workgroup barrier: LDS operations of a wave execute in program order.  Only the
// global <-> LDS staging and the mode split / combine cross waves (2 barriers per tile).
//
// LDS image: sub-row sr at float2 offset sr*272 + 4*(sr/S); pass A output index i is stored at
// i + (i >> 4) (16-element blocks padded by one) -> conflict-free radix-16 scatter and radix-15
// gathers; the 4*(row) rotation makes the 8-rows-per-mode accesses of the split step conflict-free.

constexpr int SH = 240;        // complex points per sub-FFT
constexpr int SP = 272;        // sub-row stride (float2), = 16 mod 32
constexpr int SNSUB = 24;      // sub-rows per workgroup
constexpr int STHREADS = 384;  // 16 lanes per sub-row
constexpr int SLDS_F2 = SNSUB * SP + 96;

__device__ __forceinline__ int sub_base(int sr, int S) { return sr * SP + 4 * (sr / S); }
__device__ __forceinline__ int phi(int i) { return i + (i >> 4); }

// radix-16 then radix-15 on the sub-row owned by this lane group; tw15[r] = exp(-2 pi i k r / 240)
__device__ __forceinline__ void split_passes(float2* lds, int tid, int S, const float2 (&tw15)[15]) {
    const int sr = tid >> 4, j = tid & 15;
    float2* base = lds + sub_base(sr, S);
    if (j < 15) {
        float2 v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = base[j + 15 * r];
        Dft<16>::run(v);
#pragma unroll
        for (int r = 0; r < 16; ++r) base[17 * j + r] = v[r];
    }
    // no barrier: the whole sub-row lives in this wave, DS ops of a wave are ordered
    float2 u[15];
#pragma unroll
    for (int r = 0; r < 15; ++r) u[r] = base[j + 17 * r];
#pragma unroll
    for (int r = 1; r < 15; ++r) u[r] = cmul(u[r], tw15[r]);
    Dft<15>::run(u);
#pragma unroll
    for (int r = 0; r < 15; ++r) base[j + 17 * r] = u[r];
}

template <typename T> struct InVec;
template <> struct InVec<float> {
    static constexpr int E = 4;
    static __device__ __forceinline__ void load(const float* p, float (&o)[4]) {
        const float4 v = *reinterpret_cast<const float4*>(p);
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    }
};
template <> struct InVec<__hip_bfloat16> {
    static constexpr int E = 8;
    static __device__ __forceinline__ void load(const __hip_bfloat16* p, float (&o)[8]) {
        const uint4 v = *reinterpret_cast<const uint4*>(p);
        const unsigned int w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            o[2 * i] = __uint_as_float(w[i] << 16);
            o[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    }
};

template <int S, typename TIn>
__global__ __launch_bounds__(STHREADS) void rfft_split_kernel(const TIn* __restrict__ x, float2* __restrict__ xf,
                                                              const float2* __restrict__ tw, int BC, int K, int M,
                                                              float scale0, float scale_m, float scale_h) {
    constexpr int N = 480 * S, G = SNSUB / S, HH = N / 2, E = InVec<TIn>::E;
    constexpr int VPR = N / E, NV = G * VPR, IT = (NV + STHREADS - 1) / STHREADS;
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    const int tid = threadIdx.x;
    const int ntile = (BC + G - 1) / G;
    const int k = blockIdx.x / ntile;
    const int bc0 = (blockIdx.x - k * ntile) * G;

    float vals[IT][E];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int v = tid + it * STHREADS;
        const int g = v / VPR, q = v - g * VPR;
        const bool ok = (v < NV) && (bc0 + g < BC);
        if (ok) {
            InVec<TIn>::load(x + ((size_t)(bc0 + g) * K + k) * N + q * E, vals[it]);
        } else {
#pragma unroll
            for (int e = 0; e < E; ++e) vals[it][e] = 0.f;
        }
    }
    float2 tw15[15];
#pragma unroll
    for (int r = 0; r < 15; ++r) tw15[r] = tw[((tid & 15) * r) * S];

    float* ldsf = reinterpret_cast<float*>(lds);
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int v = tid + it * STHREADS;
        if (v < NV) {
            const int g = v / VPR, q = v - g * VPR;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int i = q * E + e;          // real index in the row: i = S*n + s, n = 2*n2 + part
                const int n = i / S, s = i - n * S;
                ldsf[2 * (sub_base(g * S + s, S) + (n >> 1)) + (n & 1)] = vals[it][e];
            }
        }
    }
    __syncthreads();
    split_passes(lds, tid, S, tw15);
    __syncthreads();

    const float2* tw2 = tw + HH;  // exp(-2 pi i m / N)
    for (int idx = tid; idx < G * M; idx += STHREADS) {
        const int m = idx / G, g = idx - m * G;
        const int bc = bc0 + g;
        if (bc >= BC) continue;
        const int i0 = (m == SH) ? 0 : m;
        const int i1 = (m == 0 || m == SH) ? 0 : SH - m;
        const int p0 = phi(i0), p1 = phi(i1);
        const float2 wsub = tw2[S * m];  // exp(-2 pi i m / 480)
        float2 acc = make_float2(0.f, 0.f);
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const float2* zs = lds + sub_base(g * S + s, S);
            const float2 a = zs[p0];
            float2 b = zs[p1];
            b.y = -b.y;
            const float2 e = make_float2(0.5f * (a.x + b.x), 0.5f * (a.y + b.y));
            const float2 d = make_float2(0.5f * (a.x - b.x), 0.5f * (a.y - b.y));
            float2 f = cadd(e, mul_mi(cmul(wsub, d)));
            if (s > 0) f = cmul(f, tw2[s * m]);
            acc = cadd(acc, f);
        }
        const float sc = (m == 0) ? scale0 : ((m == HH) ? scale_h : scale_m);
        xf[((size_t)m * K + k) * BC + bc] = make_float2(sc * acc.x, sc * acc.y);
    }
}

template <int S>
__global__ __launch_bounds__(STHREADS) void irfft_split_kernel(const float2* __restrict__ xf, float* __restrict__ x,
                                                               const float2* __restrict__ tw, int BC, int K, int M,
                                                               float scale0, float scale_m, float scale_h) {
    constexpr int N = 480 * S, G = SNSUB / S, HH = N / 2;
    constexpr int VPR = N / 4, NV = G * VPR, IT = (NV + STHREADS - 1) / STHREADS;
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    const int tid = threadIdx.x;
    const int ntile = (BC + G - 1) / G;
    const int k = blockIdx.x / ntile;
    const int bc0 = (blockIdx.x - k * ntile) * G;
    const float2* tw2 = tw + HH;

    float2 tw15[15];
#pragma unroll
    for (int r = 0; r < 15; ++r) tw15[r] = tw[((tid & 15) * r) * S];

    for (int idx = tid; idx < G * SH; idx += STHREADS) {
        const int j = idx / G, g = idx - j * G;
        const int bc = bc0 + g;
        const int jm = SH - j;  // 1 .. 240
        float2 xa = make_float2(0.f, 0.f), xb = make_float2(0.f, 0.f);
        if (bc < BC) {
            if (j < M) {
                xa = xf[((size_t)j * K + k) * BC + bc];
                const float sc = (j == 0) ? scale0 : scale_m;
                xa.x *= sc;
                xa.y *= sc;
            }
            if (jm < M) {
                xb = xf[((size_t)jm * K + k) * BC + bc];
                const float sc = (jm == HH) ? scale_h : scale_m;
                xb.x *= sc;
                xb.y *= sc;
            }
        }
        float2 wsub = tw2[S * j];
        wsub.y = -wsub.y;  // exp(+2 pi i j / 480)
#pragma unroll
        for (int s = 0; s < S; ++s) {
            float2 ya = xa, yb = xb;
            if (s > 0) {  // Y_s[m] = X[m] W_N^{-s m}
                float2 wa = tw2[s * j], wb = tw2[s * jm];
                wa.y = -wa.y;
                wb.y = -wb.y;
                ya = cmul(ya, wa);
                yb = cmul(yb, wb);
            }
            if (j == 0) ya.y = 0.f;             // imaginary part of the zero mode is ignored
            if (jm == SH) {                     // Nyquist of the 480-point sub-transform
                yb.y = 0.f;
                if (S > 1) yb.x *= 2.f;         // interior mode of the long transform: both conjugate halves
            }
            yb.y = -yb.y;                       // conj
            const float2 e = cadd(ya, yb), d = csub(ya, yb);
            const float2 t = cmul(wsub, d);
            // z = e + i t, stored conjugated (inverse transform = conj . forward . conj)
            lds[sub_base(g * S + s, S) + j] = make_float2(e.x - t.y, -(e.y + t.x));
        }
    }
    __syncthreads();
    split_passes(lds, tid, S, tw15);
    __syncthreads();

    const float* ldsf = reinterpret_cast<const float*>(lds);
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int v = tid + it * STHREADS;
        const int g = v / VPR, q = v - g * VPR;
        if (v < NV && bc0 + g < BC) {
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = q * 4 + e;
                const int n = i / S, s = i - n * S;
                const float val = ldsf[2 * (sub_base(g * S + s, S) + phi(n >> 1)) + (n & 1)];
                o[e] = (n & 1) ? -val : val;
            }
            *reinterpret_cast<float4*>(x + ((size_t)(bc0 + g) * K + k) * N + q * 4) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
}

template <int S>
int launch_rfft_split(const void* x, int x_dtype, float* xf, const float* tw, int bc, int nlat, int mmax, float s0,
                      float sm, float sh, hipStream_t st) {
    constexpr int G = SNSUB / S;
    const dim3 grid((unsigned)(mk::ceil_div(bc, G) * nlat));
    const size_t lds = sizeof(float2) * SLDS_F2;
    if (x_dtype == 0)
        hipLaunchKernelGGL((rfft_split_kernel<S, float>), grid, dim3(STHREADS), lds, st, (const float*)x, (float2*)xf,
                           (const float2*)tw, bc, nlat, mmax, s0, sm, sh);
    else
        hipLaunchKernelGGL((rfft_split_kernel<S, __hip_bfloat16>), grid, dim3(STHREADS), lds, st,
                           (const __hip_bfloat16*)x, (float2*)xf, (const float2*)tw, bc, nlat, mmax, s0, sm, sh);
    return 0;
}

template <int S>
int launch_irfft_split(const float* xf, float* x, const float* tw, int bc, int nlat, int mmax, float s0, float sm,
                       float sh, hipStream_t st) {
    constexpr int G = SNSUB / S;
    const dim3 grid((unsigned)(mk::ceil_div(bc, G) * nlat));
    const size_t lds = sizeof(float2) * SLDS_F2;
    hipLaunchKernelGGL((irfft_split_kernel<S>), grid, dim3(STHREADS), lds, st, (const float2*)xf, x, (const float2*)tw,
                       bc, nlat, mmax, s0, sm, sh);
    return 0;
}

static inline bool fft_legacy() {
    static const bool v = [] {
        const char* e = getenv("MK_FFT_LEGACY");
        return e && e[0] == '1';
    }();
    return v;
}
