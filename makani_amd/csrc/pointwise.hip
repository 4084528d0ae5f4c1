// Fused pointwise kernels of the FNO block for gfx950: bias + GELU and
// instance normalisation (+ optional GELU), forward and backward, on NCHW fields viewed as
// [rows = B*C][P = H*W].  All are HBM streaming kernels: 16-byte accesses, fp32 arithmetic,
// bf16 or fp32 storage; one pass for bias+GELU, two passes (statistics, apply) for the norm.
//
// Replaces in the reference block: `nn.Conv2d` bias add + `nn.GELU` (layers.py:95-99,158-206),
// `nn.InstanceNorm2d(eps=1e-6, affine=True)` + `act_layer0` (sfnonet.py:239-253, 375-380).
#include "common.h"
#include "../../include/makani_amd.h"

#include <hip/hip_bf16.h>

namespace {

constexpr int kT = 256;          // threads per workgroup
constexpr int kE = 8;            // elements per thread per step
constexpr int kSteps = 4;        // steps per workgroup -> 8192 elements per workgroup
constexpr int kChunk = kT * kE * kSteps;

template <typename T> struct IO;
template <> struct IO<float> {
    static __device__ __forceinline__ void load(const float* p, float (&v)[kE]) {
        const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
    static __device__ __forceinline__ void store(float* p, const float (&v)[kE]) {
        reinterpret_cast<float4*>(p)[0] = make_float4(v[0], v[1], v[2], v[3]);
        reinterpret_cast<float4*>(p)[1] = make_float4(v[4], v[5], v[6], v[7]);
    }
    static __device__ __forceinline__ float ld1(const float* p) { return *p; }
    static __device__ __forceinline__ void st1(float* p, float v) { *p = v; }
};
template <> struct IO<__hip_bfloat16> {
    static __device__ __forceinline__ void load(const __hip_bfloat16* p, float (&v)[kE]) {
        const uint4 u = *reinterpret_cast<const uint4*>(p);
        const unsigned int w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = __uint_as_float(w[i] << 16);
            v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    }
    static __device__ __forceinline__ void store(__hip_bfloat16* p, const float (&v)[kE]) {
        __hip_bfloat16 h[kE];
#pragma unroll
        for (int i = 0; i < kE; ++i) h[i] = __float2bfloat16(v[i]);  // round to nearest even, NaN safe
        *reinterpret_cast<uint4*>(p) = *reinterpret_cast<const uint4*>(h);
    }
    static __device__ __forceinline__ float ld1(const __hip_bfloat16* p) { return __bfloat162float(*p); }
    static __device__ __forceinline__ void st1(__hip_bfloat16* p, float v) { *p = __float2bfloat16(v); }
};

// GELU (exact erf form, nn.GELU()) and its derivative.  fp32 fields: erff().  bf16 fields: the normal CDF through
// erfc(|z|/sqrt2) in the Abramowitz-Stegun 7.1.26 rational-exponential form (|error| < 1.5e-7 absolute on erf, three orders
// below the 2^-9 rounding of the stored value): 2 transcendentals + ~12 FMAs instead of erff()'s ~35 instructions plus a
// separate exp for the derivative -- with erff() the norm+GELU passes were VALU-bound (the backward sums pass ran at
// 3.3 TB/s against 5.4 TB/s for the same pass without the activation).
struct PhiPair {
    float Phi, phi;   // standard normal CDF and PDF
};
__device__ __forceinline__ PhiPair normal_cdf_pdf_fast(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float q = fmaf(1.061405429f, t, -1.453152027f);
    q = fmaf(q, t, 1.421413741f);
    q = fmaf(q, t, -0.284496736f);
    q = fmaf(q, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);   // exp(-x^2 / 2)
    const float half_erfc = 0.5f * q * t * e;                               // Phi(-|x|)
    PhiPair r;
    r.Phi = x < 0.f ? half_erfc : 1.0f - half_erfc;
    r.phi = 0.3989422804014327f * e;
    return r;
}
template <typename T> struct Act;
template <> struct Act<float> {
    static __device__ __forceinline__ float gelu(float z) { return 0.5f * z * (1.f + erff(z * 0.70710678118654752440f)); }
    static __device__ __forceinline__ float gelu_grad(float z) {
        return 0.5f * (1.f + erff(z * 0.70710678118654752440f)) + z * 0.39894228040143267794f * __expf(-0.5f * z * z);
    }
};
template <> struct Act<__hip_bfloat16> {
    static __device__ __forceinline__ float gelu(float z) { return z * normal_cdf_pdf_fast(z).Phi; }
    static __device__ __forceinline__ float gelu_grad(float z) {
        const PhiPair c = normal_cdf_pdf_fast(z);
        return fmaf(z, c.phi, c.Phi);
    }
};

// visit the elements [p0, p1) of one row in vectors of kE (plus a scalar tail); F(offset, float(&)[kE], n)
template <typename T, class F>
__device__ __forceinline__ void for_chunk(long long P, F&& f) {
    const long long c0 = (long long)blockIdx.x * kChunk;
#pragma unroll
    for (int s = 0; s < kSteps; ++s) {
        const long long off = c0 + ((long long)s * kT + threadIdx.x) * kE;
        if (off + kE <= P)
            f(off, kE);
        else if (off < P)
            f(off, (int)(P - off));
    }
}

__device__ __forceinline__ float block_sum(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// ------------------------------------------------------------------ bias + GELU
template <typename T>
__global__ __launch_bounds__(kT) void bias_gelu_fwd_kernel(const T* __restrict__ x, const float* __restrict__ bias,
                                                           T* __restrict__ y, int C, long long P) {
    const int row = blockIdx.y;
    const float b = bias ? bias[row % C] : 0.f;
    const T* xr = x + (long long)row * P;
    T* yr = y + (long long)row * P;
    for_chunk<T>(P, [&](long long off, int n) {
        float v[kE];
        if (n == kE) {
            IO<T>::load(xr + off, v);
#pragma unroll
            for (int i = 0; i < kE; ++i) v[i] = Act<T>::gelu(v[i] + b);
            IO<T>::store(yr + off, v);
        } else {
            for (int i = 0; i < n; ++i) IO<T>::st1(yr + off + i, Act<T>::gelu(IO<T>::ld1(xr + off + i) + b));
        }
    });
}

template <typename T>
__global__ __launch_bounds__(kT) void bias_gelu_bwd_kernel(const T* __restrict__ x, const float* __restrict__ bias,
                                                           const T* __restrict__ gy, T* __restrict__ gx,
                                                           float* __restrict__ gbias, int C, long long P) {
    __shared__ float red[4];
    const int row = blockIdx.y;
    const float b = bias ? bias[row % C] : 0.f;
    const long long ro = (long long)row * P;
    float acc = 0.f;
    for_chunk<T>(P, [&](long long off, int n) {
        float v[kE], g[kE];
        if (n == kE) {
            IO<T>::load(x + ro + off, v);
            IO<T>::load(gy + ro + off, g);
#pragma unroll
            for (int i = 0; i < kE; ++i) {
                g[i] *= Act<T>::gelu_grad(v[i] + b);
                acc += g[i];
            }
            IO<T>::store(gx + ro + off, g);
        } else {
            for (int i = 0; i < n; ++i) {
                const float r = IO<T>::ld1(gy + ro + off + i) * Act<T>::gelu_grad(IO<T>::ld1(x + ro + off + i) + b);
                acc += r;
                IO<T>::st1(gx + ro + off + i, r);
            }
        }
    });
    if (gbias) {
        const float s = block_sum(acc, red);
        if (threadIdx.x == 0) atomicAdd(&gbias[row % C], s);
    }
}

// ------------------------------------------------------------------ instance norm
// pass 1: per-row sums (sum x, sum x^2) accumulated in double across workgroups
template <typename T>
__global__ __launch_bounds__(kT) void rowsum2_kernel(const T* __restrict__ x, double* __restrict__ sums, long long P) {
    __shared__ float red[4];
    const int row = blockIdx.y;
    const T* xr = x + (long long)row * P;
    float s1 = 0.f, s2 = 0.f;
    for_chunk<T>(P, [&](long long off, int n) {
        float v[kE];
        if (n == kE) {
            IO<T>::load(xr + off, v);
#pragma unroll
            for (int i = 0; i < kE; ++i) {
                s1 += v[i];
                s2 = fmaf(v[i], v[i], s2);
            }
        } else {
            for (int i = 0; i < n; ++i) {
                const float t = IO<T>::ld1(xr + off + i);
                s1 += t;
                s2 = fmaf(t, t, s2);
            }
        }
    });
    const float t1 = block_sum(s1, red);
    const float t2 = block_sum(s2, red);
    if (threadIdx.x == 0) {
        atomicAdd(&sums[2 * row], (double)t1);
        atomicAdd(&sums[2 * row + 1], (double)t2);
    }
}

// pass 2: y = act((x - mean) * rstd * w + b); stats[row] = (mean, rstd) kept for backward
template <typename T, bool GELU>
__global__ __launch_bounds__(kT) void instnorm_apply_kernel(const T* __restrict__ x, const double* __restrict__ sums,
                                                            const float* __restrict__ w, const float* __restrict__ b,
                                                            T* __restrict__ y, float* __restrict__ stats, int C,
                                                            long long P, double count, float eps) {
    // count = number of elements the sums run over: P, or the global H*W when the sums were reduced over ranks
    const int row = blockIdx.y, c = row % C;
    const double mean_d = sums[2 * row] / count;
    double var_d = sums[2 * row + 1] / count - mean_d * mean_d;
    if (var_d < 0.0) var_d = 0.0;
    const float mean = (float)mean_d, rstd = (float)(1.0 / sqrt(var_d + (double)eps));
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        stats[2 * row] = mean;
        stats[2 * row + 1] = rstd;
    }
    const float sc = rstd * (w ? w[c] : 1.f), sh = (b ? b[c] : 0.f) - mean * sc;
    const T* xr = x + (long long)row * P;
    T* yr = y + (long long)row * P;
    for_chunk<T>(P, [&](long long off, int n) {
        float v[kE];
        if (n == kE) {
            IO<T>::load(xr + off, v);
#pragma unroll
            for (int i = 0; i < kE; ++i) {
                const float z = fmaf(v[i], sc, sh);
                v[i] = GELU ? Act<T>::gelu(z) : z;
            }
            IO<T>::store(yr + off, v);
        } else {
            for (int i = 0; i < n; ++i) {
                const float z = fmaf(IO<T>::ld1(xr + off + i), sc, sh);
                IO<T>::st1(yr + off + i, GELU ? Act<T>::gelu(z) : z);
            }
        }
    });
}

// backward pass 1: per-row sums of g' and g' * xhat, g' = gy * act'(z) * 1 (w applied later)
template <typename T, bool GELU>
__global__ __launch_bounds__(kT) void instnorm_bwd_sums_kernel(const T* __restrict__ x, const T* __restrict__ gy,
                                                               const float* __restrict__ stats,
                                                               const float* __restrict__ w, const float* __restrict__ b,
                                                               double* __restrict__ sums, int C, long long P) {
    __shared__ float red[4];
    const int row = blockIdx.y, c = row % C;
    const float mean = stats[2 * row], rstd = stats[2 * row + 1];
    const float wc = w ? w[c] : 1.f, bc = b ? b[c] : 0.f;
    const long long ro = (long long)row * P;
    float s1 = 0.f, s2 = 0.f;
    for_chunk<T>(P, [&](long long off, int n) {
        float v[kE], g[kE];
        if (n == kE) {
            IO<T>::load(x + ro + off, v);
            IO<T>::load(gy + ro + off, g);
        } else {
            for (int i = 0; i < kE; ++i) {
                v[i] = i < n ? IO<T>::ld1(x + ro + off + i) : mean;
                g[i] = i < n ? IO<T>::ld1(gy + ro + off + i) : 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < kE; ++i) {
            const float xh = (v[i] - mean) * rstd;
            float gi = g[i];
            if (GELU) gi *= Act<T>::gelu_grad(fmaf(xh, wc, bc));
            s1 += gi;
            s2 = fmaf(gi, xh, s2);
        }
    });
    const float t1 = block_sum(s1, red);
    const float t2 = block_sum(s2, red);
    if (threadIdx.x == 0) {
        atomicAdd(&sums[2 * row], (double)t1);
        atomicAdd(&sums[2 * row + 1], (double)t2);
    }
}

// backward pass 2: gx = w * rstd * (g' - mean(g') - xhat * mean(g' xhat))
template <typename T, bool GELU>
__global__ __launch_bounds__(kT) void instnorm_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ gy,
                                                                const float* __restrict__ stats,
                                                                const float* __restrict__ w, const float* __restrict__ b,
                                                                const double* __restrict__ sums, T* __restrict__ gx, int C,
                                                                long long P, double count, float* __restrict__ gwb) {
    const int row = blockIdx.y, c = row % C;
    const float mean = stats[2 * row], rstd = stats[2 * row + 1];
    const float wc = w ? w[c] : 1.f, bc = b ? b[c] : 0.f;
    const float m1 = (float)(sums[2 * row] / count), m2 = (float)(sums[2 * row + 1] / count);
    // one sample per parameter row (rows == C): the row sums ARE the affine parameters' gradients -- gwb[0][c] = weight
    // gradient (sum g' xhat), gwb[1][c] = bias gradient (sum g'); saves the caller a copy / cast launch per norm
    if (gwb != nullptr && blockIdx.x == 0 && threadIdx.x == 0) {
        gwb[c] = (float)sums[2 * row + 1];
        gwb[C + c] = (float)sums[2 * row];
    }
    const float k = wc * rstd;
    const long long ro = (long long)row * P;
    for_chunk<T>(P, [&](long long off, int n) {
        float v[kE], g[kE];
        if (n == kE) {
            IO<T>::load(x + ro + off, v);
            IO<T>::load(gy + ro + off, g);
#pragma unroll
            for (int i = 0; i < kE; ++i) {
                const float xh = (v[i] - mean) * rstd;
                float gi = g[i];
                if (GELU) gi *= Act<T>::gelu_grad(fmaf(xh, wc, bc));
                g[i] = k * (gi - m1 - xh * m2);
            }
            IO<T>::store(gx + ro + off, g);
        } else {
            for (int i = 0; i < n; ++i) {
                const float xh = (IO<T>::ld1(x + ro + off + i) - mean) * rstd;
                float gi = IO<T>::ld1(gy + ro + off + i);
                if (GELU) gi *= Act<T>::gelu_grad(fmaf(xh, wc, bc));
                IO<T>::st1(gx + ro + off + i, k * (gi - m1 - xh * m2));
            }
        }
    });
}

inline dim3 pw_grid(long long rows, long long P) { return dim3((unsigned)((P + kChunk - 1) / kChunk), (unsigned)rows); }

// Accumulators are zeroed by a kernel, not hipMemsetAsync: under stream capture the memset node of a small buffer whose
// storage the allocator hands out again inside the same graph was observed to leave stale values on replay
// (tools/graph_bisect.py: a bias gradient summed into such a buffer turned non-finite from the second replay on).
__global__ void zero_doubles_kernel(double* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0.0;
}
static inline void zero_doubles(double* p, int n, hipStream_t st) {
    hipLaunchKernelGGL(zero_doubles_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, n);
}

}  // namespace

#define PW_CHECK()                                                                              \
    MK_REQUIRE(rows > 0 && rows <= 65535 && C > 0 && P > 0, "bad sizes (rows = B*C must be <= 65535)"); \
    MK_REQUIRE(dtype == 0 || dtype == 1, "dtype must be 0 (fp32) or 1 (bf16)");                 \
    MK_REQUIRE((P % 8) == 0, "P = H*W must be a multiple of 8 (16-byte row alignment)")

extern "C" int mk_bias_gelu_fwd(const void* x, const float* bias, void* y, int dtype, int rows, int C, long long P,
                                void* stream) {
    MK_REQUIRE(x && y, "null pointer");
    PW_CHECK();
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0)
        hipLaunchKernelGGL(bias_gelu_fwd_kernel<float>, pw_grid(rows, P), dim3(kT), 0, st, (const float*)x, bias,
                           (float*)y, C, P);
    else
        hipLaunchKernelGGL(bias_gelu_fwd_kernel<__hip_bfloat16>, pw_grid(rows, P), dim3(kT), 0, st,
                           (const __hip_bfloat16*)x, bias, (__hip_bfloat16*)y, C, P);
    MK_LAUNCH_CHECK();
    return 0;
}

extern "C" int mk_bias_gelu_bwd(const void* x, const float* bias, const void* gy, void* gx, float* gbias, int dtype,
                                int rows, int C, long long P, void* stream) {
    MK_REQUIRE(x && gy && gx, "null pointer");
    PW_CHECK();
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0)
        hipLaunchKernelGGL(bias_gelu_bwd_kernel<float>, pw_grid(rows, P), dim3(kT), 0, st, (const float*)x, bias,
                           (const float*)gy, (float*)gx, gbias, C, P);
    else
        hipLaunchKernelGGL(bias_gelu_bwd_kernel<__hip_bfloat16>, pw_grid(rows, P), dim3(kT), 0, st,
                           (const __hip_bfloat16*)x, bias, (const __hip_bfloat16*)gy, (__hip_bfloat16*)gx, gbias, C, P);
    MK_LAUNCH_CHECK();
    return 0;
}

// phase 0: sums + apply (single GPU).  phase 1: local row sums only (into workspace).  phase 2: apply only --
// workspace holds the sums reduced over the ranks that share the rows, count their total element count.
extern "C" int mk_instnorm_fwd_ex(const void* x, const float* weight, const float* bias, void* y, float* stats,
                                  double* workspace, int dtype, int rows, int C, long long P, long long count, float eps,
                                  int fuse_gelu, int phase, void* stream) {
    MK_REQUIRE(x && workspace, "null pointer");
    MK_REQUIRE(phase == 1 || (y && stats), "null pointer");
    MK_REQUIRE(phase >= 0 && phase <= 2 && count > 0, "bad phase / count");
    PW_CHECK();
    hipStream_t st = (hipStream_t)stream;
    if (phase != 2) zero_doubles(workspace, 2 * rows, st);
    const dim3 g = pw_grid(rows, P);
    const double cnt = (double)count;
#define LAUNCH(T)                                                                                              \
    if (phase != 2) hipLaunchKernelGGL(rowsum2_kernel<T>, g, dim3(kT), 0, st, (const T*)x, workspace, P);      \
    if (phase != 1) {                                                                                          \
        if (fuse_gelu)                                                                                         \
            hipLaunchKernelGGL((instnorm_apply_kernel<T, true>), g, dim3(kT), 0, st, (const T*)x, workspace, weight, \
                               bias, (T*)y, stats, C, P, cnt, eps);                                            \
        else                                                                                                   \
            hipLaunchKernelGGL((instnorm_apply_kernel<T, false>), g, dim3(kT), 0, st, (const T*)x, workspace, weight, \
                               bias, (T*)y, stats, C, P, cnt, eps);                                            \
    }
    if (dtype == 0) {
        LAUNCH(float)
    } else {
        LAUNCH(__hip_bfloat16)
    }
#undef LAUNCH
    MK_LAUNCH_CHECK();
    return 0;
}

extern "C" int mk_instnorm_fwd(const void* x, const float* weight, const float* bias, void* y, float* stats,
                               double* workspace, int dtype, int rows, int C, long long P, float eps, int fuse_gelu,
                               void* stream) {
    return mk_instnorm_fwd_ex(x, weight, bias, y, stats, workspace, dtype, rows, C, P, P, eps, fuse_gelu, 0, stream);
}

namespace {
__global__ void instnorm_coeffs_kernel(const double* __restrict__ sums, const float* __restrict__ w, const float* __restrict__ b,
                                       float* __restrict__ stats, float* __restrict__ affine, int rows, int C, double count,
                                       float eps) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= rows) return;
    const int c = row % C;
    const double mean_d = sums[2 * row] / count;            // the arithmetic of instnorm_apply_kernel, once per row
    double var_d = sums[2 * row + 1] / count - mean_d * mean_d;
    if (var_d < 0.0) var_d = 0.0;
    const float mean = (float)mean_d, rstd = (float)(1.0 / sqrt(var_d + (double)eps));
    stats[2 * row] = mean;
    stats[2 * row + 1] = rstd;
    const float sc = rstd * (w ? w[c] : 1.f);
    affine[2 * row] = sc;
    affine[2 * row + 1] = (b ? b[c] : 0.f) - mean * sc;
}
}  // namespace

// The instance norm of a field as a per-row affine map y = affine[row][0] * x + affine[row][1], from the row sums
// (sum x, sum x^2; summed over the ranks that share a row when it is sharded) -- for a consumer that applies it while it
// reads x (mk_pce_gemm_ex, addend_affine).  stats[row] = (mean, rstd) is what mk_instnorm_bwd_ex takes.
extern "C" int mk_instnorm_coeffs(const double* sums, const float* weight, const float* bias, float* stats, float* affine,
                                  int rows, int C, long long count, float eps, void* stream) {
    MK_REQUIRE(sums && stats && affine, "null pointer");
    MK_REQUIRE(rows > 0 && C > 0 && count > 0, "bad sizes");
    hipLaunchKernelGGL(instnorm_coeffs_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, sums,
                       weight, bias, stats, affine, rows, C, (double)count, eps);
    MK_LAUNCH_CHECK();
    return 0;
}

// phases as in mk_instnorm_fwd_ex; the phase-1 sums (sum g', sum g' xhat per row) are also the LOCAL bias / weight
// gradient contributions
static int instnorm_bwd_launch(const void* x, const void* gy, const float* stats, const float* weight,
                               const float* bias, void* gx, double* workspace, int dtype, int rows, int C, long long P,
                               long long count, int fuse_gelu, int phase, float* gwb, void* stream) {
    MK_REQUIRE(x && gy && stats && workspace, "null pointer");
    MK_REQUIRE(gwb == nullptr || (rows == C && phase == 0), "parameter gradients from the kernel: one sample, unsharded rows only");
    MK_REQUIRE(phase == 1 || gx, "null pointer");
    MK_REQUIRE(phase >= 0 && phase <= 2 && count > 0, "bad phase / count");
    PW_CHECK();
    hipStream_t st = (hipStream_t)stream;
    if (phase != 2) zero_doubles(workspace, 2 * rows, st);
    const dim3 g = pw_grid(rows, P);
    const double cnt = (double)count;
#define LAUNCH(T, G)                                                                                               \
    if (phase != 2)                                                                                                \
        hipLaunchKernelGGL((instnorm_bwd_sums_kernel<T, G>), g, dim3(kT), 0, st, (const T*)x, (const T*)gy, stats, weight, \
                           bias, workspace, C, P);                                                                 \
    if (phase != 1)                                                                                                \
        hipLaunchKernelGGL((instnorm_bwd_apply_kernel<T, G>), g, dim3(kT), 0, st, (const T*)x, (const T*)gy, stats, weight, \
                           bias, workspace, (T*)gx, C, P, cnt, gwb);
    if (dtype == 0) {
        if (fuse_gelu) { LAUNCH(float, true) } else { LAUNCH(float, false) }
    } else {
        if (fuse_gelu) { LAUNCH(__hip_bfloat16, true) } else { LAUNCH(__hip_bfloat16, false) }
    }
#undef LAUNCH
    MK_LAUNCH_CHECK();
    return 0;
}

extern "C" int mk_instnorm_bwd_ex(const void* x, const void* gy, const float* stats, const float* weight,
                                  const float* bias, void* gx, double* workspace, int dtype, int rows, int C, long long P,
                                  long long count, int fuse_gelu, int phase, void* stream) {
    return instnorm_bwd_launch(x, gy, stats, weight, bias, gx, workspace, dtype, rows, C, P, count, fuse_gelu, phase, nullptr, stream);
}

// mk_instnorm_bwd that also writes the affine parameters' gradients (fp32 [2][C]: weight row, bias row) -- batch 1 only
extern "C" int mk_instnorm_bwd_wb(const void* x, const void* gy, const float* stats, const float* weight,
                                  const float* bias, void* gx, double* workspace, float* gwb, int dtype, int C, long long P,
                                  int fuse_gelu, void* stream) {
    MK_REQUIRE(gwb != nullptr, "null pointer");
    return instnorm_bwd_launch(x, gy, stats, weight, bias, gx, workspace, dtype, C, C, P, P, fuse_gelu, 0, gwb, stream);
}

extern "C" int mk_instnorm_bwd(const void* x, const void* gy, const float* stats, const float* weight,
                               const float* bias, void* gx, double* workspace, int dtype, int rows, int C, long long P,
                               int fuse_gelu, void* stream) {
    return mk_instnorm_bwd_ex(x, gy, stats, weight, bias, gx, workspace, dtype, rows, C, P, P, fuse_gelu, 0, stream);
}

// ------------------------------------------------------------------ latitude-weighted squared error (training loss)
// loss = scale * sum_{r, w} wrow[r % H] * (pred[r][w] - tar[r][w])^2 over rows r = (b, c, h) of W points: the
// quadrature-weighted MSE of the bench / trainer harness (the reference's losses reduce with the same
// latitude weights, makani/utils/losses.py:149-271).  One streaming pass forward, one backward
//   gpred[r][w] = 2 * scale * gloss * wrow[r % H] * (pred - tar)
// instead of eight elementwise torch kernels over the 73 x 721 x 1440 field.
namespace {

template <typename T, bool BWD>
__global__ __launch_bounds__(kT) void wmse_kernel(const T* __restrict__ pred, const float* __restrict__ tar,
                                                  const float* __restrict__ wrow, double* __restrict__ acc,
                                                  const float* __restrict__ gloss, T* __restrict__ gpred, long long rows,
                                                  int H, int W, float scale) {
    // Forward: a workgroup walks rows blockIdx.x, + gridDim.x, ... and adds ONE double to the loss at the end (one
    // atomic per row -- 52 k of them on one address at 73 x 721 -- serialised the whole pass: 0.64 ms for 455 MB).
    __shared__ double redd[4];
    double tot = 0.0;
    for (long long row = blockIdx.x; row < rows; row += gridDim.x) {
        const float wr = wrow[row % H];
        const T* p = pred + row * W;
        const float* t = tar + row * W;
        float s = 0.f;
        const float k = BWD ? 2.f * scale * gloss[0] * wr : 0.f;
        for (int i = threadIdx.x * kE; i < W; i += kT * kE) {   // W is a multiple of 8
            float v[kE], u[kE];
            IO<T>::load(p + i, v);
            IO<float>::load(t + i, u);
#pragma unroll
            for (int e = 0; e < kE; ++e) {
                const float d = v[e] - u[e];
                if (BWD) v[e] = k * d;
                else s = fmaf(d, d, s);
            }
            if (BWD) IO<T>::store(gpred + row * W + i, v);
        }
        if (!BWD) tot += (double)(s * wr);
    }
    if (!BWD) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) tot += __shfl_down(tot, o, 64);
        if ((threadIdx.x & 63) == 0) redd[threadIdx.x >> 6] = tot;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(acc, (redd[0] + redd[1] + redd[2] + redd[3]) * (double)scale);
    }
}

}  // namespace

extern "C" int mk_wmse_fwd(const void* pred, int dtype, const float* tar, const float* wrow, double* loss, long long rows,
                           int H, int W, float scale, void* stream) {
    MK_REQUIRE(pred && tar && wrow && loss, "null pointer");
    MK_REQUIRE(rows > 0 && rows < 2147483647LL && H > 0 && W > 0 && (W % 8) == 0, "bad sizes (W must be a multiple of 8)");
    MK_REQUIRE(dtype == 0 || dtype == 1, "dtype must be 0 (fp32) or 1 (bf16)");
    hipStream_t st = (hipStream_t)stream;
    zero_doubles(loss, 1, st);
    if (dtype == 0)
        hipLaunchKernelGGL((wmse_kernel<float, false>), dim3((unsigned)(rows < 4096 ? rows : 4096)), dim3(kT), 0, st,
                           (const float*)pred, tar, wrow, loss, (const float*)nullptr, (float*)nullptr, rows, H, W, scale);
    else
        hipLaunchKernelGGL((wmse_kernel<__hip_bfloat16, false>), dim3((unsigned)(rows < 4096 ? rows : 4096)), dim3(kT), 0, st,
                           (const __hip_bfloat16*)pred, tar, wrow, loss, (const float*)nullptr, (__hip_bfloat16*)nullptr, rows,
                           H, W, scale);
    MK_LAUNCH_CHECK();
    return 0;
}

extern "C" int mk_wmse_bwd(const void* pred, int dtype, const float* tar, const float* wrow, const float* gloss, void* gpred,
                           long long rows, int H, int W, float scale, void* stream) {
    MK_REQUIRE(pred && tar && wrow && gloss && gpred, "null pointer");
    MK_REQUIRE(rows > 0 && rows < 2147483647LL && H > 0 && W > 0 && (W % 8) == 0, "bad sizes (W must be a multiple of 8)");
    MK_REQUIRE(dtype == 0 || dtype == 1, "dtype must be 0 (fp32) or 1 (bf16)");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0)
        hipLaunchKernelGGL((wmse_kernel<float, true>), dim3((unsigned)rows), dim3(kT), 0, st, (const float*)pred, tar, wrow,
                           (double*)nullptr, gloss, (float*)gpred, rows, H, W, scale);
    else
        hipLaunchKernelGGL((wmse_kernel<__hip_bfloat16, true>), dim3((unsigned)rows), dim3(kT), 0, st,
                           (const __hip_bfloat16*)pred, tar, wrow, (double*)nullptr, gloss, (__hip_bfloat16*)gpred, rows, H, W, scale);
    MK_LAUNCH_CHECK();
    return 0;
}
