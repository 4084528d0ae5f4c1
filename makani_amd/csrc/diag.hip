// "diagonal" spectral filter: a separate complex weight for every (l, m),
//     y[b][o][p] = sum_i x[b][i][p] * w[i][o][p],      p = l * M + m
// on the reference's own (public) layout -- x [B][I][P], w [I][O][P], y [B][O][P], complex64, P contiguous.
// Replaces _contract_diagonal `einsum("bixy,ioxy->boxy")` (makani/models/common/contractions.py:121-127) and its
// two gradients.  The contraction is elementwise in p: one flop per byte of weight, HBM bound, so these are
// streaming kernels (a lane owns one p, every access is coalesced along p) -- not GEMMs.
#include "common.h"
#include "../../include/makani_amd.h"

namespace {

constexpr int DT = 256;   // threads: consecutive p
constexpr int DB = 4;     // batch items (forward / dgrad) accumulated per pass over the weights

__device__ __forceinline__ float2 cfma(float2 a, float2 b, float2 c) {   // c + a * b
    c.x = fmaf(a.x, b.x, c.x);
    c.x = fmaf(-a.y, b.y, c.x);
    c.y = fmaf(a.x, b.y, c.y);
    c.y = fmaf(a.y, b.x, c.y);
    return c;
}
__device__ __forceinline__ float2 cfma_conj_b(float2 a, float2 b, float2 c) {   // c + a * conj(b)
    c.x = fmaf(a.x, b.x, c.x);
    c.x = fmaf(a.y, b.y, c.x);
    c.y = fmaf(a.y, b.x, c.y);
    c.y = fmaf(-a.x, b.y, c.y);
    return c;
}

// CONJ = false: y[b][o][p] = sum_i x[b][i][p] w[i][o][p]          (grid.y = o, contraction over i)
// CONJ = true:  gx[b][i][p] = sum_o gy[b][o][p] conj(w[i][o][p])  (grid.y = i, contraction over o)
template <bool CONJ>
__global__ __launch_bounds__(DT) void diag_apply_kernel(const float2* __restrict__ src, const float2* __restrict__ w,
                                                        float2* __restrict__ dst, int B, int I, int O, long long P) {
    const long long p = (long long)blockIdx.x * DT + threadIdx.x;
    if (p >= P) return;
    const int row = blockIdx.y;
    const int nred = CONJ ? O : I, nsrc = CONJ ? O : I, ndst = CONJ ? I : O;
    for (int b0 = 0; b0 < B; b0 += DB) {
        float2 acc[DB];
#pragma unroll
        for (int bb = 0; bb < DB; ++bb) acc[bb] = make_float2(0.f, 0.f);
        for (int r = 0; r < nred; ++r) {
            const float2 wv = CONJ ? w[((long long)row * O + r) * P + p] : w[((long long)r * O + row) * P + p];
#pragma unroll
            for (int bb = 0; bb < DB; ++bb)
                if (b0 + bb < B) {
                    const float2 sv = src[((long long)(b0 + bb) * nsrc + r) * P + p];
                    acc[bb] = CONJ ? cfma_conj_b(sv, wv, acc[bb]) : cfma(sv, wv, acc[bb]);
                }
        }
#pragma unroll
        for (int bb = 0; bb < DB; ++bb)
            if (b0 + bb < B) dst[((long long)(b0 + bb) * ndst + row) * P + p] = acc[bb];
    }
}

// gw[i][o][p] = sum_b conj(x[b][i][p]) gy[b][o][p]     (grid.y = o, grid.z = i)
__global__ __launch_bounds__(DT) void diag_wgrad_kernel(const float2* __restrict__ x, const float2* __restrict__ gy,
                                                        float2* __restrict__ gw, int B, int I, int O, long long P) {
    const long long p = (long long)blockIdx.x * DT + threadIdx.x;
    if (p >= P) return;
    const int o = blockIdx.y, i = blockIdx.z;
    float2 acc = make_float2(0.f, 0.f);
    for (int b = 0; b < B; ++b)
        acc = cfma_conj_b(gy[((long long)b * O + o) * P + p], x[((long long)b * I + i) * P + p], acc);
    gw[((long long)i * O + o) * P + p] = acc;
}

int diag_check(const void* a, const void* b, const void* c, int batch, int cin, int cout, long long P) {
    MK_REQUIRE(a && b && c, "null pointer");
    MK_REQUIRE(batch > 0 && cin > 0 && cout > 0 && P > 0, "bad sizes");
    MK_REQUIRE(cin <= 65535 && cout <= 65535 && (P + DT - 1) / DT < 2147483647LL, "grid too large");
    return 0;
}

}  // namespace

extern "C" int mk_diag_fwd(const float* x, const float* w, float* y, int batch, int cin, int cout, long long P, void* stream) {
    if (int e = diag_check(x, w, y, batch, cin, cout, P)) return e;
    const dim3 grid((unsigned)((P + DT - 1) / DT), (unsigned)cout);
    hipLaunchKernelGGL(diag_apply_kernel<false>, grid, dim3(DT), 0, (hipStream_t)stream, (const float2*)x, (const float2*)w,
                       (float2*)y, batch, cin, cout, P);
    MK_LAUNCH_CHECK();
    return 0;
}

extern "C" int mk_diag_dgrad(const float* gy, const float* w, float* gx, int batch, int cin, int cout, long long P,
                             void* stream) {
    if (int e = diag_check(gy, w, gx, batch, cin, cout, P)) return e;
    const dim3 grid((unsigned)((P + DT - 1) / DT), (unsigned)cin);
    hipLaunchKernelGGL(diag_apply_kernel<true>, grid, dim3(DT), 0, (hipStream_t)stream, (const float2*)gy, (const float2*)w,
                       (float2*)gx, batch, cin, cout, P);
    MK_LAUNCH_CHECK();
    return 0;
}

extern "C" int mk_diag_wgrad(const float* x, const float* gy, float* gw, int batch, int cin, int cout, long long P,
                             void* stream) {
    if (int e = diag_check(x, gy, gw, batch, cin, cout, P)) return e;
    const dim3 grid((unsigned)((P + DT - 1) / DT), (unsigned)cout, (unsigned)cin);
    hipLaunchKernelGGL(diag_wgrad_kernel, grid, dim3(DT), 0, (hipStream_t)stream, (const float2*)x, (const float2*)gy,
                       (float2*)gw, batch, cin, cout, P);
    MK_LAUNCH_CHECK();
    return 0;
}
