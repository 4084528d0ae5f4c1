// Adam update as one streaming pass over a parameter tensor (the optimizer step of the training harness,
// makani/utils/trainer.py:762-763 -- torch.optim.Adam on the model's parameters, complex ones as pairs of reals).
// HBM-bound: 4 fp32 reads (p, g, m, v) + 3 writes per element = 28 bytes; the 283 M spectral weights of the
// north-star net are 7.9 GB per step.  torch's multi-tensor fused kernel moves them at 2.1 TB/s on MI355X (3.9 ms per
// step); a plain grid-stride float4 pass per large tensor runs at the streaming rate of the chip.
#include "common.h"
#include "../../include/makani_amd.h"

#include <cstdint>

namespace {

struct AdamArgs {
    float lr, beta1, beta2, eps, weight_decay, bc1, bc2_sqrt;
};

__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, const AdamArgs& a) {
    g = fmaf(a.weight_decay, p, g);
    m = fmaf(a.beta1, m, (1.f - a.beta1) * g);
    v = fmaf(a.beta2, v, (1.f - a.beta2) * g * g);
    const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
    p -= (a.lr / a.bc1) * (m / denom);
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long long n4, long long n, AdamArgs a) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 pp = reinterpret_cast<float4*>(p)[i];
        const float4 gg = reinterpret_cast<const float4*>(g)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
        adam1(pp.x, gg.x, mm.x, vv.x, a);
        adam1(pp.y, gg.y, mm.y, vv.y, a);
        adam1(pp.z, gg.z, mm.z, vv.z, a);
        adam1(pp.w, gg.w, mm.w, vv.w, a);
        reinterpret_cast<float4*>(p)[i] = pp;
        reinterpret_cast<float4*>(m)[i] = mm;
        reinterpret_cast<float4*>(v)[i] = vv;
    }
    // tail (n not a multiple of 4)
    for (long long i = 4 * n4 + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) adam1(p[i], g[i], m[i], v[i], a);
}

}  // namespace

extern "C" int mk_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                            float eps, float weight_decay, int step, void* stream) {
    MK_REQUIRE(p && g && m && v, "null pointer");
    MK_REQUIRE(n > 0 && step >= 1, "bad sizes");
    MK_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "buffers must be 16-byte aligned");
    AdamArgs a;
    a.lr = lr;
    a.beta1 = beta1;
    a.beta2 = beta2;
    a.eps = eps;
    a.weight_decay = weight_decay;
    a.bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    a.bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    const long long n4 = n / 4;
    long long blocks = (n4 + 255) / 256;
    if (blocks > 256LL * 16) blocks = 256LL * 16;     // 16 workgroups per CU, grid-stride beyond that
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n4, n, a);
    MK_LAUNCH_CHECK();
    return 0;
}
