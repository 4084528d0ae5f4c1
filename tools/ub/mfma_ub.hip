// Microbenchmark: fp32 MFMA issue rate under the access patterns of gemm.hip (tuning aid, not product code).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#ifndef RANDOM
#define RANDOM 0
#endif
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, int ldsbytes) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 8192; i += 256) { unsigned h = (i + 1u) * 2654435761u + blockIdx.x * 40503u; h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15; lds[i] = RANDOM ? ((int)(h & 0xffffff) - 0x800000) * (1.0f / 0x800000) : 0.001f * i; }
    __syncthreads();
    f32x16 acc[NACC];
    for (int n = 0; n < NACC; ++n)
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    float a = lane * 0.01f, b = lane * 0.02f;
    const float* ap = lds + (lane & 31) + (lane >> 5) * 65;
    const float* bp = lds + 4096 + (lane & 31) + (lane >> 5) * 128;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int ks = 0; ks < 16; ++ks)
#pragma unroll
                for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[n], 0, 0, 0);
        } else {
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                const float av = ap[ks * 130];
                float bv[NACC];
#pragma unroll
                for (int n = 0; n < NACC; ++n) bv[n] = bp[ks * 256 + n * 32];
#pragma unroll
                for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[n], acc[n], 0, 0, 0);
            }
            if (MODE == 2) __syncthreads();
        }
    }
    float s = 0.f;
    for (int n = 0; n < NACC; ++n)
        for (int r = 0; r < 16; ++r) s += acc[n][r];
    out[blockIdx.x * 256 + tid] = s;
}

template <int MODE, int NACC>
void run(const char* name, int wgs, int lds, int iters) {
    float* out;
    hipMalloc(&out, sizeof(float) * wgs * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, NACC>), dim3(wgs), dim3(256), lds, 0, out, iters, lds);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double flops = (double)wgs * 4 * iters * 16 * NACC * 4096.0;
        if (rep == 2) printf("%-28s wgs=%5d lds=%6d: %.3f ms  %.1f TFLOP/s\n", name, wgs, lds, ms, flops / ms / 1e9);
    }
    hipFree(out);
}

int main() {
    run<2, 2>("short 24it 3456wg lds 49920", 3456, 49920, 24);
    run<2, 2>("short 24it 3456wg lds 32768", 3456, 32768+4096, 24);
    run<2, 2>("short 24it 6912wg lds 49920", 6912, 49920, 24);
    run<2, 2>("short 12it 6912wg lds 49920", 6912, 49920, 12);
    run<2, 2>("short 48it 1728wg lds 49920", 1728, 49920, 48);
    run<2, 2>("short 240it 768wg lds 49920", 768, 49920, 240);
    run<2, 2>("short 24it 3456wg lds 65536", 3456, 65536+8192, 24);
    run<2, 2>("short WGs: 24 iters, 768 wgs", 768, 49920, 24);
    run<2, 2>("short WGs: 96 iters, 3456 wgs", 3456, 49920, 96);
    run<2, 2>("short WGs: 24 iters, 34560 wgs", 34560, 49920, 24);
    const int it = 2000;
    run<0, 2>("pure mfma, 2 acc, 1wg/cu", 256, 65536 * 2, it);
    run<0, 2>("pure mfma, 2 acc, 3wg/cu", 768, 49920, it);
    run<0, 4>("pure mfma, 4 acc, 1wg/cu", 256, 65536 * 2, it);
    run<1, 2>("lds+mfma,  2 acc, 1wg/cu", 256, 65536 * 2, it);
    run<1, 2>("lds+mfma,  2 acc, 3wg/cu", 768, 49920, it);
    run<1, 4>("lds+mfma,  4 acc, 2wg/cu", 512, 66304, it);
    run<2, 2>("lds+mfma+barrier 2acc 3wg", 768, 49920, it);
    run<2, 4>("lds+mfma+barrier 4acc 2wg", 512, 66304, it);
    return 0;
}
