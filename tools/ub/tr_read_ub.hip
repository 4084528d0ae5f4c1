// Semantics check of ds_read_b64_tr_b16 for the 32x32x16 bf16 MFMA B operand (k-major LDS tile).
// build: hipcc -O2 --offload-arch=gfx950 tools/ub/tr_read_ub.hip -o /tmp/tr_read_ub
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef short s16x4 __attribute__((ext_vector_type(4)));
constexpr int PITCH = 320;   // bytes per k row (128 n * 2 B + 64 pad)

__global__ void k_tr(const unsigned short* src /*[16][128]*/, unsigned short* out /*[64 lanes][8]*/, int n0) {
    __shared__ __attribute__((aligned(16))) char lds[16 * PITCH];
    for (int i = threadIdx.x; i < 16 * 128; i += 64) {
        const int k = i / 128, n = i % 128;
        *reinterpret_cast<unsigned short*>(lds + k * PITCH + n * 2) = src[i];
    }
    __syncthreads();
    const int L = threadIdx.x;
    unsigned short res[8];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int row = 8 * (L >> 5) + 4 * r + ((L & 15) >> 2);
        const int col = n0 + 16 * ((L >> 4) & 1) + 4 * (L & 3);
        const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (s16x4 __attribute__((address_space(3)))*)(__attribute__((address_space(3))) char*)(lds + row * PITCH + col * 2));
        for (int q = 0; q < 4; ++q) res[4 * r + q] = (unsigned short)v[q];
    }
    for (int q = 0; q < 8; ++q) out[L * 8 + q] = res[q];
}

int main() {
    std::vector<unsigned short> h(16 * 128), o(64 * 8);
    for (int k = 0; k < 16; ++k)
        for (int n = 0; n < 128; ++n) h[k * 128 + n] = (unsigned short)(k * 128 + n);   // raw 16-bit tags
    unsigned short *d, *dout;
    hipMalloc(&d, h.size() * 2);
    hipMalloc(&dout, o.size() * 2);
    hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    int bad = 0;
    for (int n0 : {0, 32, 96}) {
        hipLaunchKernelGGL(k_tr, dim3(1), dim3(64), 0, 0, d, dout, n0);
        hipMemcpy(o.data(), dout, o.size() * 2, hipMemcpyDeviceToHost);
        for (int L = 0; L < 64; ++L)
            for (int q = 0; q < 8; ++q) {
                const int k = 8 * (L >> 5) + q, n = n0 + (L & 31);
                if (o[L * 8 + q] != k * 128 + n) {
                    if (bad < 10) printf("n0 %d lane %d q %d: got (k %d, n %d) want (k %d, n %d)\n", n0, L, q, o[L * 8 + q] / 128, o[L * 8 + q] % 128, k, n);
                    ++bad;
                }
            }
    }
    printf(bad ? "FAIL %d\n" : "tr_read OK\n", bad);
    return bad != 0;
}
