// Layout conversion between the public torch layout [BC][L][M] complex64 and the
// private channels-last spectral layout [L][M][BC] complex64 (see makani_amd.h).
// A complex transpose of a [BC] x [L*M] matrix through 32 x 32 LDS tiles; HBM bound.
#include "common.h"
#include "../../include/makani_amd.h"

namespace {

constexpr int TS = 32;

// dst[c][r] = src[r][c] for a [R][C] float2 matrix -> [C][R]; optional zero mask on
// the (l, m) index when it is the column (pack: no mask) or row (unpack) index.
// UNPACK: src = private [LM][BC], dst = std [BC][LM], zero where l_off + l < m_off + m.
template <bool UNPACK>
__global__ __launch_bounds__(256) void transpose_c64_kernel(const float2* __restrict__ src, float2* __restrict__ dst,
                                                            int R, int C, int mloc, int l_off, int m_off) {
    __shared__ float2 tile[TS][TS + 1];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const long long tiles_c = (C + TS - 1) / TS;
    const long long bid = blockIdx.x;
    const int r0 = (int)(bid / tiles_c) * TS, c0 = (int)(bid % tiles_c) * TS;
#pragma unroll
    for (int j = 0; j < TS; j += 8) {
        const int r = r0 + ty + j, c = c0 + tx;
        float2 v = make_float2(0.f, 0.f);
        if (r < R && c < C) {
            v = src[(long long)r * C + c];
            if (UNPACK) {  // rows are (l, m)
                const int l = r / mloc, m = r - l * mloc;
                if (l_off + l < m_off + m) v = make_float2(0.f, 0.f);
            }
        }
        tile[ty + j][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TS; j += 8) {
        const int c = c0 + ty + j, r = r0 + tx;
        if (r < R && c < C) dst[(long long)c * R + r] = tile[tx][ty + j];
    }
}

}  // namespace

extern "C" int mk_spec_pack(const float* c_std, float* c_prv, int bc, int lloc, int mloc, void* stream) {
    MK_REQUIRE(c_std && c_prv, "null pointer");
    MK_REQUIRE(bc > 0 && lloc > 0 && mloc > 0, "bad sizes");
    const int R = bc, C = lloc * mloc;
    const long long nblk = (long long)mk::ceil_div(R, TS) * mk::ceil_div(C, TS);
    MK_REQUIRE(nblk < 2147483647LL, "grid too large");
    hipLaunchKernelGGL(transpose_c64_kernel<false>, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream,
                       (const float2*)c_std, (float2*)c_prv, R, C, mloc, 0, 0);
    MK_LAUNCH_CHECK();
    return 0;
}

extern "C" int mk_spec_unpack(const float* c_prv, float* c_std, int bc, int lloc, int mloc, int l_off, int m_off,
                              void* stream) {
    MK_REQUIRE(c_std && c_prv, "null pointer");
    MK_REQUIRE(bc > 0 && lloc > 0 && mloc > 0, "bad sizes");
    const int R = lloc * mloc, C = bc;
    const long long nblk = (long long)mk::ceil_div(R, TS) * mk::ceil_div(C, TS);
    MK_REQUIRE(nblk < 2147483647LL, "grid too large");
    hipLaunchKernelGGL(transpose_c64_kernel<true>, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream,
                       (const float2*)c_prv, (float2*)c_std, R, C, mloc, l_off, m_off);
    MK_LAUNCH_CHECK();
    return 0;
}
