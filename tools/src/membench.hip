// HBM pattern microbenchmark for the pixel-column engine: persistent workgroups move [rows][S-byte] row segments of a
// [rows][P] bf16 matrix (rows 2 P bytes apart) -- the access pattern of the engine's X tiles and Y tiles -- and report
// GB/s.  usage: membench rows P_px seg_bytes mode(0 copy, 1 read, 2 write, 3 read by LDS-DMA, 4 write nontemporal, 5 copy with nontemporal stores) order(0 strided tiles, 1 blocked tiles) [wgs]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, bool NT = false>
__global__ __launch_bounds__(256) void seg_kernel(const char* __restrict__ src, char* __restrict__ dst, long long rowbytes,
                                                  int rows, int seg, long long ntiles, int blocked, unsigned* sink) {
    const int lanes_per_row = seg / 16;                 // threads that cover one row segment
    const int rows_per_pass = 256 / lanes_per_row;      // rows one workgroup instruction covers
    const int r0 = threadIdx.x / lanes_per_row, c0 = (threadIdx.x % lanes_per_row) * 16;
    const long long per = (ntiles + gridDim.x - 1) / gridDim.x;
    u32x4 acc = {0, 0, 0, 0};
    for (long long i = 0; i < per; ++i) {
        const long long t = blocked ? blockIdx.x * per + i : i * gridDim.x + blockIdx.x;
        if (t >= ntiles) break;
        const long long col = t * seg + c0;
        for (int r = r0; r < rows; r += rows_per_pass * 8) {
            u32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int rr = r + u * rows_per_pass;
                if (MODE != 2) v[u] = rr < rows ? *reinterpret_cast<const u32x4*>(src + rr * rowbytes + col) : acc;
                else v[u] = acc;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int rr = r + u * rows_per_pass;
                if (MODE != 1) {
                    if (rr < rows) {
                        if (NT) __builtin_nontemporal_store(v[u], reinterpret_cast<u32x4*>(dst + rr * rowbytes + col));
                        else *reinterpret_cast<u32x4*>(dst + rr * rowbytes + col) = v[u];
                    }
                } else {
                    acc ^= v[u];
                }
            }
        }
    }
    if (MODE == 1 && acc[0] == 0x12345678u) sink[0] = acc[1];
}

// mode 3: the read side through LDS-DMA (global_load_lds_dwordx4, 1 KB per wave instruction = 8 rows x 128 B or 4 rows x
// 256 B ...), nothing stored: what the engine's X path can reach
__global__ __launch_bounds__(256) void dma_kernel(const char* __restrict__ src, long long rowbytes, int rows, int seg,
                                                  long long ntiles, unsigned* sink, int depth) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lanes_per_row = seg / 16, rows_per_piece = 64 / lanes_per_row;
    const int r0 = lane / lanes_per_row, c0 = (lane % lanes_per_row) * 16;
    int slot = 0;
    for (long long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const long long col = t * seg + c0;
        for (int r = wave * rows_per_piece + r0; r < rows; r += 4 * rows_per_piece) {
            const char* g = src + (long long)r * rowbytes + col;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(lds + (slot * 4 + wave) * 1024), 16, 0, 0);
            if (++slot == depth) slot = 0;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lds[threadIdx.x] == 77 && src[0] == 78) sink[0] = 1;
}

int main(int argc, char** argv) {
    const int rows = argc > 1 ? atoi(argv[1]) : 384;
    const long long P = argc > 2 ? atoll(argv[2]) : 721LL * 1440;
    const int seg = argc > 3 ? atoi(argv[3]) : 128;
    const int mode = argc > 4 ? atoi(argv[4]) : 0;
    const int blocked = argc > 5 ? atoi(argv[5]) : 0;
    const int wgs = argc > 6 ? atoi(argv[6]) : 256;
    const long long rowbytes = 2 * P, bytes = rowbytes * rows;
    char *a, *b;
    unsigned* sink;
    (void)hipMalloc(&a, bytes + 4096);
    (void)hipMalloc(&b, bytes + 4096);
    (void)hipMalloc(&sink, 64);
    (void)hipMemset(a, 1, bytes);
    (void)hipMemset(b, 2, bytes);
    const long long ntiles = rowbytes / seg;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 36 * 4096);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < 6; ++it) {
        (void)hipEventRecord(e0);
        for (int k = 0; k < 4; ++k) {
            if (mode == 0) hipLaunchKernelGGL(seg_kernel<0>, dim3(wgs), dim3(256), 0, 0, a, b, rowbytes, rows, seg, ntiles, blocked, sink);
            if (mode == 1) hipLaunchKernelGGL(seg_kernel<1>, dim3(wgs), dim3(256), 0, 0, a, b, rowbytes, rows, seg, ntiles, blocked, sink);
            if (mode == 2) hipLaunchKernelGGL(seg_kernel<2>, dim3(wgs), dim3(256), 0, 0, a, b, rowbytes, rows, seg, ntiles, blocked, sink);
            if (mode == 4) hipLaunchKernelGGL((seg_kernel<2, true>), dim3(wgs), dim3(256), 0, 0, a, b, rowbytes, rows, seg, ntiles, blocked, sink);
            if (mode == 5) hipLaunchKernelGGL((seg_kernel<0, true>), dim3(wgs), dim3(256), 0, 0, a, b, rowbytes, rows, seg, ntiles, blocked, sink);
            if (mode == 3) hipLaunchKernelGGL(dma_kernel, dim3(wgs), dim3(256), 36 * 4096, 0, a, rowbytes, rows, seg, ntiles, sink, 36);
        }
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (it > 0 && ms / 4 < best) best = ms / 4;
    }
    const double moved = ((mode == 0 || mode == 5) ? 2.0 : 1.0) * (double)ntiles * seg * rows;
    printf("rows %d seg %4d B mode %d order %d wgs %d: %.3f ms  %.0f GB/s\n", rows, seg, mode, blocked, wgs, best, moved / best / 1e6);
    return 0;
}
