// bf16 MFMA weight-gradient kernels for the 1x1 convolutions of the FNO block / encoder / decoder on NCHW fields
// viewed as [C][P = H*W] (the pointwise stack between the spectral ops).
//
// mk_conv1x1_wgrad:  gW[o][i] += sum_p gY[o][p] * X[i][p]    (contraction over the ~1e5..1e6 pixels)
//
// Both operands are contiguous along the contraction index p, which is exactly the fragment shape of
// v_mfma_f32_32x32x16_bf16 (lane (r, h) holds 8 consecutive k of row r): tiles are staged with plain
// 16-byte loads / ds_write_b128 into rows padded to 144 bytes (conflict-free ds_read_b128), no
// transposed reads.  A workgroup owns a 128 x 128 block of gW and a slab of pixels; the slabs are
// combined with fp32 atomics (shaped as 128-byte row segments).  All blocks of one pixel slab are dealt
// to the same XCD so the slab is read from HBM once and re-read from that XCD's L2.
#include "common.h"
#include "../../include/makani_amd.h"
#include "pce_common.h"

#include <hip/hip_bf16.h>
#include <cstdint>
#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// GELU on the 8 bf16 values of a staged vector (the x operand of `mk_conv1x1_wgrad_act`: the weight gradient of the second
// convolution of an MLP reads the kept PRE-activation and applies the activation on the way into LDS, so the hidden
// field itself never has to exist in HBM); exact-erf form rounded to bf16 = the values the forward pass multiplied with
__device__ __forceinline__ void act8(uint4& v) {
    uint32_t* w = reinterpret_cast<uint32_t*>(&v);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float lo = __uint_as_float(w[i] << 16), hi = __uint_as_float(w[i] & 0xFFFF0000u);
        w[i] = pce::pack_bf16x2(pce::gelu_f(lo), pce::gelu_f(hi));
    }
}

constexpr int WT = 256;          // threads (4 waves, 2 x 2)
constexpr int WTI = 128;                      // workgroup tile: (64 * MT) o  x  128 i, k-step TK pixels (template)

struct WgradParams {
    const __hip_bfloat16* gy;   // [B][O][P]
    const __hip_bfloat16* x;    // [B][I][P]
    float* gw;                  // [O][I], accumulated
    int O, I, B;
    long long P;
    int nblk_o, nblk_i, nslab;  // nslab slabs per batch item
    int slab;                   // pixels per slab (multiple of 64)
    int exp;                    // MK_WGRAD_EXP ablation: 1 = no atomic epilogue (wrong results)
};

// (NV * 256 / WVPR) rows x TK k bf16 tile: NV 16-byte vectors per thread, WVPR = TK / 8 vectors per row,
// LDS row pitch WPITCH = 2 TK + 16 bytes (an odd number of 16-byte units: conflict-free ds_read_b128)
// Loads are raw buffer loads: a lane outside the tile (row past the operand, pixel past the slab) gets an offset past the
// descriptor's range and reads zeros -- no exec-mask branch per vector, nothing that makes the compiler wait for a load
// where it is issued (same finding as in gemm_x3.hip).  Offsets are bytes from the block's first row (< 2^31: at most
// 256 rows of 2 P bytes).
template <int NV, int WVPR>
__device__ __forceinline__ void wg_load(const __hip_bfloat16* base, long long ld, int rows_valid, long long k0,
                                        long long kend, uint4 (&r)[NV], int tid) {
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<__hip_bfloat16*>(base), 0, 0x7FFFFFFF, 0x00020000);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int v = tid + i * WT;
        const int row = v / WVPR, c = v % WVPR;
        const long long k = k0 + c * 8;
        const unsigned off = (row < rows_valid && k < kend) ? (unsigned)(((long long)row * ld + k) * 2) : 0x80000000u;
        r[i] = __builtin_bit_cast(uint4, (u4)__builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
    }
}
template <int NV, int WVPR>
__device__ __forceinline__ void wg_store(char* lds, const uint4 (&r)[NV], int tid) {
    constexpr int WPITCH = WVPR * 16 + 16;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int v = tid + i * WT;
        const int row = v / WVPR, c = v % WVPR;
        *reinterpret_cast<uint4*>(lds + row * WPITCH + c * 16) = r[i];
    }
}

template <int MT, int WTK, int RD = 1, bool ACT = false>   // 32-row o-tiles per wave: workgroup tile (64 * MT) x 128; k-step (pixels); ring depth; GELU on x
__global__ __launch_bounds__(WT) void conv1x1_wgrad_kernel(WgradParams p) {
    constexpr int WVPR = WTK / 8, WPITCH = WTK * 2 + 16, WTILE_BYTES = WTI * WPITCH;
    constexpr int WTO = 64 * MT, ATILE = WTO * WPITCH, BUF = ATILE + WTILE_BYTES, NVA = WTO * WVPR / WT, NVB = WTI * WVPR / WT;
    extern __shared__ __attribute__((aligned(16))) char lds[];   // [2 buffers][A tile | B tile]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    // block -> (slab, o-block, i-block); all blocks of a slab share blockIdx % 8 (one XCD)
    const int nblk = p.nblk_o * p.nblk_i;
    const long long bid = blockIdx.x;
    const int xcd = (int)(bid & 7);
    const long long seq = bid >> 3;
    const long long slab_lin = (seq / nblk) * 8 + xcd;
    const int blk = (int)(seq % nblk);
    const long long nslab_tot = (long long)p.nslab * p.B;
    if (slab_lin >= nslab_tot) return;
    const int b = (int)(slab_lin / p.nslab);
    const long long k_begin = (slab_lin % p.nslab) * p.slab;
    const long long k_end = (k_begin + p.slab < p.P) ? k_begin + p.slab : p.P;
    const int o0 = (blk / p.nblk_i) * WTO, i0 = (blk % p.nblk_i) * WTI;
    const __hip_bfloat16* ga = p.gy + ((long long)b * p.O + o0) * p.P;
    const __hip_bfloat16* gb = p.x + ((long long)b * p.I + i0) * p.P;
    const int ov = p.O - o0, iv = p.I - i0;

    f32x16 acc[MT][2];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;
    // accumulators in the AccVGPR half of the register file (AGPR form of the MFMAs): see x3_tile in gemm_x3.hip
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c) asm volatile("" : "+a"(acc[a][c]));

    // Register ring RD k-steps deep: one k-step of MFMAs is only 16 x 32 = 512 cycles, far less than a load's flight
    // time, so the loads of k-step kt + 1 + RD are issued when the registers of k-step kt + 1 have been written to LDS.
    uint4 ra[RD][NVA], rb[RD][NVB];
    const int nk = (int)((k_end - k_begin + WTK - 1) / WTK);
    wg_load<NVA, WVPR>(ga, p.P, ov, k_begin, k_end, ra[0], tid);
    wg_load<NVB, WVPR>(gb, p.P, iv, k_begin, k_end, rb[0], tid);
    wg_store<NVA, WVPR>(lds, ra[0], tid);
    if constexpr (ACT) {
#pragma unroll
        for (int i = 0; i < NVB; ++i) act8(rb[0][i]);
    }
    wg_store<NVB, WVPR>(lds + ATILE, rb[0], tid);
#pragma unroll
    for (int d = 0; d < RD; ++d)
        if (d + 1 < nk) {
            const long long k0 = k_begin + (long long)(d + 1) * WTK;
            wg_load<NVA, WVPR>(ga, p.P, ov, k0, k_end, ra[d], tid);
            wg_load<NVB, WVPR>(gb, p.P, iv, k0, k_end, rb[d], tid);
        }
    __syncthreads();
    const int fr = lane & 31, fh = lane >> 5;
    for (int ktb = 0; ktb < nk; ktb += RD) {
#pragma unroll
        for (int d = 0; d < RD; ++d) {      // ring slot d holds k-step kt + 1
            const int kt = ktb + d;
            if (kt >= nk) break;
            const int cur = kt & 1;
            const char* As = lds + cur * BUF;
            const char* Bs = As + ATILE;
#pragma unroll
            for (int ks = 0; ks < WTK / 16; ++ks) {
                bf16x8 af[MT], bf[2];
#pragma unroll
                for (int a = 0; a < MT; ++a)
                    af[a] = *reinterpret_cast<const bf16x8*>(As + (wr * 32 * MT + a * 32 + fr) * WPITCH + ks * 32 + fh * 16);
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    bf[c] = *reinterpret_cast<const bf16x8*>(Bs + (wc * 64 + c * 32 + fr) * WPITCH + ks * 32 + fh * 16);
#pragma unroll
                for (int a = 0; a < MT; ++a)
#pragma unroll
                    for (int c = 0; c < 2; ++c)
                        acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bf[c], acc[a][c], 0, 0, 0);
            }
            if (kt + 1 < nk) {
                wg_store<NVA, WVPR>(lds + (cur ^ 1) * BUF, ra[d], tid);
                if constexpr (ACT) {
#pragma unroll
                    for (int i = 0; i < NVB; ++i) act8(rb[d][i]);
                }
                wg_store<NVB, WVPR>(lds + (cur ^ 1) * BUF + ATILE, rb[d], tid);
                if (kt + 1 + RD < nk) {
                    const long long k0 = k_begin + (long long)(kt + 1 + RD) * WTK;
                    wg_load<NVA, WVPR>(ga, p.P, ov, k0, k_end, ra[d], tid);
                    wg_load<NVB, WVPR>(gb, p.P, iv, k0, k_end, rb[d], tid);
                }
            }
            __syncthreads();
        }
    }
    // C/D map: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); rows = o, cols = i
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int col = i0 + wc * 64 + c * 32 + fr;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = o0 + wr * 32 * MT + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (row < p.O && col < p.I && (!(p.exp & 1) || acc[a][c][r] == 12345.678f))
                    atomicAdd(&p.gw[(long long)row * p.I + col], acc[a][c][r]);
            }
        }
}



// ---------------------------------------------------------------------------
// Large-block weight gradient: 512 threads (8 waves as WR x WC), a (WR*RT*32) x (WC*CT*32) block of gW per workgroup,
// one workgroup per CU.  All GEMM-shaped kernels of this library stall at the same ~10 TB/s of L2 -> CU traffic
// (~43 GB/s per CU); the 128 x 128 block moves 32 bytes per thousand multiply-adds through that path, 256 x 192 moves 18.
// Used where the block divides the matrix: 768 x 384 as 256 x 192 (4 x 2 waves of 64 x 96), 384 x 768 as 192 x 256.
// ---------------------------------------------------------------------------
constexpr int WT8 = 512;

template <int NV>
__device__ __forceinline__ void wg8_load(const __hip_bfloat16* base, long long ld, int rows_valid, long long k0,
                                         long long kend, uint4 (&r)[NV], int tid) {
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<__hip_bfloat16*>(base), 0, 0x7FFFFFFF, 0x00020000);     // see wg_load
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int v = tid + i * WT8, row = v >> 3, c = v & 7;
        const long long k = k0 + c * 8;
        const unsigned off = (row < rows_valid && k < kend) ? (unsigned)(((long long)row * ld + k) * 2) : 0x80000000u;
        r[i] = __builtin_bit_cast(uint4, (u4)__builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
    }
}
template <int NV>
__device__ __forceinline__ void wg8_store(char* lds, const uint4 (&r)[NV], int tid) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int v = tid + i * WT8, row = v >> 3, c = v & 7;
        *reinterpret_cast<uint4*>(lds + row * 144 + c * 16) = r[i];
    }
}

template <int WR, int WC, int RT, int CT, bool ACT = false>
__global__ __launch_bounds__(WT8) void conv1x1_wgrad_big_kernel(WgradParams p) {
    constexpr int TMB = WR * RT * 32, TNB = WC * CT * 32, PITCH = 144;
    constexpr int ATILE = TMB * PITCH, BTILE = TNB * PITCH, BUF = ATILE + BTILE;
    constexpr int NVA = TMB * 8 / WT8, NVB = TNB * 8 / WT8;
    static_assert(WR * WC == 8 && TMB * 8 % WT8 == 0 && TNB * 8 % WT8 == 0, "tile / thread mapping");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WC, wc = wave % WC;
    const int nblk = p.nblk_o * p.nblk_i;
    const long long bid = blockIdx.x;
    const int xcd = (int)(bid & 7);
    const long long seq = bid >> 3;
    const long long slab_lin = (seq / nblk) * 8 + xcd;
    const int blk = (int)(seq % nblk);
    const long long nslab_tot = (long long)p.nslab * p.B;
    if (slab_lin >= nslab_tot) return;
    const int b = (int)(slab_lin / p.nslab);
    const long long k_begin = (slab_lin % p.nslab) * p.slab;
    const long long k_end = (k_begin + p.slab < p.P) ? k_begin + p.slab : p.P;
    const int o0 = (blk / p.nblk_i) * TMB, i0 = (blk % p.nblk_i) * TNB;
    const __hip_bfloat16* ga = p.gy + ((long long)b * p.O + o0) * p.P;
    const __hip_bfloat16* gb = p.x + ((long long)b * p.I + i0) * p.P;
    const int ov = p.O - o0, iv = p.I - i0;

    f32x16 acc[RT][CT];
#pragma unroll
    for (int a = 0; a < RT; ++a)
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;
#pragma unroll
    for (int a = 0; a < RT; ++a)
#pragma unroll
        for (int c = 0; c < CT; ++c) asm volatile("" : "+a"(acc[a][c]));      // AGPR form of the MFMAs: see x3_tile in gemm_x3.hip

    uint4 ra[NVA], rb[NVB];
    const int nk = (int)((k_end - k_begin + 63) / 64);
    wg8_load<NVA>(ga, p.P, ov, k_begin, k_end, ra, tid);
    wg8_load<NVB>(gb, p.P, iv, k_begin, k_end, rb, tid);
    wg8_store<NVA>(lds, ra, tid);
    if constexpr (ACT) {
#pragma unroll
        for (int i = 0; i < NVB; ++i) act8(rb[i]);
    }
    wg8_store<NVB>(lds + ATILE, rb, tid);
    __syncthreads();
    const int fr = lane & 31, fh = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            const long long k0 = k_begin + (long long)(kt + 1) * 64;
            wg8_load<NVA>(ga, p.P, ov, k0, k_end, ra, tid);
            wg8_load<NVB>(gb, p.P, iv, k0, k_end, rb, tid);
        }
        const char* As = lds + cur * BUF;
        const char* Bs = As + ATILE;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 af[RT], bf[CT];
#pragma unroll
            for (int a = 0; a < RT; ++a)
                af[a] = *reinterpret_cast<const bf16x8*>(As + ((wr * RT + a) * 32 + fr) * PITCH + ks * 32 + fh * 16);
#pragma unroll
            for (int c = 0; c < CT; ++c)
                bf[c] = *reinterpret_cast<const bf16x8*>(Bs + ((wc * CT + c) * 32 + fr) * PITCH + ks * 32 + fh * 16);
#pragma unroll
            for (int a = 0; a < RT; ++a)
#pragma unroll
                for (int c = 0; c < CT; ++c)
                    acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bf[c], acc[a][c], 0, 0, 0);
        }
        if (kt + 1 < nk) {
            wg8_store<NVA>(lds + (cur ^ 1) * BUF, ra, tid);
            if constexpr (ACT) {
#pragma unroll
                for (int i = 0; i < NVB; ++i) act8(rb[i]);
            }
            wg8_store<NVB>(lds + (cur ^ 1) * BUF + ATILE, rb, tid);
        }
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < RT; ++a)
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const int col = i0 + (wc * CT + c) * 32 + fr;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = o0 + (wr * RT + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (row < p.O && col < p.I && (!(p.exp & 1) || acc[a][c][r] == 12345.678f))
                    atomicAdd(&p.gw[(long long)row * p.I + col], acc[a][c][r]);
            }
        }
}

}  // namespace

// slabs per batch item: all blocks of a slab run on one XCD (`slots` workgroup slots per XCD) and the slabs are dealt
// round-robin to the 8 XCDs: take s slabs per XCD so that s * (blocks per slab) just fills w rounds of the slots (w <= 4,
// fewest rounds among the well-filled choices).  Measured on 768x384 / 384x384 blocks: 56 slabs (2 x 63 and 1 x 63 of 64
// slots) beat a ">= 1536 workgroups" rule by 17-25 %, 28 or 86 slabs lose to the partly empty last round.
static void wgrad_slabs(WgradParams& p, long long nblk1, int slots, int batch) {
    long long best_s = 1;
    double best = -1.0;
    for (int w = 1; w <= 4; ++w) {
        const long long sx = ((long long)slots * w) / nblk1;
        if (sx < 1) continue;
        const double score = (double)(sx * nblk1) / ((double)slots * w) - 0.03 * w;
        if (score > best) {
            best = score;
            best_s = sx;
        }
    }
    long long want = (8 * best_s + batch - 1) / batch;
    if (want < 1) want = 1;
    long long slab = (p.P + want - 1) / want;
    slab = (slab + 63) / 64 * 64;
    if (slab < 512) slab = 512;
    p.slab = (int)slab;
    p.nslab = (int)((p.P + p.slab - 1) / p.slab);
}

template <int WR, int WC, int RT, int CT, bool ACT>
static int wgrad_big_launch(WgradParams& p, int batch, hipStream_t st) {
    constexpr int TMB = WR * RT * 32, TNB = WC * CT * 32, LDS = 2 * (TMB + TNB) * 144;
    p.nblk_o = mk::ceil_div(p.O, TMB);
    p.nblk_i = mk::ceil_div(p.I, TNB);
    const long long nblk1 = (long long)p.nblk_o * p.nblk_i;
    wgrad_slabs(p, nblk1, 32, batch);                 // 32 workgroup slots per XCD (one 512-thread workgroup per CU)
    const long long grid = (((long long)p.nslab * batch + 7) / 8) * 8 * nblk1;
    if (grid >= 2147483647LL) return 1;
    static const bool once = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv1x1_wgrad_big_kernel<WR, WC, RT, CT, ACT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        return true;
    }();
    (void)once;
    hipLaunchKernelGGL((conv1x1_wgrad_big_kernel<WR, WC, RT, CT, ACT>), dim3((unsigned)grid), dim3(WT8), LDS, st, p);
    return 0;
}

template <int MT, bool ACT>
static int wgrad_block_launch(WgradParams& p, int batch, hipStream_t st) {
    constexpr int WTO = 64 * MT, LDS = 2 * (WTO + WTI) * (64 * 2 + 16);
    p.nblk_o = mk::ceil_div(p.O, WTO);
    p.nblk_i = mk::ceil_div(p.I, WTI);
    wgrad_slabs(p, (long long)p.nblk_o * p.nblk_i, MT == 4 ? 32 : 64, batch);
    const long long grid = (((long long)p.nslab * batch + 7) / 8) * 8 * p.nblk_o * p.nblk_i;
    if (grid >= 2147483647LL) return 1;
    static const bool once = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv1x1_wgrad_kernel<MT, 64, 1, ACT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        return true;
    }();
    (void)once;
    hipLaunchKernelGGL((conv1x1_wgrad_kernel<MT, 64, 1, ACT>), dim3((unsigned)grid), dim3(WT), LDS, st, p);
    return 0;
}

extern "C" int mk_conv1x1_wgrad_act(const void* gy, const void* x, float* gw, int batch, int cout, int cin, long long P,
                                    int x_act, void* stream) {
    MK_REQUIRE(gy && x && gw, "null pointer");
    MK_REQUIRE(batch > 0 && cout > 0 && cin > 0 && P > 0, "bad sizes");
    MK_REQUIRE((P % 8) == 0, "P = H*W must be a multiple of 8 (16-byte row alignment)");
    MK_REQUIRE((((uintptr_t)gy | (uintptr_t)x) & 15) == 0, "gy and x must be 16-byte aligned (the tiles are read as uint4)");
    MK_REQUIRE(x_act == 0 || x_act == 1, "x_act must be 0 (none) or 1 (exact GELU)");
    MK_REQUIRE(x_act == 0 || cout <= 384, "the activated operand is built for cout <= 384 (one block row covers all of gy)");
    WgradParams p;
    p.gy = (const __hip_bfloat16*)gy;
    p.x = (const __hip_bfloat16*)x;
    p.gw = gw;
    p.O = cout;
    p.I = cin;
    p.B = batch;
    p.P = P;
    static const int wexp = [] { const char* e = getenv("MK_WGRAD_EXP"); return e ? atoi(e) : 0; }();
    p.exp = wexp;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (x_act) {
        // every x tile must be staged by ONE workgroup (the activation is evaluated while staging): block rows that cover all of gy
        if (cout <= 128) rc = wgrad_block_launch<2, true>(p, batch, st);
        else if (cout <= 256) rc = wgrad_block_launch<4, true>(p, batch, st);
        else rc = wgrad_big_launch<4, 2, 3, 2, true>(p, batch, st);            // 384 x 128 blocks
    } else {
        // Large blocks where they divide the matrix exactly and the contraction is long: measured 10 % faster at 721 x 1440
        // pixels (0.96 vs 1.07 ms for 768 x 384), 35 % slower at 240 x 480 where one workgroup per CU cannot hide its own
        // load latency over 45 k-steps.
        const bool big_ok = P * batch >= 400000;
        if (big_ok && cout % 256 == 0 && cin % 192 == 0) rc = wgrad_big_launch<4, 2, 2, 3, false>(p, batch, st);        // 256 x 192
        else if (big_ok && cout % 192 == 0 && cin % 256 == 0) rc = wgrad_big_launch<2, 4, 3, 2, false>(p, batch, st);   // 192 x 256
        else rc = wgrad_block_launch<2, false>(p, batch, st);                                                       // 128 x 128
    }
    MK_REQUIRE(rc == 0, "grid too large");
    MK_LAUNCH_CHECK();
    return 0;
}

extern "C" int mk_conv1x1_wgrad(const void* gy, const void* x, float* gw, int batch, int cout, int cin, long long P,
                                void* stream) {
    return mk_conv1x1_wgrad_act(gy, x, gw, batch, cout, cin, P, 0, stream);
}
