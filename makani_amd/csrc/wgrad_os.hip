// Weight gradient of the 1x1 convolutions, output-stationary:  gW[o][i] += sum_{b,p} gY[b][o][p] * X[b][i][p]
// (makani/models/common/layers.py:86-216 run backward; contraction over the 1e5..1e6 pixels of the field).
//
// A pure read-streaming problem: (O + I) rows of P pixels in, O x I numbers out.  A workgroup of 4 waves (one per SIMD)
// keeps a 192 x RB block of gW in registers for the whole launch -- wave (wa, wb) owns 96 x RB/2: 9 or 18 accumulators of
// 32 x 32 -- and walks over 64-pixel tiles of its pixel stream.  Per tile the 192 + RB operand rows arrive by LDS-DMA in a
// ring of LDS buffers while an earlier tile is being multiplied; both operands are contiguous along the contraction index,
// so an MFMA fragment is one ds_read_b128 of 8 pixels of a row (16-byte chunks rotated by row >> 1: conflict free), 6 or 9
// fragment reads per 9 or 18 MFMAs.  At the end every workgroup adds its block to gW with fp32 atomics (128-byte row
// segments).  RB = 192 (default): three 48 KB buffers, two tiles in flight, every operand row travels L2 -> CU twice;
// RB = 384: two 72 KB buffers, one tile in flight, 1.5 times (the 128 x 128-block kernel of conv_gemm.hip: 3 times).  HBM ->
// L2 once either way (the workgroups w and w + 8 of an XCD walk the same tiles on different blocks).
//
// Measured (tools/wgrad_bench.py, 1x MI355X): RB = 192 0.52 ms for 384 x 384 at 721 x 1440 (block kernel 0.55), but
// 0.088 vs 0.066 ms at 240 x 480 (the closing round of atomics and the ring prologue are fixed costs) and 0.27 vs 0.20 ms
// for 384 x 73 (the block is mostly padding); RB = 384 is latency-bound with its single tile in flight (0.59 / 0.13 ms).
// The block kernel stays the default; MK_WGRAD=os selects this one (MK_WGRAD_OS_RB=384 the larger block).
#include "common.h"
#include "lds_dma.h"
#include "../../include/makani_amd.h"

#include <hip/hip_bf16.h>
#include <cstdint>
#include <cstdlib>
#include <type_traits>
#include <utility>

namespace {

using namespace mkdma;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int OS_THREADS = 256;
constexpr int OS_PN = 64;                      // pixels per tile
constexpr int OS_RA = 192;                     // rows of operand a per workgroup; operand b: RB = 384 or 192 (template)

// RB = 384: 192 x 384 block, two 72 KB tile buffers (one tile in flight while one is multiplied).
// RB = 192: 192 x 192 block, three 48 KB buffers (two tiles in flight): more L2 -> CU traffic per flop (each operand row
//           travels twice instead of 1.5 times), twice the latency cover.
template <int RB>
struct OsGeom {
    static constexpr int BUF = (OS_RA + RB) * 128;          // one tile: rows x 64 px bf16
    static constexpr int NBUF = RB == 384 ? 2 : 3;
    static constexpr int LDS = NBUF * BUF;
    static constexpr int NPW = (OS_RA + RB) / 8 / 4;        // DMA pieces (8 rows x 64 px) per wave and tile
    static constexpr int NPA = OS_RA / 8 / 4;               // ... of which operand a
    static constexpr int TB = RB / 64;                      // 32-row tiles of operand b per wave
};

struct WgosParams {
    const __hip_bfloat16* a;      // [B][A][P]  operand of the block rows
    const __hip_bfloat16* b;      // [B][Bn][P] operand of the block columns
    const char* zeros;            // 64 zero bytes: the source of DMA lanes whose pixels lie past the end of the field
    float* out;                   // element (ia, ib) at out[ia * ld_a + ib * ld_b]
    int A, Bn, batch;
    int nblk_a, nblk_b;
    long long ld_a, ld_b;
    long long P, tiles_per_b;
};

template <int RB>
__global__ __launch_bounds__(OS_THREADS, 1) void wgrad_os_kernel(WgosParams p) {
    using G = OsGeom<RB>;
    constexpr int OS_BUF = G::BUF, NBUF = G::NBUF, NPW = G::NPW, NPA = G::NPA, TB = G::TB;
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wa = wave >> 1, wb = wave & 1;
    const int nblk = p.nblk_a * p.nblk_b;
    const int blk = ((int)blockIdx.x >> 3) % nblk;
    const int stream = ((int)blockIdx.x & 7) + 8 * ((int)blockIdx.x / (8 * nblk));
    const int nstreams = (int)gridDim.x / nblk;
    const int a0 = (blk / p.nblk_b) * OS_RA, b0 = (blk % p.nblk_b) * RB;         // first rows of this block
    const int tiles_per_b = (int)p.tiles_per_b;
    const long long rowbytes = 2 * p.P;
    auto opaque_lane = [&]() {
        int l = lane;
        asm volatile("" : "+v"(l));
        return l;
    };

    // ---- DMA issue side: tile = 72 pieces of 8 rows x 64 px; wave w issues pieces w, w + 4, ...: its rows advance by 32,
    //      which leaves the chunk rotation ((row >> 1) & 7) of a lane unchanged ----
    int it_b = 0, it_t = stream, it_buf = 0, it_piece = 0;        // tile being issued; pieces of it issued so far (0 .. 18)
    auto norm = [&](int& bb, int& tt) {
        while (tt >= tiles_per_b && bb < p.batch) {
            tt -= tiles_per_b;
            ++bb;
        }
    };
    norm(it_b, it_t);
    const char* it_ptr = nullptr;
    int it_row = 0, it_row_s = 0;           // this lane's row / the first row of the wave's piece (wave-uniform)
    bool it_px_ok = false;
    auto start_operand = [&](bool second) {
        const int l = opaque_lane();
        const int r8 = l >> 3, cpos = l & 7;
        const int row = 8 * wave + r8;                               // first row of this lane inside the operand's block
        const int c = (cpos - ((row >> 1) & 7)) & 7;
        const long long px = (long long)it_t * OS_PN + 8 * c;
        it_px_ok = px < p.P;
        it_row_s = (second ? b0 : a0) + 8 * wave;
        it_row = it_row_s + r8;
        const __hip_bfloat16* base = second ? p.b : p.a;
        it_ptr = reinterpret_cast<const char*>(base + ((long long)it_b * (second ? p.Bn : p.A) + it_row) * p.P + px);
    };
    auto issue_piece = [&]() {              // pieces 0 .. NPA-1 of a wave: operand a, then operand b
        if (it_piece == 0) start_operand(false);
        if (it_piece == NPA) start_operand(true);
        const bool second = it_piece >= NPA;
        const int rows = second ? p.Bn : p.A;
        const int n = second ? (OS_RA / 8) + wave + 4 * (it_piece - NPA) : wave + 4 * it_piece;    // piece of the buffer
        // Every piece is issued (the vmcnt bookkeeping below is static).  Rows past the operand or tiles past the end read
        // any valid address: their accumulators are garbage that no one reads; pixels past the field read zeros.
        const void* src = (it_b < p.batch && !it_px_ok) ? (const void*)p.zeros
                                                        : ((it_b < p.batch && it_row < rows) ? (const void*)it_ptr : (const void*)p.a);
        dma16(src, lds + it_buf * OS_BUF + n * 1024);
        it_ptr += 32 * rowbytes;
        it_row += 32;
        it_row_s += 32;
        if (++it_piece == NPW) {
            it_piece = 0;
            if (++it_buf == NBUF) it_buf = 0;
            it_t += nstreams;
            norm(it_b, it_t);
        }
    };
#pragma unroll 1
    for (int i = 0; i < (NBUF - 1) * NPW; ++i) issue_piece();

    f32x16 acc[3][TB];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < TB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    int b = 0, tb = stream, buf = 0;
    norm(b, tb);
    const uint32_t lds0 = lds_addr(lds);
    while (b < p.batch) {
        wait_vm<NPW * (NBUF - 2)>();   // this tile has landed: all but the pieces of the tiles issued after it ...
        block_sync();                  // ... and everybody's; the buffer of the tile before it is free
        // fragment addresses: row = l % 32 (+ 32 tile), 8 pixels 16 s + 8 (l / 32) = chunk 2 s + l / 32, rotated by row >> 1
        const int l = opaque_lane();
        const int ml = l & 31, h = l >> 5;
        const uint32_t fa = lds0 + buf * OS_BUF + (96 * wa + ml) * 128;
        const uint32_t fb = lds0 + buf * OS_BUF + OS_RA * 128 + (32 * TB * wb + ml) * 128;
        const int rot = (ml >> 1) & 7;
        bf16x8 fr[2][3 + TB];
        auto issue_reads = [&](auto S) {
            constexpr int s = decltype(S)::value;
            const uint32_t off = 16 * ((2 * s + h + rot) & 7);
            fr[s & 1][0] = __builtin_bit_cast(bf16x8, lds_read_b128<0>(fa + off));
            fr[s & 1][1] = __builtin_bit_cast(bf16x8, lds_read_b128<4096>(fa + off));
            fr[s & 1][2] = __builtin_bit_cast(bf16x8, lds_read_b128<8192>(fa + off));
            fr[s & 1][3] = __builtin_bit_cast(bf16x8, lds_read_b128<0>(fb + off));
            fr[s & 1][4] = __builtin_bit_cast(bf16x8, lds_read_b128<4096>(fb + off));
            fr[s & 1][5] = __builtin_bit_cast(bf16x8, lds_read_b128<8192>(fb + off));
            if constexpr (TB == 6) {
                fr[s & 1][6] = __builtin_bit_cast(bf16x8, lds_read_b128<12288>(fb + off));
                fr[s & 1][7] = __builtin_bit_cast(bf16x8, lds_read_b128<16384>(fb + off));
                fr[s & 1][8] = __builtin_bit_cast(bf16x8, lds_read_b128<20480>(fb + off));
            }
        };
        auto step = [&](auto S) {
            constexpr int s = decltype(S)::value;
            if constexpr (s + 1 < 4) {
                issue_reads(std::integral_constant<int, s + 1>{});
                wait_lgkm<3 + TB>();
            } else {
                wait_lgkm<0>();
            }
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < TB; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[s & 1][i], fr[s & 1][3 + j], acc[i][j], 0, 0, 0);
            // the DMA of the tile NBUF - 1 ahead goes out in the shadow of the matrix work
            constexpr int per = (NPW + 3) / 4;
#pragma unroll
            for (int i = s * per; i < (s + 1) * per && i < NPW; ++i) issue_piece();
        };
        issue_reads(std::integral_constant<int, 0>{});
        step(std::integral_constant<int, 0>{});
        step(std::integral_constant<int, 1>{});
        step(std::integral_constant<int, 2>{});
        step(std::integral_constant<int, 3>{});
        if (++buf == NBUF) buf = 0;
        tb += nstreams;
        norm(b, tb);
    }
    wait_vm<0>();      // nothing may be in flight into LDS when the workgroup's LDS is released

    // ---- this workgroup's share of the block: D[ia][ib], lane = ib % 32, registers = ia 8 g + 4 h + j ----
    {
        const int l = opaque_lane();
        const int ml = l & 31, h = l >> 5;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < TB; ++j) {
                const int ib = b0 + 32 * TB * wb + 32 * j + ml;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ia = a0 + 96 * wa + 32 * i + 8 * (r >> 2) + 4 * h + (r & 3);
                    if (ia < p.A && ib < p.Bn) atomicAdd(p.out + ia * p.ld_a + ib * p.ld_b, acc[i][j][r]);
                }
            }
    }
}

}  // namespace

// gw [cout][cin] fp32, accumulated (the caller zeroes it); gy [B][cout][P], x [B][cin][P] bf16, P a multiple of 8.
extern "C" int mk_conv1x1_wgrad_os(const void* gy, const void* x, float* gw, int batch, int cout, int cin, long long P,
                                   void* stream) {
    MK_REQUIRE(gy && x && gw, "null pointer");
    MK_REQUIRE(batch > 0 && cout > 0 && cin > 0 && P > 0, "bad sizes");
    MK_REQUIRE((P % 8) == 0, "P = H*W must be a multiple of 8 (16-byte row alignment)");
    MK_REQUIRE((((uintptr_t)gy | (uintptr_t)x) & 15) == 0, "gy and x must be 16-byte aligned");
    static const char* zeros = [] {
        char* q = nullptr;
        if (hipMalloc(&q, 256) != hipSuccess) return (const char*)nullptr;
        (void)hipMemset(q, 0, 256);
        return (const char*)q;
    }();      // first use must not be inside a stream capture
    MK_REQUIRE(zeros, "cannot allocate the zero block");
    WgosParams p;
    p.zeros = zeros;
    p.out = gw;
    p.batch = batch;
    p.P = P;
    p.tiles_per_b = (P + OS_PN - 1) / OS_PN;
    MK_REQUIRE(p.tiles_per_b * batch < 2147483647LL, "too many pixel tiles");
    static const int ncu = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n > 0 ? n : 256;
    }();
    static const int rb_env = [] { const char* e = getenv("MK_WGRAD_OS_RB"); return e ? atoi(e) : 192; }();
    const int RB = rb_env == 384 ? 384 : 192;
    // the 192-row side of the block goes to the operand it wastes fewer rows on
    auto cost = [&](int ra, int rb) { return (long long)mk::ceil_div(ra, OS_RA) * mk::ceil_div(rb, RB); };
    const bool swap = cost(cin, cout) < cost(cout, cin);
    p.a = (const __hip_bfloat16*)(swap ? x : gy);
    p.b = (const __hip_bfloat16*)(swap ? gy : x);
    p.A = swap ? cin : cout;
    p.Bn = swap ? cout : cin;
    p.ld_a = swap ? 1 : cin;
    p.ld_b = swap ? cin : 1;
    p.nblk_a = mk::ceil_div(p.A, OS_RA);
    p.nblk_b = mk::ceil_div(p.Bn, RB);
    const int unit = 8 * p.nblk_a * p.nblk_b;          // workgroups w and w + 8 (same XCD): same tiles, different blocks
    long long grid = (long long)(ncu / unit) * unit;
    if (grid < unit) grid = unit;
    static const bool once = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_os_kernel<384>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  OsGeom<384>::LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_os_kernel<192>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  OsGeom<192>::LDS);
        return true;
    }();
    (void)once;
    if (RB == 384)
        hipLaunchKernelGGL(wgrad_os_kernel<384>, dim3((unsigned)grid), dim3(OS_THREADS), OsGeom<384>::LDS, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL(wgrad_os_kernel<192>, dim3((unsigned)grid), dim3(OS_THREADS), OsGeom<192>::LDS, (hipStream_t)stream, p);
    MK_LAUNCH_CHECK();
    return 0;
}
