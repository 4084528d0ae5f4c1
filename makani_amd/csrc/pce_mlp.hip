// Fused two-layer pointwise node of the SFNO stack: the MLP / encoder / decoder of makani/models/common/layers.py:86-216
// (call sites sfnonet.py:207,379,463) as ONE launch with the hidden field kept on chip,
//
//     forward  (MODE 0):  pre = A1 x + b1 (kept, bf16);   y = A2 gelu(pre) (+ b2)
//     backward (MODE 1):  gpre = (A1 gy) * gelu'(pre) (kept, bf16; its pixel sums = the fc1 bias gradient);   gx = A2 gpre
//
// with A1 = W1 [Hd x K1], A2 = W2 [M x Hd] forward and A1 = W2^T, A2 = W1^T backward.  The unfused pair of engine launches
// (pce.hip) writes the Hd-row hidden field and reads it back: 3 Hd + K1 + M row passes forward where this kernel moves
// Hd + K1 + M (Hd = 2 K1 = 2 M in the FNO block: 4 instead of 8 field passes; the hidden itself never exists in HBM).
//
// Shape of the work on one CU.  One 256-thread workgroup (ONE wave per SIMD, up to 512 registers each) walks over 128-pixel
// tiles; wave w owns pixel columns 32 w .. 32 w + 31 and ALL rows:
//   * X tile [K1 <= 384][128 px]: fetched once by LDS-DMA (prefetched a tile ahead, two 32 KB regions + one phase parked in
//     registers), pulled into registers with the transposing LDS read and kept as up to 24 MFMA B fragments (96 VGPRs);
//   * the hidden rows go by in CHUNKS of 32: stage 1 computes T[32 hid][32 px] = A1 chunk x X (one accumulator, K1 / 16 MFMAs)
//     with the hidden row on the accumulator's register index and the pixel on the lane -- which IS the operand layout of the
//     next product (guide: "an accumulator tile as the next MFMA's operand"): bias / GELU (or the GELU-gradient multiply) run
//     on the accumulator, the packed bf16 result is used as the A fragments of stage 2, y^T[32 px][M] += H^T chunk x A2 chunk
//     (M / 32 accumulators = 192 VGPRs, resident for the whole tile).  The k order inside such a fragment is permuted
//     (element j of lane half h = chunk row 16 s + 8 (j >> 2) + 4 h + (j & 3)); the packed image of A2 carries the same order;
//   * both weight matrices stream from L2 through a three-buffer LDS-DMA ring in GROUPS (one chunk of A1: K1 / 16 fragments of
//     1 KB; one chunk of A2: 2 M / 32 fragments), two groups in flight while one is multiplied; fragment reads are
//     conflict-free ds_read_b128 issued a few MFMAs ahead of their use;
//   * the GELU of chunk c runs as pure VALU work between the stage-2 MFMAs of chunk c - 1 (the matrix pipe and the vector
//     pipe of a SIMD overlap), so the stream is  G1(0), [G1(c), G2(c - 1)] for c = 1 .. NC - 1, G2(NC - 1);
//   * pre / gpre leave through a wave-private 2 KB LDS tile: written pixel-major as the fragments lie (two ds_write_b128),
//     read back row-major with the transposing read, stored 16 bytes per lane (64-byte row segments, the engine's store shape);
//     y leaves as in pce.hip (accumulator rows on the lanes, 4 consecutive pixels per register quad).
// vmcnt discipline: every global access of the loop is an unconditional wave instruction (masked lanes fetch from a zero block or
// store past a buffer descriptor's range), a wave counts what it issues per iteration and every iteration starts with
// "all but what I issued in the previous iteration has landed" + one barrier: group i (issued in iteration i - 2) is
// visible, the buffer of group i - 1 is free, and a whole iteration of DMA / loads / stores stays in flight.
#include "common.h"
#include "../../include/makani_amd.h"
#include "pce_common.h"

#include <cstdlib>
#include <type_traits>
#include <utility>

namespace {
using namespace pce;

constexpr int QT = 512;             // threads: waves 0-3 = stage-1 waves of pixel groups 0-3, waves 4-7 = their stage-2 partners
constexpr int QPN = 128;            // pixels per tile
constexpr int QXROW = QPN * 2;      // bytes of one k row of the X tile in LDS
constexpr int QSTG = 2048;          // a wave tile
constexpr int QNCMAX = 24;          // hidden chunks (Hd <= 768)
constexpr int QMTMAX = 12;          // output row tiles (M <= 384)
constexpr int QLAG = 2;             // chunks the stage-2 waves run behind the stage-1 waves
#ifndef MK_MLP_ABL                  // timing ablations (wrong results; tools/build_variant.sh): 1 no GELU pairs, 2 no kept-field stores,
#define MK_MLP_ABL 0                // 4 no stage-1 MFMAs, 8 no stage-2 MFMAs, 16 no weight DMA inside the loop, 32 no y stores
#endif
constexpr int QABL = MK_MLP_ABL;

struct MlpParams {
    const __hip_bfloat16* x;        // [B][K1][P]
    const char* wimg;               // mk_pce_mlp_pack image
    const char* zeros;              // 64 zero bytes behind the image
    __hip_bfloat16* y;              // [B][M][P]
    __hip_bfloat16* mid_out;        // [B][Hd][P]: pre (MODE 0) / gpre (MODE 1)
    const __hip_bfloat16* mid_in;   // [B][Hd][P]: pre (MODE 1)
    const float* b1;                // [nb1] floats, read at min(row, nb1 - 1) (MODE 0)
    const float* b2;                // [nb2]
    int nb1, nb2;
    double* rowstats;               // [B][M][2] or null: += (sum, sum of squares) over the pixels of the stored y rows
    double* midsum;                 // [B][Hd] or null: += sum over the pixels of the stored mid_out rows (MODE 1: fc1 bias gradient)
    int M, Hd, K1, B, NC;           // NC = ceil(Hd / 32)
    long long P, tiles_per_b, ntiles;
    int xcd_runs;
    unsigned long long* dbg;        // MK_MLP_STAMPS build: s_memtime stamps of workgroup 0, second tile (8 waves x 64 slots)
};

typedef unsigned int bu4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t q_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7FFFFFFF, 0x00020000);
}
constexpr unsigned Q_OOB = 0x80000000u;
// unconditional 16-byte store: lanes with an offset past the descriptor's range are dropped by the hardware (exact vmcnt
// bookkeeping, no exec-mask branch); nontemporal (the rows are read by a later kernel at the earliest)
__device__ __forceinline__ void q_store16(__amdgpu_buffer_rsrc_t rs, unsigned off, u32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(bu4, v), rs, off, 0, 2);
}

// A [32 hid rows][32 px] bf16 tile as the two MFMA fragments an accumulator turns into (fragment s, lane (px, h), element j = row
// 16 s + 8 (j >> 2) + 4 h + (j & 3)): 16 bytes per lane and fragment; the h = 1 lanes sit four slots further so that the
// transposing reads below meet 64 different banks.
__device__ __forceinline__ uint32_t frag_slot(int px, int h) { return (uint32_t)((h * 32 + ((px + 4 * h) & 31)) * 16); }
// ... read back row-major: lane (i, g) of a transposing read gets hid row 16 * half16 + i, pixels 8 g + 4 sub .. + 3
__device__ __forceinline__ uint32_t frag_tr_addr(int lane, int half16, int sub) {
    const int i = lane & 15, g = lane >> 4;
    const int px = 8 * g + 4 * sub + (i >> 2);
    const int b = 4 * half16 + (i & 3);                 // block of four rows: fragment b >> 2, lane half b & 1, element quad (b >> 1) & 1
    return (uint32_t)((b >> 2) * 1024) + frag_slot(px, b & 1) + (uint32_t)(((b >> 1) & 1) * 8);
}

template <int KS1, int MT, int MODE>
__global__ __launch_bounds__(QT, 2) void pce_mlp_kernel(MlpParams p) {
    constexpr int PHS = KS1 < 6 ? KS1 : 6;             // k16 steps per X phase
    constexpr int NPX = (KS1 + PHS - 1) / PHS;         // X phases: 1, 2 or 4
    constexpr int REGX = PHS * 16 * QXROW;             // bytes of one X region
    constexpr int NREGX = NPX >= 2 ? 2 : 1;
    constexpr int NH1 = (KS1 + 1) / 2;                 // stage-1 fragments of a half iteration (k16 steps)
    constexpr int MTH = (MT + 1) / 2;                  // output row tiles of a half iteration
    constexpr int SLOTF = NH1 + 2 * MTH;               // fragments of a ring slot: stage-1 part, then stage-2 part
    constexpr int SLOTB = SLOTF * 1024;
    constexpr int NFB = 4;                             // weight fragments in flight from LDS
    constexpr int TMIN = 6;                            // half iterations the X prefetch schedule of a tile needs
    static_assert(NPX == 1 || NPX == 2 || NPX == 4, "K1 <= 384");
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    char* XS = lds;                                    // X regions
    char* WB = XS + NREGX * REGX;                      // ring: 3 slots
    char* HX = WB + 3 * SLOTB;                         // 4 tiles: the H fragments of a chunk, stage-1 wave -> stage-2 wave of a pixel group
    char* TB = HX + 4 * QSTG;                          // 8 tiles of the stage-1 waves: MODE 0 pre staging ([pg][0]); MODE 1 pre rows [pg][2]
    char* YS = TB + 8 * QSTG;                          // 4 tiles: y staging of the stage-2 waves
    char* SM = YS + 4 * QSTG;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pg = wave & 3;
    const bool s1 = wave < 4;                          // stage-1 wave; else stage-2 wave
    const int ntiles = (int)p.ntiles, tiles_per_b = (int)p.tiles_per_b;
    const int NC = p.NC, NSLOT = 2 * NC + 2 * QLAG;
    const int TN = NSLOT > TMIN ? NSLOT : TMIN;        // half iterations per tile (even)
    const int TXB = TN - 2;                            // ... the one in which the stage-1 waves take over the next tile's X
    const int wg = (int)blockIdx.x, nwg = (int)gridDim.x;
    const int hh = lane >> 5, ml = lane & 31;

    const uint32_t sm_lds = lds_addr(SM);
    const uint32_t b1_lds = sm_lds;                                           // MODE 0: [QNCMAX * 32] floats, bias of the hidden rows
    const uint32_t ms_lds = sm_lds;                                           // MODE 1: [QNCMAX * 32] floats, sums of the mid rows
    const uint32_t b2_lds = sm_lds + QNCMAX * 32 * 4;                         // [QMTMAX * 32] floats
    const uint32_t rs_lds = b2_lds + QMTMAX * 32 * 4;                         // [QMTMAX * 32][2] floats: sums of the y rows
    const uint32_t hx_lds = lds_addr(HX) + pg * QSTG;
    const uint32_t ys_lds = lds_addr(YS) + pg * QSTG;
    auto tb_lds = [&](int buf) __attribute__((always_inline)) { return lds_addr(TB) + (2 * pg + buf) * QSTG; };

    for (int i = tid; i < NC * 32; i += QT) {
        if constexpr (MODE == 0) lds_write_b32(b1_lds + 4 * i, p.b1[min(i, p.nb1 - 1)]);
        else lds_write_b32(ms_lds + 4 * i, 0.f);
    }
    for (int i = tid; i < MT * 32; i += QT) {
        lds_write_b32(b2_lds + 4 * i, p.b2[min(i, p.nb2 - 1)]);
        lds_write_b32(rs_lds + 8 * i, 0.f);
        lds_write_b32(rs_lds + 8 * i + 4, 0.f);
    }
    wait_lgkm<0>();

    auto phys_tile = [&](int t) __attribute__((always_inline)) {
        const int W = nwg, base = t - wg;
        if (!p.xcd_runs || base + W > ntiles) return t;        // the last, partial window keeps the plain order
        return base + (wg & 7) * (W >> 3) + (wg >> 3);
    };
    auto tile_batch = [&](int work) __attribute__((always_inline)) { return phys_tile(work) / tiles_per_b; };

#ifdef MK_MLP_STAMPS
    int stamp_n = 0;
    bool stamp_on = false;
    auto stamp = [&]() __attribute__((always_inline)) {
        if (stamp_on && stamp_n < 64) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (lane == 0) p.dbg[wave * 64 + stamp_n] = t;
            ++stamp_n;
        }
    };
#else
    auto stamp = [&]() __attribute__((always_inline)) {};
#endif

    // ---- vmcnt bookkeeping: a wave counts the vector-memory operations it issues; "everything up to mark has landed" is
    //      "all but my (ops - mark) youngest operations".  The stage-2 waves' queues hold the weight stream (L2 latency) and the y
    //      stores of a tile's end; the stage-1 waves' everything that goes to / comes from HBM inside the loop ----
    int ops = 0;
    auto wait_for = [&](int mark) __attribute__((always_inline)) { wait_vm_upto(ops - mark); };

    // ---- weight stream (stage-2 waves): slot k of the tile stream -> ring buffer buf; piece q by stage-2 wave q % 4; the
    //      stage-1 part is skipped for k >= 2 NC, the stage-2 part for k < 2 LAG (nobody reads them) ----
    int gmark[3] = {0, 0, 0};
    auto issue_slot = [&](int k, int buf) __attribute__((always_inline)) {
        const char* src = p.wimg + (long long)k * SLOTB + lane * 16 + pg * 1024;
        asm volatile("" : "+v"(src));      // (recomputed per call: as loop invariants the per-piece pointers are hoisted and spilled)
        char* dst = WB + buf * SLOTB + pg * 1024;
        const int q_lo = k < 2 * NC ? 0 : NH1, q_hi = k >= 2 * QLAG ? SLOTF : NH1;
#pragma unroll
        for (int j = 0; j < (SLOTF + 3) / 4; ++j) {
            const int q = pg + 4 * j;
            if (q >= q_lo && q < q_hi) {
                dma16(src + j * 4096, dst + j * 4096);
                ++ops;
            }
        }
        gmark[buf] = ops;
    };

    // ---- X tile (stage-1 waves): region image as in pce.hip (k row r at r * 256 B, its 16 chunks of 8 px rotated by 4 (r & 3)); piece n
    //      = k rows 4 n .. 4 n + 3 of the phase, by stage-1 wave n % 4; every piece is issued by all lanes (zeros outside the field /
    //      past K1) ----
    int xmark = 0;
    auto issue_x = [&](int work, int phase, int region) __attribute__((always_inline)) {
        const int tile = phys_tile(work);
        const int b = tile / tiles_per_b;
        const int x_chunk = ((lane & 15) - 4 * ((lane >> 4) & 3)) & 15;
        const long long n = (long long)(tile - b * tiles_per_b) * QPN + x_chunk * 8;
        const int krow0 = phase * PHS * 16 + (lane >> 4) + 4 * pg;
        const char* src0 = reinterpret_cast<const char*>(p.x + ((long long)b * p.K1 + krow0) * p.P + n);
        char* dst = XS + region * REGX + pg * 1024;
        const bool lane_ok = n < p.P;
        const int steps = (KS1 - PHS * phase) < PHS ? (KS1 - PHS * phase) : PHS;
        const long long sstep = 32 * p.P;               // 16 k rows of bf16
#pragma unroll
        for (int j = 0; j < PHS; ++j)                   // pieces pg + 4 j: k rows 16 j + 4 pg .. + 3 of the phase
            if (j < steps) {
                const void* src = (krow0 + 16 * j < p.K1 && lane_ok) ? (const void*)(src0 + j * sstep) : (const void*)p.zeros;
                dma16(src, dst + j * 4096);
                ++ops;
            }
        xmark = ops;
    };
    // fragments of one phase out of its region: lane (px, h) gets k = 16 s + 8 h + 0..7 of its pixel
    const int rowq = (lane & 15) >> 2;
    const int xch = 4 * pg + 2 * ((lane >> 4) & 1) + ((lane & 3) >> 1);
    const uint32_t xfrag_lane = lds_addr(XS) + (8 * hh + rowq) * QXROW + ((xch + 4 * rowq) & 15) * 16 + (lane & 1) * 8;
    auto read_phase = [&]<int S0, int NS>(bf16x8* dst, int region, std::integer_sequence<int, S0, NS>) __attribute__((always_inline)) {
        const uint32_t a = xfrag_lane + region * REGX;
        u32x2 lo[NS], hi[NS];
        [&]<int... SS>(std::integer_sequence<int, SS...>) {
            ((lo[SS] = lds_read_tr16<SS * 16 * QXROW>(a), hi[SS] = lds_read_tr16<SS * 16 * QXROW + 4 * QXROW>(a)), ...);
        }(std::make_integer_sequence<int, NS>{});
        wait_lgkm<0>();
#pragma unroll
        for (int s = 0; s < NS; ++s) dst[S0 + s] = __builtin_bit_cast(bf16x8, u32x4{lo[s][0], lo[s][1], hi[s][0], hi[s][1]});
    };

    // ---- second input of MODE 1 (stage-1 waves): the pre rows of chunk c of tile `work` for this wave's pixels, [32 hid rows][32 px] =
    //      2 KB, by LDS-DMA straight into one of the wave's two tiles (lane = (row lane / 4 (+ 16), 8 px): the lane-linear DMA image IS
    //      the row-major tile), two chunks ahead of their use ----
    int pmark[2] = {0, 0};
    auto issue_pin = [&](int work, int c, int buf) __attribute__((always_inline)) {
        const int tile = phys_tile(work);
        const int b = tile / tiles_per_b;
        const long long n0 = (long long)(tile - b * tiles_per_b) * QPN + 32 * pg + 8 * (lane & 3);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int row = 32 * c + (lane >> 2) + 16 * q;
            const void* src = (row < p.Hd && n0 < p.P) ? (const void*)(p.mid_in + ((long long)b * p.Hd + row) * p.P + n0)
                                                      : (const void*)p.zeros;
            dma16(src, TB + (2 * pg + buf) * QSTG + q * 1024);
            ++ops;
        }
        pmark[buf] = ops;
    };
    auto issue_pin_ahead = [&](int tile0, int c, int buf) __attribute__((always_inline)) {      // chunk c counted from tile0's chunk 0, whichever tile that falls into
        int t = tile0, j = c;
        while (j >= NC) {
            j -= NC;
            t += nwg;
        }
        if (t < ntiles) issue_pin(t, j, buf);
    };

    // ---- per-row sums (LDS float adds per tile), flushed to the fp64 results per batch item; every wave calls these at the same points ----
    auto flush_y = [&](int b) __attribute__((always_inline)) {
        wait_lgkm<0>();
        block_sync();
        for (int i = tid; i < MT * 32; i += QT) {
            const float a = lds_read_b32(rs_lds + 8 * i), c = lds_read_b32(rs_lds + 8 * i + 4);
            wait_lgkm<0>();
            if (i < p.M) {
                atomicAdd(p.rowstats + ((long long)b * p.M + i) * 2, (double)a);
                atomicAdd(p.rowstats + ((long long)b * p.M + i) * 2 + 1, (double)c);
            }
            lds_write_b32(rs_lds + 8 * i, 0.f);
            lds_write_b32(rs_lds + 8 * i + 4, 0.f);
        }
        wait_lgkm<0>();
        block_sync();
    };
    auto flush_mid = [&](int b) __attribute__((always_inline)) {
        wait_lgkm<0>();
        block_sync();
        for (int i = tid; i < NC * 32; i += QT) {
            const float a = lds_read_b32(ms_lds + 4 * i);
            wait_lgkm<0>();
            if (i < p.Hd) atomicAdd(p.midsum + (long long)b * p.Hd + i, (double)a);
            lds_write_b32(ms_lds + 4 * i, 0.f);
        }
        wait_lgkm<0>();
        block_sync();
    };
    const bool want_y_sums = MODE == 0 && p.rowstats != nullptr, want_mid_sums = MODE == 1 && p.midsum != nullptr;

    // the kept field of a chunk out of a fragment image (frag_slot layout) into HBM, row-major, 16 bytes per lane (64-byte row
    // segments); and its row sums
    auto store_kept = [&](uint32_t img, int b, int c, long long px0) __attribute__((always_inline)) {
        if constexpr (QABL & 2) return;
        u32x2 q[2][2];
#pragma unroll
        for (int h16 = 0; h16 < 2; ++h16)
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) q[h16][sub] = lds_read_tr16<0>(img + frag_tr_addr(lane, h16, sub));
        wait_lgkm<0>();
        const int i = lane & 15, g = lane >> 4;
        const __amdgpu_buffer_rsrc_t rs = q_rsrc(p.mid_out + ((long long)b * p.Hd + 32 * c) * p.P + px0);
        const bool px_ok = px0 + 8 * g < p.P;
#pragma unroll
        for (int h16 = 0; h16 < 2; ++h16) {
            const int row = 16 * h16 + i;
            const bool ok = px_ok && 32 * c + row < p.Hd;
            const u32x4 o = u32x4{q[h16][0][0], q[h16][0][1], q[h16][1][0], q[h16][1][1]};
            q_store16(rs, ok ? (unsigned)(((long long)row * p.P + 8 * g) * 2) : Q_OOB, o);
            ++ops;
            if (want_mid_sums) {
                float sm = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) sm += __uint_as_float(o[e] << 16) + __uint_as_float(o[e] & 0xFFFF0000u);
                if (ok) lds_add_f32(ms_lds + 4 * (32 * c + row), sm);
            }
        }
    };

    if (wg >= ntiles) return;

    // The two roles run the same sequence of barriers on disjoint state: one generic body, instantiated once per role (the
    // register allocator then sees either the X fragments or the y accumulators, never both live in one function path).
    auto run = [&]<bool S1>(std::bool_constant<S1>) __attribute__((always_inline)) {
        struct Stage1State {
            bf16x8 xf[KS1];
            f32x16 acc_h;
            f32x16 ah;                                  // MODE 1: the accumulator of the chunk in the middle (the next chunk's MFMAs reuse acc_h)
            uint32_t pk[8];                             // the chunk in the middle: packed pre, turned pair by pair into its H fragments
        };
        struct Stage2State {
            f32x16 acc_y[MT];                           // y^T accumulators
            bf16x8 hf0, hf1;                            // the H fragments of the chunk being multiplied
        };
        std::conditional_t<S1, Stage1State, Stage2State> st;

        int tile = wg;
        int gbase = 0;          // global half-iteration index of iteration 0 of the current tile (iteration g reads ring buffer g % 3)
        int cc0 = 0;            // parity of the global chunk index of chunk 0 of the current tile (chunk g's pre rows: tile buffer g & 1)

        // ---- prologue: the first tile's X, the first two ring slots, the first pre rows (three barriers, every wave) ----
        if constexpr (S1) {
            issue_x(tile, 0, 0);
            if constexpr (NPX >= 2) issue_x(tile, 1, 1);
            if constexpr (MODE == 1) {
                issue_pin_ahead(tile, 0, 0);
                issue_pin_ahead(tile, 1, 1);
            }
        } else {
            issue_slot(0, 0);
            issue_slot(1, 1);
        }
        wait_vm0();
        block_sync();
        if constexpr (S1) {
            read_phase(st.xf, 0, std::integer_sequence<int, 0, PHS>{});
            if constexpr (NPX >= 2) read_phase(st.xf, 1, std::integer_sequence<int, PHS, (NPX == 2 ? KS1 - PHS : PHS)>{});
        }
        block_sync();
        if constexpr (S1 && NPX == 4) {
            issue_x(tile, 2, 0);
            issue_x(tile, 3, 1);
        }
        wait_vm0();
        block_sync();
        if constexpr (S1) {
            if constexpr (NPX == 4) {
                read_phase(st.xf, 0, std::integer_sequence<int, 2 * PHS, PHS>{});
                read_phase(st.xf, 1, std::integer_sequence<int, 3 * PHS, KS1 - 3 * PHS>{});
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) st.pk[i] = 0;
        } else {
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) st.acc_y[t][r] = 0.f;
        }

        // y rows of output tiles [T0, T1) out (stage-2 waves): accumulator rows m on the lanes, 4 consecutive pixels per register quad
        // (pce.hip); the accumulators are cleared for the next tile
        auto epilogue_part = [&]<int T0, int T1>(auto& S, int b, long long px0, std::integer_sequence<int, T0, T1>) __attribute__((always_inline)) {
            if constexpr (!S1) {
                const int rot = (ml >> 1) & 3;
                const uint32_t st_acc = ys_lds + ml * 64 + hh * 8;
                const int row_lin = lane >> 2, px_lin = (lane & 3) * 8;
                const uint32_t st_lin = ys_lds + row_lin * 64 + (((lane & 3) + (row_lin >> 1)) & 3) * 16;
                // per output tile the DESCRIPTOR moves (scalar arithmetic); the two lane offsets stay (as per-tile lane offsets the
                // compiler hoisted all 2 MT of them out of the tile loop and spilled them)
                const __hip_bfloat16* ybase = p.y + (long long)b * p.M * p.P + px0;
                const unsigned voff0 = (unsigned)(((long long)row_lin * p.P + px_lin) * 2), voff1 = (unsigned)(((long long)(row_lin + 16) * p.P + px_lin) * 2);
                const bool px_ok = px0 + px_lin < p.P;
                int gmask = 0;
#pragma unroll
                for (int gg = 0; gg < 4; ++gg) gmask |= (px0 + 8 * gg < p.P) ? (1 << gg) : 0;
                // (addresses that depend on the tile are base + compile-time offset: as run-time sums the compiler hoists all 2 MT of
                // them out of the tile loop, spills them, and every reload waits for the wave's whole DMA queue)
                const uint32_t b2_a = b2_lds + 4 * ml, rs_a = rs_lds + 8 * ml;
                auto tile_out = [&]<int T>(std::integral_constant<int, T>) __attribute__((always_inline)) {
                    float s1v = 0.f, s2v = 0.f;
                    const float bias_t = lds_read_b32_off<4 * 32 * T>(b2_a);
                    wait_lgkm<0>();
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        u32x2 q;
                        q[0] = pack_bf16x2(S.acc_y[T][4 * g] + bias_t, S.acc_y[T][4 * g + 1] + bias_t);
                        q[1] = pack_bf16x2(S.acc_y[T][4 * g + 2] + bias_t, S.acc_y[T][4 * g + 3] + bias_t);
                        lds_write_b64<0>(st_acc + 16 * ((g + rot) & 3), q);
                        if (want_y_sums && (gmask & (1 << g))) {
                            const float a0 = __uint_as_float(q[0] << 16), a1 = __uint_as_float(q[0] & 0xFFFF0000u);
                            const float a2 = __uint_as_float(q[1] << 16), a3 = __uint_as_float(q[1] & 0xFFFF0000u);
                            s1v += (a0 + a1) + (a2 + a3);
                            s2v = fmaf(a0, a0, fmaf(a1, a1, fmaf(a2, a2, fmaf(a3, a3, s2v))));
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) S.acc_y[T][r] = 0.f;
                    if (want_y_sums) {          // this lane's 16 pixels of row 32 T + ml (the two lane halves add up in LDS)
                        lds_add_f32_off<8 * 32 * T>(rs_a, s1v);
                        lds_add_f32_off<8 * 32 * T + 4>(rs_a, s2v);
                    }
                    wait_lgkm<0>();
                    const u32x4 o0 = lds_read_b128<0>(st_lin), o1 = lds_read_b128<1024>(st_lin);
                    wait_lgkm<0>();
                    const int m0 = 32 * T + row_lin;
                    const __amdgpu_buffer_rsrc_t rs = q_rsrc(ybase + (long long)(32 * T) * p.P);
                    unsigned v0 = voff0, v1 = voff1;
                    asm volatile("" : "+v"(v0), "+v"(v1));      // (opaque per tile: else 2 MT selected offsets are hoisted and spilled)
                    if constexpr (!(QABL & 32)) {
                        q_store16(rs, (px_ok && m0 < p.M) ? v0 : Q_OOB, o0);
                        q_store16(rs, (px_ok && m0 + 16 < p.M) ? v1 : Q_OOB, o1);
                        ops += 2;
                    }
                };
                [&]<int... TT>(std::integer_sequence<int, TT...>) { (tile_out(std::integral_constant<int, T0 + TT>{}), ...); }(
                    std::make_integer_sequence<int, T1 - T0>{});
            }
        };
        // the four epilogue parts of a tile: output tiles [ceil(q MT / 4), ceil((q + 1) MT / 4))
        auto epilogue_quarter = [&]<int Q>(auto& S, std::integral_constant<int, Q>, int b, long long px0) __attribute__((always_inline)) {
            epilogue_part(S, b, px0, std::integer_sequence<int, (Q * MT + 3) / 4, ((Q + 1) * MT + 3) / 4>{});
        };

        int prev_b = -1;                 // batch item of the previous tile (its y rows leave during this tile's first iterations)
        long long prev_px0 = 0;
        for (; tile < ntiles; tile += nwg) {
            const int next_tile = tile + nwg;
            const bool has_next = next_tile < ntiles;
            const int ptile = phys_tile(tile);
            const int b = ptile / tiles_per_b;
            const long long n0 = (long long)(ptile - b * tiles_per_b) * QPN;
            const long long px0 = n0 + 32 * pg;                   // this pixel group's first pixel inside the batch item
#ifdef MK_MLP_STAMPS
            stamp_on = p.dbg && blockIdx.x == 0 && tile == (int)gridDim.x;
#endif
            if (want_mid_sums && prev_b >= 0 && b != prev_b) flush_mid(prev_b);     // before the first middle of the new batch item

            // ---- the element-wise middle of chunk c (stage-1 waves) ----
            // begin: bias and rounding, the kept pre rows out (MODE 0) / the pre rows of the chunk into accumulator layout (MODE 1)
            auto mid_begin = [&](auto& S, int c) __attribute__((always_inline)) {
                if constexpr (S1) {
                    if constexpr (MODE == 0) {
                        u32x4 bq[4];
                        bq[0] = lds_read_b128<0>(b1_lds + (32 * c + 4 * hh) * 4);
                        bq[1] = lds_read_b128<32>(b1_lds + (32 * c + 4 * hh) * 4);
                        bq[2] = lds_read_b128<64>(b1_lds + (32 * c + 4 * hh) * 4);
                        bq[3] = lds_read_b128<96>(b1_lds + (32 * c + 4 * hh) * 4);
                        wait_lgkm<0>();
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const float v0 = S.acc_h[2 * i] + __uint_as_float(bq[i >> 1][(2 * i) & 3]);
                            const float v1 = S.acc_h[2 * i + 1] + __uint_as_float(bq[i >> 1][(2 * i + 1) & 3]);
                            S.pk[i] = pack_bf16x2(v0, v1);
                        }
                        const uint32_t img = tb_lds(0);
                        lds_write_b128<0>(img + frag_slot(ml, hh), u32x4{S.pk[0], S.pk[1], S.pk[2], S.pk[3]});
                        lds_write_b128<1024>(img + frag_slot(ml, hh), u32x4{S.pk[4], S.pk[5], S.pk[6], S.pk[7]});
                        wait_lgkm<0>();
                        store_kept(img, b, c, px0);
                    } else {
                        // the DMA tile [32 hid rows][32 px] (64-byte rows) has landed (waited for before this iteration's barrier).
                        // Lane (px, h) wants rows 8 k + 4 h + 0..3 of its pixel: group g = lane >> 4 covers px half g & 1, h = g >> 1
                        const int buf = (cc0 + c) & 1;
                        const int i = lane & 15, g = lane >> 4;
                        const uint32_t a = tb_lds(buf) + (4 * (g >> 1) + (i >> 2)) * 64 + (2 * (g & 1) + ((i & 3) >> 1)) * 16 + (i & 1) * 8;
                        const u32x2 t0 = lds_read_tr16<0>(a), t1 = lds_read_tr16<8 * 64>(a), t2 = lds_read_tr16<16 * 64>(a),
                                    t3 = lds_read_tr16<24 * 64>(a);
                        wait_lgkm<0>();
                        S.pk[0] = t0[0]; S.pk[1] = t0[1]; S.pk[2] = t1[0]; S.pk[3] = t1[1];
                        S.pk[4] = t2[0]; S.pk[5] = t2[1]; S.pk[6] = t3[0]; S.pk[7] = t3[1];
                        S.ah = S.acc_h;
                        issue_pin_ahead(tile, c + 2, buf);             // the buffer is free again: the pre rows of the chunk after next
                    }
                }
            };
            auto mid_pair = [&](auto& S, int i) __attribute__((always_inline)) {      // rows 2 i, 2 i + 1 of the lane's 16
                if constexpr (S1 && !(QABL & 1)) {
                    const float lo = __uint_as_float(S.pk[i] << 16), hi = __uint_as_float(S.pk[i] & 0xFFFF0000u);
                    if constexpr (MODE == 0) S.pk[i] = pack_bf16x2(gelu_f(lo), gelu_f(hi));
                    else S.pk[i] = pack_bf16x2(S.ah[2 * i] * gelu_grad_f(lo), S.ah[2 * i + 1] * gelu_grad_f(hi));
                }
            };
            // end: the H fragments for the partner wave; MODE 1: they are the kept field (gpre), out they go
            auto mid_end = [&](auto& S, int c) __attribute__((always_inline)) {
                if constexpr (S1) {
                    lds_write_b128<0>(hx_lds + frag_slot(ml, hh), u32x4{S.pk[0], S.pk[1], S.pk[2], S.pk[3]});
                    lds_write_b128<1024>(hx_lds + frag_slot(ml, hh), u32x4{S.pk[4], S.pk[5], S.pk[6], S.pk[7]});
                    wait_lgkm<0>();
                    if constexpr (MODE == 1) store_kept(hx_lds, b, c, px0);
                }
            };

            // ---- stage 1, one half: acc_h += A1 chunk half x X, with four GELU pairs of the previous chunk between the MFMAs ----
            auto stage1_half = [&]<int HALF>(auto& S, int buf, bool with_mid, std::integral_constant<int, HALF>) __attribute__((always_inline)) {
                if constexpr (S1) {
                    constexpr int S0 = HALF * NH1, NS = (HALF == 0) ? NH1 : KS1 - NH1;
                    const uint32_t a = lds_addr(WB) + buf * SLOTB + lane * 16;
                    bf16x8 wf[NFB];
                    if constexpr (HALF == 0) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) S.acc_h[r] = 0.f;
                    }
                    [&]<int... J>(std::integer_sequence<int, J...>) {
                        ((J < NFB ? (void)(wf[J % NFB] = lds_read_frag<J * 1024>(a)) : (void)0), ...);
                        ((wait_lgkm<(NS - 1 - J) < (NFB - 1) ? (NS - 1 - J) : (NFB - 1)>(),
                          ((QABL & 4) ? (void)0 : (void)(S.acc_h = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[J % NFB], S.xf[S0 + J], S.acc_h, 0, 0, 0))),     // D[hid][px]
                          (J + NFB < NS ? (void)(wf[J % NFB] = lds_read_frag<(J + NFB < NS ? J + NFB : 0) * 1024>(a)) : (void)0),
                          [&] {
                              if (with_mid) {
#pragma unroll
                                  for (int i = 4 * HALF + J * 4 / NS; i < 4 * HALF + (J + 1) * 4 / NS; ++i) mid_pair(S, i);
                              }
                          }()), ...);
                    }(std::make_integer_sequence<int, NS>{});
                }
            };
            // ---- stage 2, one half: acc_y[t] += H^T chunk x A2 chunk for the half's output tiles ----
            auto stage2_half = [&]<int HALF>(auto& S, int buf, std::integral_constant<int, HALF>) __attribute__((always_inline)) {
                constexpr int T0 = HALF * MTH, NT = (HALF == 0) ? MTH : MT - MTH, NF = 2 * NT;
                if constexpr (!S1 && NT > 0) {
                    constexpr int NFB = 3;        // (the accumulators leave this role few registers; 2, 3, 4 in flight measured alike)
                    const uint32_t a = lds_addr(WB) + buf * SLOTB + NH1 * 1024 + lane * 16;
                    bf16x8 wf[NFB];
                    [&]<int... J>(std::integer_sequence<int, J...>) {
                        ((J < NFB ? (void)(wf[J % NFB] = lds_read_frag<J * 1024>(a)) : (void)0), ...);
                        ((wait_lgkm<(NF - 1 - J) < (NFB - 1) ? (NF - 1 - J) : (NFB - 1)>(),
                          ((QABL & 8) ? (void)0 : (void)(S.acc_y[T0 + (J >> 1)] =
                              __builtin_amdgcn_mfma_f32_32x32x16_bf16((J & 1) ? S.hf1 : S.hf0, wf[J % NFB], S.acc_y[T0 + (J >> 1)], 0, 0, 0))),   // D[px][m]
                          (J + NFB < NF ? (void)(wf[J % NFB] = lds_read_frag<(J + NFB < NF ? J + NFB : 0) * 1024>(a)) : (void)0)), ...);
                    }(std::make_integer_sequence<int, NF>{});
                }
            };

            // one half iteration; QUARTER >= 0: one of the first 2 LAG, in which the stage-2 waves send off that quarter of the previous
            // tile's y rows (a compile-time quarter: selecting the accumulators by a run-time index inside the loop cost 500 spilled registers)
            // ... PARITY = t & 1 at compile time: the two halves of a chunk touch different accumulators, and a run-time choice between
            // them inside the loop makes the register allocator keep old and new copies of the tiles apart (100+ spilled registers)
            auto iteration = [&]<int QUARTER, int PARITY>(int t, std::integral_constant<int, QUARTER>, std::integral_constant<int, PARITY>) __attribute__((always_inline)) {
                const int buf = (gbase + t) % 3;
                stamp();   // A: iteration entry
                // ---- what this wave has to see landed before the barrier ----
                if constexpr (!S1) {
                    if (t < NSLOT) wait_for(gmark[buf]);                      // my pieces of this iteration's slot
                } else {
                    if constexpr (MODE == 1) {
                        if (PARITY == 1 && t < 2 * NC) wait_for(pmark[(cc0 + (t >> 1)) & 1]);      // the pre rows of the chunk whose middle begins here
                    }
                    if (has_next && t == TXB) wait_for(xmark);                 // X phases about to be read
                    if (NPX == 4 && t == 0 && prev_b >= 0) wait_for(xmark);    // ... and the second pair of a long X, issued at the last tile's end
                }
                stamp();   // B: my waits done
                block_sync();
                stamp();   // C: barrier passed
                if constexpr (!S1) {
                    // ---- weight stream: the slot two iterations ahead (of this tile, or of the next one) ----
                    const int k = t + 2;
                    if constexpr (!(QABL & 16)) {
                        if (k < TN) {
                            if (k < NSLOT) issue_slot(k, (gbase + k) % 3);
                        } else if (has_next) {
                            issue_slot(k - TN, (gbase + k) % 3);
                        }
                    }
                    // ---- the previous tile's y rows, a quarter per iteration; then this tile's stage 2, two chunks behind stage 1 ----
                    if constexpr (QUARTER >= 0) {
                        if (prev_b >= 0) epilogue_quarter(st, std::integral_constant<int, QUARTER>{}, prev_b, prev_px0);
                    } else if (t < NSLOT) {
                        if constexpr (PARITY == 0) {
                            st.hf0 = lds_read_frag<0>(hx_lds + frag_slot(ml, hh));
                            st.hf1 = lds_read_frag<1024>(hx_lds + frag_slot(ml, hh));
                            wait_lgkm<0>();
                        }
                        stage2_half(st, buf, std::integral_constant<int, PARITY>{});
                    }
                } else {
                    // ---- X of the next tile ----
                    if constexpr (NPX == 4) {
                        // phases 2, 3 of THIS tile (issued when the previous tile's stage 1 was done) complete the X fragments
                        if (t == 0 && prev_b >= 0) {
                            read_phase(st.xf, 0, std::integer_sequence<int, 2 * PHS, PHS>{});
                            read_phase(st.xf, 1, std::integer_sequence<int, 3 * PHS, KS1 - 3 * PHS>{});
                        }
                    }
                    if (has_next) {
                        if (t == 1) issue_x(next_tile, 0, 0);                    // (behind the barrier that follows the reads above)
                        if (NPX >= 2 && t == 2) issue_x(next_tile, 1, 1);
                    }
                    // ---- stage 1 of chunk t / 2 with the GELU pairs of the chunk before it; the middle's begin / end around it ----
                    const int c = t >> 1;
                    if (t < 2 * NC) {
                        stage1_half(st, buf, c > 0, std::integral_constant<int, PARITY>{});
                        if constexpr (PARITY == 1) {
                            if (c > 0) mid_end(st, c - 1);
                            mid_begin(st, c);
                        }
                    } else if (t < 2 * NC + 2) {                  // the last chunk's pairs have no MFMAs left to hide behind
#pragma unroll
                        for (int i = 4 * PARITY; i < 4 * PARITY + 4; ++i) mid_pair(st, i);
                        if constexpr (PARITY == 1) mid_end(st, NC - 1);
                    }
                    if (has_next && t == TXB) {                   // this wave's X fragments of the next tile (stage 1 of this tile is done)
                        read_phase(st.xf, 0, std::integer_sequence<int, 0, PHS>{});
                        if constexpr (NPX >= 2) read_phase(st.xf, 1, std::integer_sequence<int, PHS, (NPX == 2 ? KS1 - PHS : PHS)>{});
                    }
                    if constexpr (NPX == 4) {                     // ... and behind the next barrier the regions take its phases 2, 3
                        if (has_next && t == TXB + 1) {
                            issue_x(next_tile, 2, 0);
                            issue_x(next_tile, 3, 1);
                        }
                    }
                }
                // the previous tile's y rows are out: their row sums, if its batch item ends with it
                if (want_y_sums && t == 2 * QLAG - 1 && prev_b >= 0 && prev_b != b) flush_y(prev_b);
                stamp();   // D: work done
            };
            static_assert(QLAG == 2 && TMIN >= 4, "the four peeled iterations");
            iteration(0, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
            iteration(1, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
            iteration(2, std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{});
            iteration(3, std::integral_constant<int, 3>{}, std::integral_constant<int, 1>{});
            for (int t = 4; t < TN; t += 2) {         // TN is even
                iteration(t, std::integral_constant<int, -1>{}, std::integral_constant<int, 0>{});
                iteration(t + 1, std::integral_constant<int, -1>{}, std::integral_constant<int, 1>{});
            }
            gbase = (gbase + TN) % 3;
            cc0 = (cc0 + NC) & 1;
            prev_b = b;
            prev_px0 = px0;
        }
        // ---- the last tile's y rows and sums ----
        if constexpr (!S1) {
            wait_vm0();
            epilogue_quarter(st, std::integral_constant<int, 0>{}, prev_b, prev_px0);
            epilogue_quarter(st, std::integral_constant<int, 1>{}, prev_b, prev_px0);
            epilogue_quarter(st, std::integral_constant<int, 2>{}, prev_b, prev_px0);
            epilogue_quarter(st, std::integral_constant<int, 3>{}, prev_b, prev_px0);
        }
        if (want_y_sums) flush_y(prev_b);
        if (want_mid_sums) flush_mid(prev_b);
    };
    if (s1) run(std::true_type{});
    else run(std::false_type{});
    wait_vm0();      // nothing may be in flight into LDS when the workgroup's LDS is released
}

// ---- weight image: ring slot t of a tile's stream = [stage-1 part: NH1 fragments | stage-2 part: 2 MTH fragments], t = 0 .. 2 NC + 2 LAG - 1
//   stage-1 part (t < 2 NC):   chunk t / 2, k16 steps (t & 1) NH1 + f:        lane (r, h), element j  <-  A1[32 c + r][16 s + 8 h + j]
//   stage-2 part (t >= 2 LAG): chunk t / 2 - LAG, row tile (t & 1) MTH + f / 2, half-chunk f & 1:
//                                                  lane (r, h), element j  <-  A2[32 tt + r][32 c + 16 s' + 8 (j >> 2) + 4 h + (j & 3)]
// (the second is the k order of an accumulator tile used as an MFMA operand); 32 zero elements at the end.
template <typename T>
__global__ void pce_mlp_pack_kernel(const T* __restrict__ a1, int t1, int lda1, const T* __restrict__ a2, int t2, int lda2, int M,
                                    int Hd, int K1, int NC, int KS1, int MT, unsigned short* __restrict__ img, long long core,
                                    long long total) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    if (idx >= core) {
        img[idx] = 0;
        return;
    }
    const int NH1 = (KS1 + 1) / 2, MTH = (MT + 1) / 2, SLOTF = NH1 + 2 * MTH;
    const int j = (int)(idx & 7);
    const int lane = (int)((idx >> 3) & 63);
    const int r = lane & 31, h = lane >> 5;
    const long long q = idx >> 9;           // fragment index in the stream
    const int t = (int)(q / SLOTF), f = (int)(q % SLOTF);
    float v = 0.f;
    if (f < NH1) {
        const int c = t >> 1, s = (t & 1) * NH1 + f;
        const int hid = 32 * c + r, k = 16 * s + 8 * h + j;
        if (t < 2 * NC && s < KS1 && hid < Hd && k < K1) v = (float)(t1 ? a1[(long long)k * lda1 + hid] : a1[(long long)hid * lda1 + k]);
    } else {
        const int f2 = f - NH1;
        const int c = (t >> 1) - QLAG, tt = (t & 1) * MTH + (f2 >> 1);
        const int m = 32 * tt + r, hid = 32 * c + 16 * (f2 & 1) + 8 * (j >> 2) + 4 * h + (j & 3);
        if (t >= 2 * QLAG && tt < MT && m < M && hid < Hd) v = (float)(t2 ? a2[(long long)hid * lda2 + m] : a2[(long long)m * lda2 + hid]);
    }
    img[idx] = f32_to_bf16_bits(v);
}

struct MlpCfg {
    int KS1, MT, NC;
};
static bool mlp_config(int M, int Hd, int K1, MlpCfg* c) {
    if (M <= 0 || Hd <= 0 || K1 <= 0 || K1 > 384 || Hd > 32 * QNCMAX || M > 32 * QMTMAX) return false;
    c->KS1 = K1 <= 80 ? 5 : (K1 <= 192 ? 12 : 24);
    c->MT = M <= 96 ? 3 : 12;
    c->NC = mk::ceil_div(Hd, 32);
    return true;
}
static long long mlp_image_core_bytes(const MlpCfg& c) {
    return (long long)(2 * c.NC + 2 * QLAG) * ((c.KS1 + 1) / 2 + 2 * ((c.MT + 1) / 2)) * 1024;
}

static const float* mlp_zero_bias() {
    static float* z = [] {
        float* q = nullptr;
        if (hipMalloc(&q, 4096) != hipSuccess) return (float*)nullptr;
        (void)hipMemset(q, 0, 4096);
        return q;
    }();
    return z;
}

static unsigned long long* mlp_dbg_buffer() {
    static unsigned long long* d = [] {
        unsigned long long* q = nullptr;
        if (getenv("MK_MLP_DBG")) {
            if (hipMalloc(&q, 8 * 64 * 8) != hipSuccess) return (unsigned long long*)nullptr;
            (void)hipMemset(q, 0, 8 * 64 * 8);
        }
        return q;
    }();
    return d;
}

template <int KS1, int MT, int MODE>
static void mlp_launch(const MlpParams& p, hipStream_t st) {
    constexpr int PHS = KS1 < 6 ? KS1 : 6, NPX = (KS1 + PHS - 1) / PHS, NREGX = NPX >= 2 ? 2 : 1;
    constexpr int SLOTF = (KS1 + 1) / 2 + 2 * ((MT + 1) / 2);
    constexpr int LDS = NREGX * PHS * 16 * QXROW + 3 * SLOTF * 1024 + 16 * QSTG + (QNCMAX + 3 * QMTMAX) * 32 * 4;
    static_assert(LDS <= 160 * 1024, "LDS budget");
    static const bool once = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pce_mlp_kernel<KS1, MT, MODE>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        return true;
    }();
    (void)once;
    static const int ncu = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n > 0 ? n : 256;
    }();
    const long long grid = p.ntiles < ncu ? p.ntiles : ncu;
    MlpParams q = p;
    q.xcd_runs = grid % 8 == 0 && (p.P * 2) % 128 != 0;      // tile order by XCD runs where rows are not whole 128-byte lines (pce.hip)
    hipLaunchKernelGGL((pce_mlp_kernel<KS1, MT, MODE>), dim3((unsigned)grid), dim3(QT), LDS, st, q);
}

}  // namespace

extern "C" long long mk_pce_mlp_image_bytes(int M, int Hd, int K1) {
    MlpCfg c;
    if (!mlp_config(M, Hd, K1, &c)) return 0;
    return mlp_image_core_bytes(c) + 64;
}

extern "C" int mk_pce_mlp_pack(const void* a1, int a1_transposed, int lda1, const void* a2, int a2_transposed, int lda2,
                               int w_dtype, int M, int Hd, int K1, void* img, void* stream) {
    MK_REQUIRE(a1 && a2 && img, "null pointer");
    MK_REQUIRE(w_dtype == 0 || w_dtype == 1, "w_dtype must be 0 (fp32) or 1 (bf16)");
    MlpCfg c;
    MK_REQUIRE(mlp_config(M, Hd, K1, &c), "unsupported shape (K1 <= 384, Hd <= 768, M <= 384)");
    MK_REQUIRE(lda1 >= (a1_transposed ? Hd : K1) && lda2 >= (a2_transposed ? M : Hd), "leading dimension too small");
    const long long total = (mlp_image_core_bytes(c) + 64) / 2;
    const long long nblk = (total + 255) / 256;
    if (w_dtype == 0)
        hipLaunchKernelGGL(pce_mlp_pack_kernel<float>, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, (const float*)a1,
                           a1_transposed, lda1, (const float*)a2, a2_transposed, lda2, M, Hd, K1, c.NC, c.KS1, c.MT,
                           (unsigned short*)img, total - 32, total);
    else
        hipLaunchKernelGGL(pce_mlp_pack_kernel<__hip_bfloat16>, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream,
                           (const __hip_bfloat16*)a1, a1_transposed, lda1, (const __hip_bfloat16*)a2, a2_transposed, lda2, M, Hd,
                           K1, c.NC, c.KS1, c.MT, (unsigned short*)img, total - 32, total);
    MK_LAUNCH_CHECK();
    return 0;
}

namespace {
__global__ void mlp_zero_kernel(double* p, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0.0;
}
}  // namespace

extern "C" int mk_pce_mlp(const void* x, const void* wimg, void* y, void* mid_out, const void* mid_in, const float* b1,
                          const float* b2, double* rowstats_y, double* rowsum_mid, int mode, int batch, int M, int Hd, int K1,
                          long long P, void* stream) {
    MK_REQUIRE(x && wimg && y && mid_out, "null pointer");
    MK_REQUIRE(mode == 0 || mode == 1, "mode must be 0 (forward: bias + GELU) or 1 (backward: GELU-gradient multiply)");
    MK_REQUIRE(mode == 0 || mid_in, "mode 1 needs the kept pre-activation (mid_in)");
    MK_REQUIRE(mode == 1 || !rowsum_mid, "row sums of the kept field are built for mode 1");
    MK_REQUIRE(batch > 0 && P > 0, "bad sizes");
    MK_REQUIRE((P % 8) == 0, "P = H*W must be a multiple of 8 (16-byte row alignment)");
    MK_REQUIRE((((uintptr_t)x | (uintptr_t)wimg | (uintptr_t)y | (uintptr_t)mid_out | (uintptr_t)mid_in) & 15) == 0,
               "fields and the weight image must be 16-byte aligned");
    MlpCfg c;
    MK_REQUIRE(mlp_config(M, Hd, K1, &c), "unsupported shape (K1 <= 384, Hd <= 768, M <= 384)");
    MK_REQUIRE((long long)(M > 32 ? M : 32) * P * 2 < (1LL << 31), "field too large for the 32-bit store offsets (M * P * 2 bytes per batch item)");
    hipStream_t st = (hipStream_t)stream;
    if (rowstats_y) {
        const long long n = 2LL * batch * M;
        hipLaunchKernelGGL(mlp_zero_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, rowstats_y, n);
    }
    if (rowsum_mid) {
        const long long n = (long long)batch * Hd;
        hipLaunchKernelGGL(mlp_zero_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, rowsum_mid, n);
    }
    const float* zb = (b1 && b2) ? nullptr : mlp_zero_bias();
    MK_REQUIRE((b1 && b2) || zb, "cannot allocate the zero bias");
    const long long tiles_per_b = (P + QPN - 1) / QPN;
    MK_REQUIRE(tiles_per_b * batch < 2147483647LL, "too many pixel tiles");
    MlpParams p;
    p.x = (const __hip_bfloat16*)x;
    p.wimg = (const char*)wimg;
    p.zeros = (const char*)wimg + mlp_image_core_bytes(c);
    p.y = (__hip_bfloat16*)y;
    p.mid_out = (__hip_bfloat16*)mid_out;
    p.mid_in = (const __hip_bfloat16*)mid_in;
    p.b1 = b1 ? b1 : zb;
    p.nb1 = b1 ? Hd : 1024;
    p.b2 = b2 ? b2 : zb;
    p.nb2 = b2 ? M : 1024;
    p.rowstats = rowstats_y;
    p.midsum = rowsum_mid;
    p.M = M;
    p.Hd = Hd;
    p.K1 = K1;
    p.B = batch;
    p.NC = c.NC;
    p.P = P;
    p.tiles_per_b = tiles_per_b;
    p.ntiles = tiles_per_b * batch;
    p.xcd_runs = 0;
    p.dbg = mlp_dbg_buffer();
    bool done = false;
#define MK_MLP_CASE(KS1_, MT_)                                        \
    if (!done && c.KS1 == KS1_ && c.MT == MT_) {                      \
        if (mode == 0) mlp_launch<KS1_, MT_, 0>(p, st);               \
        else mlp_launch<KS1_, MT_, 1>(p, st);                         \
        done = true;                                                  \
    }
    MK_MLP_CASE(5, 3) MK_MLP_CASE(5, 12) MK_MLP_CASE(12, 3) MK_MLP_CASE(12, 12) MK_MLP_CASE(24, 3) MK_MLP_CASE(24, 12)
#undef MK_MLP_CASE
    MK_REQUIRE(done, "no kernel instance for this shape");
    MK_LAUNCH_CHECK();
    return 0;
}

// debug: s_memtime stamps of workgroup 0, second tile, of the last launch (MK_MLP_DBG=1, build with -DMK_MLP_STAMPS): 4 waves x 128
extern "C" int mk_pce_mlp_debug_stamps(unsigned long long* out512) {
    MK_REQUIRE(out512, "null pointer");
    unsigned long long* d = mlp_dbg_buffer();
    MK_REQUIRE(d, "MK_MLP_DBG is not set");
    MK_REQUIRE(hipMemcpy(out512, d, 8 * 64 * 8, hipMemcpyDeviceToHost) == hipSuccess, "copy failed");
    return 0;
}
