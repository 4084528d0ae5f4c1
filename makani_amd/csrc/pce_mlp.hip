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
#include <utility>

namespace {
using namespace pce;

constexpr int QT = 256;             // threads: 4 waves = 4 pixel groups of 32
constexpr int QPN = 128;            // pixels per tile
constexpr int QXROW = QPN * 2;      // bytes of one k row of the X tile in LDS
constexpr int QSTG = 2048;          // wave-private staging tile
constexpr int QNCMAX = 24;          // hidden chunks (Hd <= 768)
constexpr int QMTMAX = 12;          // output row tiles (M <= 384)

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
    unsigned long long* dbg;        // MK_MLP_STAMPS build: s_memtime stamps of workgroup 0, second tile (4 waves x 128 slots)
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

// chunk rotation of the pixel-major staging image (rows = pixels, 64 B each = 4 chunks of 8 values): makes both the
// ds_write_b128 of 8 consecutive pixels and the transposing reads of 4 + 4 pixel rows hit distinct banks
__device__ __forceinline__ int q_rot(int px) { return ((px >> 1) & 1) | ((((px >> 2) ^ (px >> 3)) & 1) << 1); }

// group k of a tile's weight stream: byte offset in the image and number of 1 KB fragments
template <int NF1, int NF2>
__device__ __forceinline__ void q_group(int k, int NC, int& off_kb, int& nf) {
    if (k == 0) {
        off_kb = 0;
        nf = NF1;
    } else if (k & 1) {
        const int c = (k + 1) >> 1;                      // G1(c), or G2(NC - 1) for the last group
        off_kb = c * NF1 + (c - 1) * NF2;
        nf = (k == 2 * NC - 1) ? NF2 : NF1;
    } else {
        off_kb = ((k >> 1) + 1) * NF1 + ((k >> 1) - 1) * NF2;     // G2(k / 2 - 1)
        nf = NF2;
    }
}

template <int KS1, int MT, int MODE>
__global__ __launch_bounds__(QT, 1) void pce_mlp_kernel(MlpParams p) {
    constexpr int NPH = (KS1 + 7) / 8;                 // X phases (LDS regions) of up to 8 k16 steps
    constexpr int KSP = KS1 < 8 ? KS1 : 8;
    constexpr int REGB = KSP * 16 * QXROW;             // bytes of one X region
    constexpr int NREG = NPH >= 2 ? 2 : 1;
    constexpr int NF1 = KS1, NF2 = 2 * MT;
    constexpr int NFMAX = NF1 > NF2 ? NF1 : NF2;
    constexpr int GROUPB = NFMAX * 1024;
#ifndef MK_MLP_NFB
#define MK_MLP_NFB 4
#endif
    constexpr int NFB = MK_MLP_NFB;                    // weight fragments in flight from LDS
    static_assert(NPH <= 3, "K1 <= 384");
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    char* XS = lds;
    char* WB = lds + NREG * REGB;
    char* TB = WB + 3 * GROUPB;                        // 8 wave tiles of 2 KB: [wave][2 buffers] (see below)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntiles = (int)p.ntiles, tiles_per_b = (int)p.tiles_per_b;
    const int NC = p.NC, NIT = 2 * NC;
    const int wg = (int)blockIdx.x, nwg = (int)gridDim.x;
    const int hh = lane >> 5, ml = lane & 31;
    // Roles.  vmcnt retires in issue order, so a wave that waits for an L2-latency weight group must not have HBM-latency
    // operations (X pieces, the stores of the kept field, the pre rows of the backward pass) queued in front of it:
    // waves 0-2 issue the weight stream and nothing else inside the loop; wave 3 issues everything that goes to / comes
    // from HBM, for all four waves (the kept field is handed over through the wave tiles in LDS).
    const bool dma_wave = wave < 3, hbm_wave = wave == 3;

    const uint32_t wb_a = lds_addr(WB) + lane * 16;                          // this lane's piece of a weight fragment
    // wave tiles: buffer (global chunk index & 1) of wave w holds, in turn, the pre rows of the chunk (MODE 1, by LDS-DMA),
    // then the kept field of the chunk on its way to wave 3's stores; both free again two chunks later
    auto tile_lds = [&](int w, int buf) { return lds_addr(TB) + (2 * w + buf) * QSTG; };
    const uint32_t sm_lds = lds_addr(TB) + 8 * QSTG;
    const uint32_t b1_lds = sm_lds;                                           // MODE 0: [QNCMAX * 32] floats, bias of the hidden rows
    const uint32_t ms_lds = sm_lds;                                           // MODE 1: [QNCMAX * 32] floats, sums of the mid rows
    const uint32_t b2_lds = sm_lds + QNCMAX * 32 * 4;                         // [QMTMAX * 32] floats
    const uint32_t rs_lds = b2_lds + QMTMAX * 32 * 4;                         // [QMTMAX * 32][2] floats: sums of the y rows

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

    auto phys_tile = [&](int t) {
        const int W = nwg, base = t - wg;
        if (!p.xcd_runs || base + W > ntiles) return t;        // the last, partial window keeps the plain order
        return base + (wg & 7) * (W >> 3) + (wg >> 3);
    };

#ifdef MK_MLP_STAMPS
    int stamp_n = 0;
    bool stamp_on = false;
    auto stamp = [&]() {
        if (stamp_on && stamp_n < 128) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (lane == 0) p.dbg[wave * 128 + stamp_n] = t;
            ++stamp_n;
        }
    };
#else
    auto stamp = [&]() {};
#endif
    // ---- vmcnt bookkeeping: a wave counts the vector-memory operations it issues; "everything up to mark has landed" is
    //      "all but my (ops - mark) youngest operations" ----
    int ops = 0;
    auto wait_for = [&](int mark) { wait_vm_upto(ops - mark); };

    // ---- DMA issue.  What an iteration has to issue is set up (`pend_*`) and issued at its start, in front of the stage.  (Issuing
    //      the pieces between the MFMAs of the stage instead did not hide them -- a piece blocks the wave's instruction stream for
    //      ~90 cycles, three MFMA slots, and with one wave per SIMD the matrix pipe idles meanwhile: stage 1 944 -> 1672 cycles,
    //      profiles/r03_mlp_stamps.txt -- and it is unsafe next to hand-counted lgkmcnt waits: the scalar loads the compiler emits
    //      for the issue count in lgkmcnt too, so a fragment read was taken for finished one read early.)
    //      waves 0-2: piece q of a weight group by wave q % 3;   wave 3: the X pieces of the next tile (region image as in pce.hip:
    //      k row r at r * 256 B, its 16 chunks of 8 px rotated by 4 (r & 3); piece n = k rows 4 n .. 4 n + 3 of the phase; every piece
    //      is issued by all lanes, zeros outside the field / past K1) ----
    int gmark[3] = {0, 0, 0};
    int xmark = 0;
    const char* pend_src = nullptr;     // this lane's source of slot 0
    char* pend_dst = nullptr;           // LDS destination of slot 0 (wave-uniform)
    long long pend_sstep = 0;           // source step per slot (bytes)
    int pend_n = 0;                     // pieces this wave issues in this iteration
    int pend_buf = -1;                  // ring buffer the group goes to (waves 0-2)
    int pend_krow = 0;                  // wave 3: this lane's k row of slot 0
    bool pend_ok = false;               // wave 3: this lane's pixels are inside the field
    auto setup_group = [&](int k, int buf) {        // group k of the tile stream -> ring buffer buf
        int off_kb, nf;
        q_group<NF1, NF2>(k, NC, off_kb, nf);
        pend_src = p.wimg + (long long)off_kb * 1024 + lane * 16 + wave * 1024;
        pend_dst = WB + buf * GROUPB + wave * 1024;
        pend_sstep = 3072;
        pend_n = nf > wave ? (nf - wave + 2) / 3 : 0;
        pend_buf = buf;
    };
    auto setup_x = [&](int work, int phase, int region, int xg) {      // pieces 8 xg .. 8 xg + 7 of the phase
        const int tile = phys_tile(work);
        const int b = tile / tiles_per_b;
        const int x_chunk = ((lane & 15) - 4 * ((lane >> 4) & 3)) & 15;
        const long long n = (long long)(tile - b * tiles_per_b) * QPN + x_chunk * 8;
        pend_krow = phase * 128 + (lane >> 4) + 32 * xg;
        pend_src = reinterpret_cast<const char*>(p.x + ((long long)b * p.K1 + pend_krow) * p.P + n);
        pend_dst = XS + region * REGB + xg * 8192;
        pend_sstep = 8 * p.P;                           // 4 k rows of bf16
        pend_ok = n < p.P;
        const int steps = (KS1 - 8 * phase) < 8 ? (KS1 - 8 * phase) : 8;
        const int left = 4 * steps - 8 * xg;
        pend_n = left < 0 ? 0 : (left > 8 ? 8 : left);
    };
    auto issue_slot = [&](int j) {
        if (j >= pend_n) return;
        if (dma_wave) {
            dma16(pend_src + j * 3072, pend_dst + j * 3072);
        } else {
            const void* src = (pend_krow + 4 * j < p.K1 && pend_ok) ? (const void*)(pend_src + j * pend_sstep) : (const void*)p.zeros;
            dma16(src, pend_dst + j * 1024);
        }
        ++ops;
    };
    auto finish_issue = [&](int first) {        // the slots the stage had no room for; then the marks
        for (int j = first; j < pend_n; ++j) issue_slot(j);
        if (pend_n > 0) {
            if (dma_wave) gmark[pend_buf] = ops;
            else xmark = ops;
        }
        pend_n = 0;
    };
    auto issue_group = [&](int k, int buf) {        // prologue: all at once
        setup_group(k, buf);
        finish_issue(0);
    };
    auto issue_x = [&](int work, int phase, int region, int xg) {
        setup_x(work, phase, region, xg);
        finish_issue(0);
    };
    // fragments of one phase out of its region: lane (px, h) gets k = 16 s + 8 h + 0..7 of its pixel
    const int rowq = (lane & 15) >> 2;
    const int xch = 4 * wave + 2 * ((lane >> 4) & 1) + ((lane & 3) >> 1);
    const uint32_t xfrag_lane = lds_addr(XS) + (8 * hh + rowq) * QXROW + ((xch + 4 * rowq) & 15) * 16 + (lane & 1) * 8;
    auto read_phase = [&]<int S0, int NS>(bf16x8* dst, int region, std::integer_sequence<int, S0, NS>) {
        const uint32_t a = xfrag_lane + region * REGB;
        u32x2 lo[NS], hi[NS];
        [&]<int... SS>(std::integer_sequence<int, SS...>) {
            ((lo[SS] = lds_read_tr16<SS * 16 * QXROW>(a), hi[SS] = lds_read_tr16<SS * 16 * QXROW + 4 * QXROW>(a)), ...);
        }(std::make_integer_sequence<int, NS>{});
        wait_lgkm<0>();
#pragma unroll
        for (int s = 0; s < NS; ++s) dst[S0 + s] = __builtin_bit_cast(bf16x8, u32x4{lo[s][0], lo[s][1], hi[s][0], hi[s][1]});
    };
    constexpr int NS0 = KS1 < 8 ? KS1 : 8, NS1 = NPH >= 2 ? (KS1 - 8 < 8 ? KS1 - 8 : 8) : 0, NS2 = NPH >= 3 ? KS1 - 16 : 0;

    bf16x8 xf[KS1];
    bf16x8 xnext[NPH == 3 ? 8 : 1];        // NPH == 3: phase 0 of the next tile, parked while its region takes phase 2

    // ---- second input of MODE 1 (wave 3): the pre rows of chunk j of tile `work`, [32 hid rows][32 px] = 2 KB per wave, by
    //      LDS-DMA straight into the wave tiles (lane = (row lane / 4 (+ 16), 8 px): the lane-linear DMA image IS the
    //      row-major tile), several iterations ahead of their use ----
    int pmark[2] = {0, 0};
    auto issue_pin = [&](int work, int c, int buf) {
        const int tile = phys_tile(work);
        const int b = tile / tiles_per_b;
        const long long nb = (long long)(tile - b * tiles_per_b) * QPN + 8 * (lane & 3);
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int row = 32 * c + (lane >> 2) + 16 * q;
                const long long n0 = nb + 32 * w;
                const void* src = (row < p.Hd && n0 < p.P) ? (const void*)(p.mid_in + ((long long)b * p.Hd + row) * p.P + n0)
                                                          : (const void*)p.zeros;
                dma16(src, TB + (2 * w + buf) * QSTG + q * 1024);
                ++ops;
            }
        pmark[buf] = ops;
    };
    // the pre rows of (this tile's chunk c) + ahead, whichever tile that falls into
    auto issue_pin_ahead = [&](int tile0, int c, int buf) {
        int t = tile0, j = c;
        while (j >= NC) {
            j -= NC;
            t += nwg;
        }
        if (t < ntiles) issue_pin(t, j, buf);
    };

    // ---- per-row sums (y rows and mid rows: LDS float adds per tile), flushed to the fp64 results per batch item ----
    int rs_b = -1;
    auto flush_sums = [&]() {            // every wave calls this at the same points
        if ((!p.rowstats && !p.midsum) || rs_b < 0) return;
        wait_lgkm<0>();
        block_sync();
        if (MODE == 0 && p.rowstats)
            for (int i = tid; i < MT * 32; i += QT) {
                const float a = lds_read_b32(rs_lds + 8 * i), b = lds_read_b32(rs_lds + 8 * i + 4);
                wait_lgkm<0>();
                if (i < p.M) {
                    atomicAdd(p.rowstats + ((long long)rs_b * p.M + i) * 2, (double)a);
                    atomicAdd(p.rowstats + ((long long)rs_b * p.M + i) * 2 + 1, (double)b);
                }
                lds_write_b32(rs_lds + 8 * i, 0.f);
                lds_write_b32(rs_lds + 8 * i + 4, 0.f);
            }
        if (MODE == 1 && p.midsum)
            for (int i = tid; i < NC * 32; i += QT) {
                const float a = lds_read_b32(ms_lds + 4 * i);
                wait_lgkm<0>();
                if (i < p.Hd) atomicAdd(p.midsum + (long long)rs_b * p.Hd + i, (double)a);
                lds_write_b32(ms_lds + 4 * i, 0.f);
            }
        wait_lgkm<0>();
        block_sync();
    };

    int tile = wg;
    if (tile >= ntiles) return;
    int gbase = 0;          // stream index of group 0 of the current tile (group g of the stream lives in buffer g % 3)
    int cc0 = 0;            // global chunk index of chunk 0 of the current tile (chunk g uses wave-tile buffer g & 1)

    // ---- prologue: the first tile's X phases, its first two weight groups, the pre rows of its first two chunks ----
    {
        if constexpr (NPH == 3) {
            if (hbm_wave)
                for (int xg = 0; xg < 4; ++xg) issue_x(tile, 0, 0, xg);
            wait_vm0();
            block_sync();
            read_phase(xnext, 0, std::integer_sequence<int, 0, 8>{});
            block_sync();
            if (hbm_wave) {
                for (int xg = 0; xg < 4; ++xg) issue_x(tile, 1, 1, xg);
                for (int xg = 0; xg < 4; ++xg) issue_x(tile, 2, 0, xg);
            }
        } else if (hbm_wave) {
            for (int xg = 0; xg < 4; ++xg) issue_x(tile, 0, 0, xg);
            if constexpr (NPH == 2)
                for (int xg = 0; xg < 4; ++xg) issue_x(tile, 1, 1, xg);
        }
        if (dma_wave) {
            issue_group(0, 0);
            if (NIT > 1) issue_group(1, 1);
        }
        if constexpr (MODE == 1) {
            if (hbm_wave) {
                issue_pin_ahead(tile, 0, 0);
                issue_pin_ahead(tile, 1, 1);
            }
        }
        wait_vm0();
        block_sync();
    }

    for (; tile < ntiles; tile += nwg) {
        const int next_tile = tile + nwg;
        const bool has_next = next_tile < ntiles;
        const int ptile = phys_tile(tile);
        const int b = ptile / tiles_per_b;
        const long long n0 = (long long)(ptile - b * tiles_per_b) * QPN;
        const long long px0 = n0 + 32 * wave;                 // this wave's first pixel inside the batch item
#ifdef MK_MLP_STAMPS
        stamp_on = p.dbg && blockIdx.x == 0 && tile == (int)gridDim.x;
#endif
        stamp();   // tile start
        if ((p.rowstats || p.midsum) && b != rs_b) {
            flush_sums();
            rs_b = b;
        }

        // ---- tile start: everything of this tile's X (and weight groups 0, 1) has landed and is visible (wait + barrier at
        //      the end of the previous tile / of the prologue) ----
        if constexpr (NPH == 3) {
#pragma unroll
            for (int s = 0; s < 8; ++s) xf[s] = xnext[s];
            read_phase(xf, 1, std::integer_sequence<int, 8, NS1>{});
            read_phase(xf, 0, std::integer_sequence<int, 16, NS2>{});
        } else {
            read_phase(xf, 0, std::integer_sequence<int, 0, NS0>{});
            if constexpr (NPH == 2) read_phase(xf, 1, std::integer_sequence<int, 8, NS1>{});
        }
        // (the barrier of iteration 0 orders these reads before the first DMA into the regions)

        f32x16 acc_y[MT];
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc_y[t][r] = 0.f;
        f32x16 acc_h;
        uint32_t hcur[8];                   // packed bf16 pairs: H fragments of the chunk in stage 2
        uint32_t pk[8];                     // the chunk in the middle: packed pre (MODE 0: computed; MODE 1: read back), turned pair by
                                            // pair into its H fragments (MODE 0: gelu(pre); MODE 1: gpre)
#pragma unroll
        for (int i = 0; i < 8; ++i) hcur[i] = pk[i] = 0;

        // X prefetch of the next tile (issued by wave 3), one step per iteration
        constexpr int XSTEPS = NPH == 3 ? 17 : 4 * NPH;
        auto xstep_wait = [&](int k) {      // before the barrier of iteration k
            if constexpr (NPH == 3) {
                if (has_next && k == 12 && hbm_wave) wait_for(xmark);          // phase 0 (and 1) of the next tile: long landed
            }
        };
        auto xstep = [&](int k) {           // after the barrier of iteration k
            if (!has_next) return;
            if constexpr (NPH == 3) {
                if (k == 12) read_phase(xnext, 0, std::integer_sequence<int, 0, 8>{});      // all waves; region 0 is free behind the next barrier
            }
            if (!hbm_wave) return;
            if (k < 4) setup_x(next_tile, 0, 0, k);
            else if (NPH >= 2 && k < 8) setup_x(next_tile, 1, 1, k - 4);
            else if (NPH == 3 && k >= 13 && k < 17) setup_x(next_tile, 2, 0, k - 13);
        };

        // the kept field of chunk c out of the wave tiles into HBM (wave 3, for all four waves), and its row sums.  Lane (i, g): column
        // i of the 16 a transposing read delivers, pixel block g; rows (i & 3) + 8 (i >> 2) + 4 pr of the chunk (pr = lane half they
        // came from), 8 pixels 8 g .. 8 g + 7 = two reads of 4 pixel rows each
        auto store_chunk = [&](int c) {
            const int buf = (cc0 + c) & 1;
            const int i = lane & 15, g = lane >> 4;
            uint32_t soff[2][2];
#pragma unroll
            for (int pr = 0; pr < 2; ++pr)
#pragma unroll
                for (int sub = 0; sub < 2; ++sub) {
                    const int row = 8 * g + 4 * sub + (i >> 2);
                    soff[pr][sub] = row * 64 + (((2 * pr + ((i & 3) >> 1)) ^ q_rot(row)) * 16) + (i & 1) * 8;
                }
            const int crow = (i & 3) + 8 * (i >> 2);
            const unsigned voff0 = (unsigned)(((long long)crow * p.P + 8 * g) * 2), voff1 = (unsigned)(((long long)(crow + 4) * p.P + 8 * g) * 2);
            const bool r0 = 32 * c + crow < p.Hd, r1 = 32 * c + crow + 4 < p.Hd;
            const __hip_bfloat16* base = p.mid_out + ((long long)b * p.Hd + 32 * c) * p.P + n0;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                u32x2 q[2][2][2];
#pragma unroll
                for (int w2 = 0; w2 < 2; ++w2) {
                    const uint32_t tl = tile_lds(2 * half + w2, buf);
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr)
#pragma unroll
                        for (int sub = 0; sub < 2; ++sub) q[w2][pr][sub] = lds_read_tr16<0>(tl + soff[pr][sub]);
                }
                wait_lgkm<0>();
#pragma unroll
                for (int w2 = 0; w2 < 2; ++w2) {
                    const int w = 2 * half + w2;
                    const __amdgpu_buffer_rsrc_t rs = q_rsrc(base + 32 * w);
                    const bool px_ok = n0 + 32 * w + 8 * g < p.P;
                    const u32x4 o0 = u32x4{q[w2][0][0][0], q[w2][0][0][1], q[w2][0][1][0], q[w2][0][1][1]};
                    const u32x4 o1 = u32x4{q[w2][1][0][0], q[w2][1][0][1], q[w2][1][1][0], q[w2][1][1][1]};
                    q_store16(rs, (px_ok && r0) ? voff0 : Q_OOB, o0);
                    q_store16(rs, (px_ok && r1) ? voff1 : Q_OOB, o1);
                    ops += 2;
                    if (MODE == 1 && p.midsum) {
                        float s0 = 0.f, s1 = 0.f;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            s0 += __uint_as_float(o0[e] << 16) + __uint_as_float(o0[e] & 0xFFFF0000u);
                            s1 += __uint_as_float(o1[e] << 16) + __uint_as_float(o1[e] & 0xFFFF0000u);
                        }
                        if (px_ok && r0) lds_add_f32(ms_lds + 4 * (32 * c + crow), s0);
                        if (px_ok && r1) lds_add_f32(ms_lds + 4 * (32 * c + crow + 4), s1);
                    }
                }
            }
        };

        auto iter_begin = [&](int it) {
            stamp();   // A: iteration entry
            if (dma_wave) wait_for(gmark[(gbase + it) % 3]);       // my pieces of group `it` have landed
            if constexpr (MODE == 1) {
                if (hbm_wave && !(it & 1)) wait_for(pmark[(cc0 + (it >> 1)) & 1]);     // the pre rows of the chunk whose middle runs now
            }
            xstep_wait(it);
            stamp();   // B: my waits done
            block_sync();                  // ... and are visible to all waves; nobody reads group it - 1 any more
            stamp();   // C: barrier passed
            if (dma_wave) {
                const int k = it + 2;
                if (k < NIT) setup_group(k, (gbase + k) % 3);
                else if (has_next && k - NIT < NIT) setup_group(k - NIT, (gbase + k) % 3);
            }
            if (it & 1) {                  // the middle of chunk (it - 1) / 2 ended in the previous iteration
                const int c = it >> 1;
                if (hbm_wave) {
                    store_chunk(c);
                    if constexpr (MODE == 1) issue_pin_ahead(tile, c + 2, (cc0 + c) & 1);     // into the buffer just read
                }
            }
            xstep(it);
            finish_issue(0);
            stamp();   // D: issues done
        };

        // ---- stage 1: acc_h = A1 chunk (fragments of the group in `buf`) x X ----
        auto stage1 = [&](int buf) {
            const uint32_t a = wb_a + buf * GROUPB;
            bf16x8 wf[NFB];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc_h[r] = 0.f;
            [&]<int... J>(std::integer_sequence<int, J...>) {
                ((J < NFB ? (void)(wf[J % NFB] = lds_read_frag<J * 1024>(a)) : (void)0), ...);
                ((wait_lgkm<(NF1 - 1 - J) < (NFB - 1) ? (NF1 - 1 - J) : (NFB - 1)>(),
                  acc_h = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[J % NFB], xf[J], acc_h, 0, 0, 0),     // D[hid][px]
                  (J + NFB < NF1 ? (void)(wf[J % NFB] = lds_read_frag<(J + NFB < NF1 ? J + NFB : 0) * 1024>(a)) : (void)0)), ...);
            }(std::make_integer_sequence<int, NF1>{});
        };

        // ---- the element-wise middle: accumulator -> packed bf16 H fragments ----
        // begin: bias and rounding (MODE 0) / the pre rows of the chunk into accumulator layout (MODE 1)
        auto mid_begin = [&](int c) {
            if constexpr (MODE == 0) {
                u32x4 bq[4];
                bq[0] = lds_read_b128<0>(b1_lds + (32 * c + 4 * hh) * 4);
                bq[1] = lds_read_b128<32>(b1_lds + (32 * c + 4 * hh) * 4);
                bq[2] = lds_read_b128<64>(b1_lds + (32 * c + 4 * hh) * 4);
                bq[3] = lds_read_b128<96>(b1_lds + (32 * c + 4 * hh) * 4);
                wait_lgkm<0>();
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float v0 = acc_h[2 * i] + __uint_as_float(bq[i >> 1][(2 * i) & 3]);
                    const float v1 = acc_h[2 * i + 1] + __uint_as_float(bq[i >> 1][(2 * i + 1) & 3]);
                    pk[i] = pack_bf16x2(v0, v1);
                }
            } else {
                // the DMA tile [32 hid rows][32 px] (64-byte rows) has landed (wave 3 waited for it before this iteration's
                // barrier).  Lane (px, h) wants rows 8 k + 4 h + 0..3 of its pixel: group g = lane >> 4 covers px half g & 1, h = g >> 1
                const int i = lane & 15, g = lane >> 4;
                const uint32_t a = tile_lds(wave, (cc0 + c) & 1) + (4 * (g >> 1) + (i >> 2)) * 64 + (2 * (g & 1) + ((i & 3) >> 1)) * 16 +
                                   (i & 1) * 8;
                const u32x2 t0 = lds_read_tr16<0>(a), t1 = lds_read_tr16<8 * 64>(a), t2 = lds_read_tr16<16 * 64>(a),
                            t3 = lds_read_tr16<24 * 64>(a);
                wait_lgkm<0>();
                pk[0] = t0[0]; pk[1] = t0[1]; pk[2] = t1[0]; pk[3] = t1[1];
                pk[4] = t2[0]; pk[5] = t2[1]; pk[6] = t3[0]; pk[7] = t3[1];
            }
        };
        // one packed pair: rows 2 i, 2 i + 1 of the lane's 16
        auto mid_pair = [&](int i) {
            const float lo = __uint_as_float(pk[i] << 16), hi = __uint_as_float(pk[i] & 0xFFFF0000u);
            if constexpr (MODE == 0) pk[i] = pack_bf16x2(gelu_f(lo), gelu_f(hi));
            else pk[i] = pack_bf16x2(acc_h[2 * i] * gelu_grad_f(lo), acc_h[2 * i + 1] * gelu_grad_f(hi));
        };
        // the kept field (pre / gpre) pixel-major, as the fragments lie, into this wave's tile (wave 3 stores it row-major behind
        // the next barrier): MODE 0 right after the rounding, MODE 1 after the last pair
        auto keep_chunk = [&](int c) {
            const int rot = q_rot(ml);
            const uint32_t tl = tile_lds(wave, (cc0 + c) & 1);
            lds_write_b128<0>(tl + ml * 64 + (((2 * hh) ^ rot) * 16), u32x4{pk[0], pk[1], pk[2], pk[3]});
            lds_write_b128<0>(tl + ml * 64 + (((2 * hh + 1) ^ rot) * 16), u32x4{pk[4], pk[5], pk[6], pk[7]});
            wait_lgkm<0>();
        };
        auto mid_end = [&](int c) {
            if constexpr (MODE == 1) keep_chunk(c);
        };

        // ---- stage 2: acc_y[t] += H^T chunk x A2 chunk, with the GELU pairs of the NEXT chunk between the MFMAs ----
        auto stage2 = [&](int buf, bool with_mid) {
            const uint32_t a = wb_a + buf * GROUPB;
            bf16x8 wf[NFB];
            const bf16x8 h0 = __builtin_bit_cast(bf16x8, u32x4{hcur[0], hcur[1], hcur[2], hcur[3]});
            const bf16x8 h1 = __builtin_bit_cast(bf16x8, u32x4{hcur[4], hcur[5], hcur[6], hcur[7]});
            [&]<int... J>(std::integer_sequence<int, J...>) {
                ((J < NFB ? (void)(wf[J % NFB] = lds_read_frag<J * 1024>(a)) : (void)0), ...);
                ((wait_lgkm<(NF2 - 1 - J) < (NFB - 1) ? (NF2 - 1 - J) : (NFB - 1)>(),
                  acc_y[J >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16((J & 1) ? h1 : h0, wf[J % NFB], acc_y[J >> 1], 0, 0, 0),   // D[px][m]
                  (J + NFB < NF2 ? (void)(wf[J % NFB] = lds_read_frag<(J + NFB < NF2 ? J + NFB : 0) * 1024>(a)) : (void)0),
                  [&] {
                      if (with_mid) {         // the middle of the next chunk, four values (two packed pairs) at a time
                          constexpr int Q0 = J * 4 / NF2, Q1 = (J + 1) * 4 / NF2;
#pragma unroll
                          for (int qd = Q0; qd < Q1; ++qd) {
                              mid_pair(2 * qd);
                              mid_pair(2 * qd + 1);
                          }
                      }
                  }()), ...);
            }(std::make_integer_sequence<int, NF2>{});
        };

        // ---- the tile's iterations ----
        iter_begin(0);
        stage1(gbase % 3);
        mid_begin(0);
        if constexpr (MODE == 0) keep_chunk(0);
#pragma unroll
        for (int i = 0; i < 8; ++i) mid_pair(i);
        mid_end(0);
#pragma unroll
        for (int i = 0; i < 8; ++i) hcur[i] = pk[i];
        for (int c = 1; c < NC; ++c) {
            iter_begin(2 * c - 1);
            stage1((gbase + 2 * c - 1) % 3);
            iter_begin(2 * c);
            mid_begin(c);
            if constexpr (MODE == 0) keep_chunk(c);
            stage2((gbase + 2 * c) % 3, true);
            mid_end(c);
#pragma unroll
            for (int i = 0; i < 8; ++i) hcur[i] = pk[i];
        }
        iter_begin(NIT - 1);
        stage2((gbase + NIT - 1) % 3, false);
        // X steps the iterations did not reach (few hidden chunks)
        for (int k = NIT; k < XSTEPS; ++k) {
            if (!has_next) break;
            wait_vm0();
            block_sync();
            xstep(k);
            finish_issue(0);
        }
        gbase = (gbase + NIT) % 3;
        cc0 = (cc0 + NC) & 1;

        // ---- epilogue: y rows out (accumulator rows m on the lanes, 4 consecutive pixels per register quad; pce.hip) ----
        stamp();   // loop end
        wait_vm0();           // the next tile's X phases / first groups / pre rows have landed; no store queues behind a DMA
        block_sync();
        {
            // staging tile: the ring buffer of this tile's last group is free (groups 0, 1 of the next tile sit in the other
            // two; the wave tiles may hold pre rows parked for the next tile)
            const uint32_t stg = lds_addr(WB) + ((gbase + 2) % 3) * GROUPB + wave * QSTG;
            const int rot = (ml >> 1) & 3;
            const uint32_t st_acc = stg + ml * 64 + hh * 8;
            const int row_lin = lane >> 2, px_lin = (lane & 3) * 8;
            const uint32_t st_lin = stg + row_lin * 64 + (((lane & 3) + (row_lin >> 1)) & 3) * 16;
            const __amdgpu_buffer_rsrc_t rs = q_rsrc(p.y + (long long)b * p.M * p.P + px0);
            const bool px_ok = px0 + px_lin < p.P;
            int gmask = 0;
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) gmask |= (px0 + 8 * gg < p.P) ? (1 << gg) : 0;
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                float v[16];
                float s1 = 0.f, s2 = 0.f;
                const float bias_t = lds_read_b32(b2_lds + 4 * (32 * t + ml));
                wait_lgkm<0>();
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = acc_y[t][r] + bias_t;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    u32x2 q;
                    q[0] = pack_bf16x2(v[4 * g], v[4 * g + 1]);
                    q[1] = pack_bf16x2(v[4 * g + 2], v[4 * g + 3]);
                    lds_write_b64<0>(st_acc + 16 * ((g + rot) & 3), q);
                    if (MODE == 0 && p.rowstats && (gmask & (1 << g))) {
                        const float a0 = __uint_as_float(q[0] << 16), a1 = __uint_as_float(q[0] & 0xFFFF0000u);
                        const float a2 = __uint_as_float(q[1] << 16), a3 = __uint_as_float(q[1] & 0xFFFF0000u);
                        s1 += (a0 + a1) + (a2 + a3);
                        s2 = fmaf(a0, a0, fmaf(a1, a1, fmaf(a2, a2, fmaf(a3, a3, s2))));
                    }
                }
                if (MODE == 0 && p.rowstats) {          // this lane's 16 pixels of row 32 t + ml (the two lane halves add up in LDS)
                    lds_add_f32(rs_lds + 8 * (32 * t + ml), s1);
                    lds_add_f32(rs_lds + 8 * (32 * t + ml) + 4, s2);
                }
                wait_lgkm<0>();
                const u32x4 o0 = lds_read_b128<0>(st_lin), o1 = lds_read_b128<1024>(st_lin);
                wait_lgkm<0>();
                const int m0 = 32 * t + row_lin;
                q_store16(rs, (px_ok && m0 < p.M) ? (unsigned)(((long long)m0 * p.P + px_lin) * 2) : Q_OOB, o0);
                q_store16(rs, (px_ok && m0 + 16 < p.M) ? (unsigned)(((long long)(m0 + 16) * p.P + px_lin) * 2) : Q_OOB, o1);
                ops += 2;
            }
        }
    }
    flush_sums();
    wait_vm0();      // nothing may be in flight into LDS when the workgroup's LDS is released
}

// ---- weight image: the tile stream G1(0), [G1(c), G2(c - 1)] c = 1 .. NC - 1, G2(NC - 1) as 1 KB MFMA fragments ------------
//   G1(c) fragment s  (k16 step):          lane (r, h), element j  <-  A1[32 c + r][16 s + 8 h + j]
//   G2(c) fragment 2 t + s' (row tile t):  lane (r, h), element j  <-  A2[32 t + r][32 c + 16 s' + 8 (j >> 2) + 4 h + (j & 3)]
// (the second is the k order of an accumulator tile used as an MFMA operand); 32 zero elements at the end.
template <typename T>
__global__ void pce_mlp_pack_kernel(const T* __restrict__ a1, int t1, int lda1, const T* __restrict__ a2, int t2, int lda2, int M,
                                    int Hd, int K1, int NC, int NF1, int NF2, unsigned short* __restrict__ img, long long core,
                                    long long total) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    if (idx >= core) {
        img[idx] = 0;
        return;
    }
    const int j = (int)(idx & 7);
    const int lane = (int)((idx >> 3) & 63);
    const int r = lane & 31, h = lane >> 5;
    long long q = idx >> 9;                 // fragment index in the stream
    int type, c, f;
    if (q < NF1) {
        type = 1; c = 0; f = (int)q;
    } else {
        q -= NF1;
        const int blk = (int)(q / (NF1 + NF2)), rr = (int)(q % (NF1 + NF2));
        if (blk < NC - 1) {
            if (rr < NF1) { type = 1; c = blk + 1; f = rr; }
            else { type = 2; c = blk; f = rr - NF1; }
        } else {
            type = 2; c = NC - 1; f = rr;
        }
    }
    float v = 0.f;
    if (type == 1) {
        const int hid = 32 * c + r, k = 16 * f + 8 * h + j;
        if (hid < Hd && k < K1) v = (float)(t1 ? a1[(long long)k * lda1 + hid] : a1[(long long)hid * lda1 + k]);
    } else {
        const int m = 32 * (f >> 1) + r, hid = 32 * c + 16 * (f & 1) + 8 * (j >> 2) + 4 * h + (j & 3);
        if (m < M && hid < Hd) v = (float)(t2 ? a2[(long long)hid * lda2 + m] : a2[(long long)m * lda2 + hid]);
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
static long long mlp_image_core_bytes(const MlpCfg& c) { return (long long)c.NC * (c.KS1 + 2 * c.MT) * 1024; }

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
            if (hipMalloc(&q, 4 * 128 * 8) != hipSuccess) return (unsigned long long*)nullptr;
            (void)hipMemset(q, 0, 4 * 128 * 8);
        }
        return q;
    }();
    return d;
}

template <int KS1, int MT, int MODE>
static void mlp_launch(const MlpParams& p, hipStream_t st) {
    constexpr int NPH = (KS1 + 7) / 8, KSP = KS1 < 8 ? KS1 : 8, NREG = NPH >= 2 ? 2 : 1;
    constexpr int NFMAX = KS1 > 2 * MT ? KS1 : 2 * MT;
    constexpr int LDS = NREG * KSP * 16 * QXROW + 3 * NFMAX * 1024 + 8 * QSTG + (QNCMAX + 3 * QMTMAX) * 32 * 4;
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
                           a1_transposed, lda1, (const float*)a2, a2_transposed, lda2, M, Hd, K1, c.NC, c.KS1, 2 * c.MT,
                           (unsigned short*)img, total - 32, total);
    else
        hipLaunchKernelGGL(pce_mlp_pack_kernel<__hip_bfloat16>, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream,
                           (const __hip_bfloat16*)a1, a1_transposed, lda1, (const __hip_bfloat16*)a2, a2_transposed, lda2, M, Hd,
                           K1, c.NC, c.KS1, 2 * c.MT, (unsigned short*)img, total - 32, total);
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
    MK_REQUIRE(hipMemcpy(out512, d, 4 * 128 * 8, hipMemcpyDeviceToHost) == hipSuccess, "copy failed");
    return 0;
}
