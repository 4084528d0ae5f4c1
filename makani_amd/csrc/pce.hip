// Pixel-column engine: the 1x1 convolutions of the SFNO pointwise stack (encoder, MLP, skips, decoder;
// makani/models/common/layers.py:86-216, sfnonet.py:239-267) as bf16 MFMA GEMMs on NCHW fields viewed as
// [C][P = H*W], with the bias / GELU / skip-add / GELU-gradient passes folded into the epilogue.
//
//     Y[b][m][p] = epi( sum_k W[m][k] * X[b][k][p] )
//
// Shape of the problem on MI355X: M, K are 73..768 channels, P is 1e5..1e6 pixels, so every one of these GEMMs
// sits at or below the bf16 ridge (intensity M*K/(M+K) <= 256 flop per HBM byte): the job is to read X once, write Y
// once and keep everything else on chip.  One persistent 512-thread workgroup per CU walks over 128-pixel tiles:
//
//   * the X tile [K][128 px] is fetched ONCE by LDS-DMA (global_load_lds_dwordx4, waves 6-7, issued a whole tile
//     ahead) and each wave pulls its 32 pixel columns out of it into registers as MFMA B fragments with the
//     transposing LDS read (ds_read_b64_tr_b16) -- the activations then stay in registers for the whole tile;
//   * the weights are pre-packed once per call (mk_pce_pack) into the exact LDS image of the A fragments
//     (1 KB per 32 rows x 16 k fragment, lane-linear) and streamed from L2 through a two-slot LDS-DMA ring by
//     waves 0-5, one slot (all rows x 32 k) ahead of the MFMAs; a fragment read is one conflict-free ds_read_b128;
//   * wave (pg, mh) owns pixel columns 32*pg.. and the output rows of half mh: 6 accumulators of 32x32 (96 VGPRs) +
//     24 resident B fragments (96 VGPRs) -> 2 waves per SIMD, which cover each other's LDS / DMA waits;
//   * the two DMA streams are issued by DIFFERENT waves because vmcnt retires in issue order: a wave that waits
//     for its next weight slot must not have the (much longer) HBM fetch of the next X tile queued in front of it.
//
// K <= 384 per phase; 384 < K <= 768 runs as two K phases into the same accumulators (the second half of the X
// tile is fetched while the first half is being multiplied).  M > 384 runs as passes of 384 rows over the same
// resident X fragments.
#include "common.h"
#include "../../include/makani_amd.h"
#include "pce_common.h"

#include <hip/hip_bf16.h>
#include <cstdint>
#include <cstdlib>
#include <type_traits>
#include <utility>

namespace {
using namespace pce;

constexpr int PT = 512;         // threads: 8 waves = 4 pixel groups x 2 row halves
constexpr int PN = 128;         // pixels per tile
constexpr int XROW = PN * 2;    // bytes of one k row of the X tile in LDS
constexpr int PCE_KPHASE = 384; // k rows resident per phase

#ifdef MK_PCE_ABLATE     // profiling build only (tools/pce_ablate.py): run-time ablation checks cost the hot loops
#define PCE_EXP(bit) (p.exp & (bit))
#else
#define PCE_EXP(bit) false
#endif

struct PceParams {
    const __hip_bfloat16* x;       // [B][K][P]
    const char* wimg;              // mk_pce_pack image
    const char* zeros;             // 64 zero bytes behind the image (source of masked DMA lanes)
    __hip_bfloat16* y;             // [B][M][P]
    const float* bias;             // [nbias] floats, read at min(row, nbias - 1)
    int nbias;
    unsigned bias_lds;             // streaming kernel: LDS byte address of the workgroup's copy of the bias (set by the kernel)
    const __hip_bfloat16* addend;  // [B][M][P] or null: y += addend
    const float* aff;              // [B][M][2] or null: the addend enters as aff[..][0] * addend + aff[..][1] (a per-row affine map:
                                   // the instance norm of the addend, applied here instead of in a pass of its own)
    const __hip_bfloat16* aux_in;  // [B][M][P] or null: y *= gelu'(aux_in)   (applied before the addend)
    __hip_bfloat16* aux_out;       // [B][M][P] or null: pre-activation (acc + bias) stored here
    double* rowstats;              // [B][M][2] or null: += (sum, sum of squares) over the pixels of the stored y rows
    int gelu;                      // y = gelu(acc + bias)
    int M, K, B;                   // M: output rows of the whole field
    int npass;                     // passes of 64 TH rows over the same X tile (the second pass finds it in L2)
    long long img_per_pass;        // bytes of one pass of the weight image
    long long P, tiles_per_b, ntiles;
    unsigned long long* dbg;       // MK_PCE_DBG: s_memtime stamps of workgroup 0 (8 waves x 64 slots), else null
    int exp;                       // MK_PCE_EXP ablations (wrong results): 1 no epilogue, 2 no MFMA, 4 no weight DMA, 8 no X DMA
    int nt;                        // output rows leave with nontemporal stores (streaming kernel only)
    int xcd_runs;                  // tile order: each XCD takes a contiguous run of the 256-tile window (see phys_tile)
    int split;                     // streaming kernel, two passes: the passes of a tile go to a PAIR of workgroups (see the kernel)
    int Mb;                        // rows of one batch item (batch stride of y / addend / aux / rowstats); = M on entry
};

// One weight slot = all 2*TH row tiles of one k16 step; this wave (row half mh) multiplies TH of them, in groups of
// NF fragments: NF reads in flight, each MFMA issued as soon as its fragment has arrived.
template <int NF, int T0, int I>
struct FragLoop {
    static __device__ __forceinline__ void issue(uint32_t a, bf16x8 (&af)[NF]) {
        af[I] = lds_read_frag<(T0 + I) * 1024>(a);
        if constexpr (I + 1 < NF) FragLoop<NF, T0, I + 1>::issue(a, af);
    }
    template <int TH>
    static __device__ __forceinline__ void mfma(const bf16x8 (&af)[NF], const bf16x8& b, f32x16 (&acc)[TH]) {
        wait_lgkm<NF - 1 - I>();
        acc[T0 + I] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, af[I], acc[T0 + I], 0, 0, 0);   // D[pixel][m]
        if constexpr (I + 1 < NF) FragLoop<NF, T0, I + 1>::template mfma<TH>(af, b, acc);
    }
};

// ---- epilogue of one pass -----------------------------------------------------------------------------------------
// The MFMAs run "transposed" (A = pixel fragment, B = weight fragment), so an accumulator holds its 32 output rows m on
// the lanes and the 32 pixels in the registers, 4 consecutive pixels per register quad.  Every 32 x 32 tile goes through
// a wave-private 2 KB LDS tile [32 rows][32 px] bf16: written as 8 bytes per lane (one quad), read back as 16 bytes per
// lane (4 lanes per 64-byte row segment) and stored with global_store_dwordx4; addend / aux_in tiles take the same
// way in reverse.  The four 16-byte chunks of a row are rotated by (row >> 1) so neither direction piles onto one bank.
// All LDS traffic here is inline asm: the compiler would drain the LDS-DMA queue (vmcnt(0)) in front of every LDS
// access it can see.
struct EpiAddr {
    uint32_t st_acc;                // + 16 * ((g + rot) & 3): this lane's row in accumulator layout (8-byte quads)
    int rot;                        // (m_local >> 1) & 3
    uint32_t st_lin;                // this lane's 16-byte piece of the linear view (second piece: + 1024)
    int row_lin, px_lin;            // linear piece q: row = 16 q + row_lin, pixel offset px_lin
};

// HAS_IN is a template parameter and the bias load is unconditional on purpose: the compiler's waitcnt pass is path
// insensitive, and a VGPR load that sits behind a run-time `if` stays "possibly pending" for it after the join -- it then
// drains vmcnt(0) in the middle of the next tile's MFMA loop, where those registers are reused (measured: 2,300 cycles
// per weight group, the whole HBM latency of the X pieces in flight).
template <int TH, bool HAS_IN>
__device__ __forceinline__ void pce_epilogue(const PceParams& p, f32x16 (&acc)[TH], const EpiAddr& ea, int m_first, int m_local,
                                             long long tile_base /* b*M*P + n0 + 32 pg */, bool px_ok, int gmask,
                                             float (&ts1)[TH], float (&ts2)[TH], int row_base /* b*M */) {
    const unsigned short* in = reinterpret_cast<const unsigned short*>(p.aux_in ? p.aux_in : p.addend);
    constexpr bool has_in = HAS_IN;
    const bool mul_gelu_grad = p.aux_in != nullptr;
    u32x4 pf[2][2];
    auto fetch = [&](int t, u32x4 (&dst)[2]) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int m = m_first + 32 * t + 16 * q + ea.row_lin;
            // rows / pixels outside the field read the zero block behind the weight image: no branch around the load
            const void* src = (m < p.M && px_ok) ? (const void*)(in + tile_base + (long long)m * p.P + ea.px_lin) : (const void*)p.zeros;
            dst[q] = *reinterpret_cast<const u32x4*>(src);
        }
    };
    auto quad_addr = [&](int g) { return ea.st_acc + 16 * ((g + ea.rot) & 3); };
    auto flush = [&](unsigned short* out, int t) {      // staging tile -> global, 16 bytes per lane
        wait_lgkm<0>();
        const u32x4 a = lds_read_b128<0>(ea.st_lin);
        const u32x4 b = lds_read_b128<1024>(ea.st_lin);
        wait_lgkm<0>();
        const int m0 = m_first + 32 * t + ea.row_lin;
        if (px_ok) {
            u32x4* d0 = reinterpret_cast<u32x4*>(out + tile_base + (long long)m0 * p.P + ea.px_lin);
            u32x4* d1 = reinterpret_cast<u32x4*>(out + tile_base + (long long)(m0 + 16) * p.P + ea.px_lin);
            if (p.nt) {
                if (m0 < p.M) __builtin_nontemporal_store(a, d0);
                if (m0 + 16 < p.M) __builtin_nontemporal_store(b, d1);
            } else {
                if (m0 < p.M) *d0 = a;
                if (m0 + 16 < p.M) *d1 = b;
            }
        }
    };
    auto stage = [&](const float (&v)[16]) {            // accumulator layout -> staging tile
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            u32x2 q;
            q[0] = pack_bf16x2(v[4 * g], v[4 * g + 1]);
            q[1] = pack_bf16x2(v[4 * g + 2], v[4 * g + 3]);
            lds_write_b64<0>(quad_addr(g), q);
        }
    };
    // all bias values up front: a load issued between the stores of two tiles would be waited for with vmcnt(0), i.e. behind
    // the round trip of every store queued before it
    // (the values come from the workgroup's LDS copy: a global load here would expose an L2 round trip per tile)
    float bias_t[TH];
#pragma unroll
    for (int t = 0; t < TH; ++t) bias_t[t] = lds_read_b32(p.bias_lds + 4 * (m_first + 32 * t + m_local));   // zeros if the layer has none
    wait_lgkm<0>();
    // the addend's per-row affine map (identity without one), loaded with the bias for the same reason
    float aff_a[TH], aff_b[TH];
#pragma unroll
    for (int t = 0; t < TH; ++t) {
        aff_a[t] = 1.f;
        aff_b[t] = 0.f;
    }
    if constexpr (has_in) {
        if (p.aff) {
#pragma unroll
            for (int t = 0; t < TH; ++t) {
                const float2 ab = *reinterpret_cast<const float2*>(p.aff + 2 * (long long)(row_base + min(m_first + 32 * t + m_local, p.M - 1)));
                aff_a[t] = ab.x;
                aff_b[t] = ab.y;
            }
        }
    }
    if constexpr (has_in) fetch(0, pf[0]);
#pragma unroll
    for (int t = 0; t < TH; ++t) {
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = acc[t][r];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] += bias_t[t];
        if (p.aux_out) {
            stage(v);
            flush(reinterpret_cast<unsigned short*>(p.aux_out), t);
        }
        if (p.gelu) {
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = gelu_f(v[r]);
        }
        if constexpr (has_in) {
            if (t + 1 < TH) fetch(t + 1, pf[(t + 1) % 2]);
            lds_write_b128<0>(ea.st_lin, pf[t % 2][0]);
            lds_write_b128<1024>(ea.st_lin, pf[t % 2][1]);
            wait_lgkm<0>();
            u32x2 q[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) q[g] = lds_read_b64(quad_addr(g));
            wait_lgkm<0>();
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float i0 = __uint_as_float(q[g][0] << 16), i1 = __uint_as_float(q[g][0] & 0xFFFF0000u);
                const float i2 = __uint_as_float(q[g][1] << 16), i3 = __uint_as_float(q[g][1] & 0xFFFF0000u);
                if (mul_gelu_grad) {
                    v[4 * g] *= gelu_grad_f(i0); v[4 * g + 1] *= gelu_grad_f(i1);
                    v[4 * g + 2] *= gelu_grad_f(i2); v[4 * g + 3] *= gelu_grad_f(i3);
                } else {
                    v[4 * g] += fmaf(aff_a[t], i0, aff_b[t]); v[4 * g + 1] += fmaf(aff_a[t], i1, aff_b[t]);
                    v[4 * g + 2] += fmaf(aff_a[t], i2, aff_b[t]); v[4 * g + 3] += fmaf(aff_a[t], i3, aff_b[t]);
                }
            }
        }
        stage(v);
        flush(reinterpret_cast<unsigned short*>(p.y), t);
        if (p.rowstats) {
            // row sums of what was stored (the bf16-rounded values), valid 8-pixel groups only; this lane's row is m_local
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (gmask & (1 << g)) {
#pragma unroll
                    for (int j = 0; j < 4; j += 2) {
                        const uint32_t pk = pack_bf16x2(v[4 * g + j], v[4 * g + j + 1]);
                        const float a = __uint_as_float(pk << 16), b = __uint_as_float(pk & 0xFFFF0000u);
                        s1 += a + b;
                        s2 = fmaf(a, a, fmaf(b, b, s2));
                    }
                }
            }
            ts1[t] += s1;
            ts2[t] += s2;
        }
    }
}

// KSP: k16 steps per K phase (2, 4 or 8: phases of <= 128 k rows), NPH: phases (1, 2, 3, 6), TH: 32-row tiles per wave.
//
// Schedule of one workgroup (all 8 waves alike).  The weight stream is cut into GROUPS of two k16 steps (24 KB at
// TH = 6) that rotate through THREE LDS buffers; fragments are read one half-step ahead of the MFMAs that use them:
//
//   iteration g:   wait for my pieces of group g + 1 (vmcnt, see below); barrier
//                  DMA: group g + 2 -> the buffer group g - 1 lived in; up to two pieces of a coming X region
//                  ds_read step 1 of group g -> af1      | 6 MFMAs on af0 (step 0 of group g, read in iteration g - 1)
//                  ds_read step 0 of group g + 1 -> af0  | 6 MFMAs on af1
//
// so the LDS latency, the DMA issue and the barrier skew of a wave sit under MFMAs (its own or those of the other wave
// of its SIMD) instead of in front of them -- the earlier two-buffer schedule read a group's fragments only after the
// barrier that published it and ran 2,000 cycles per 768 cycles of matrix work.
// vmcnt retires in issue order, so a wave issues its weight pieces BEFORE its X pieces: the wait for the weights is then
// "all but my youngest cx operations", which leaves the X pieces (HBM latency) in flight for another iteration.  X pieces
// go out in the first half of a phase only, so the vmcnt(0) waits of its later groups have retired them long before the
// region is read (two phases later) and before the epilogue's stores are queued.
template <int KSP, int NPH, int TH, bool HAS_IN>
__global__ __launch_bounds__(PT, 2) void pce_kernel(PceParams p) {
    constexpr int SLOT = 2 * TH * 1024;          // 2 TH row tiles x one k16 step
    constexpr int GS = 2;                        // k16 steps per group
    constexpr int GROUP = GS * SLOT;
    constexpr int NBUF = 3;
    constexpr int NG = KSP / GS;                 // groups per phase
    constexpr int NPG = GS * 2 * TH;             // 1 KB weight pieces per group, piece q issued by wave q % 8
    constexpr int NREG = NPH > 1 ? 2 : 1;        // X regions in LDS: the q-th phase a workgroup runs lives in region q & 1
    constexpr int REG_BYTES = 16 * KSP * XROW;   // one region = the k rows of one phase x 128 px
    constexpr int NXP = 4 * KSP;                 // 1 KB pieces per X region (4 k rows each)
    constexpr int XG = (NXP + 15) / 16;          // groups of a phase that carry X pieces (2 per wave and group)
    static_assert(XG <= NG, "X pieces must fit the groups of a phase");
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    char* XS = lds;
    char* WB = lds + NREG * REG_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pg = wave & 3, mh = wave >> 2;
    constexpr int ngroup_tile = NPH * NG;
    const int ntiles = (int)p.ntiles, tiles_per_b = (int)p.tiles_per_b;

    // Two passes (M > 64 TH rows) on ONE workgroup fetch the X tile twice, and at 721 x 1440 the second fetch came from HBM again
    // (PMC: 1.68 GB read per 384 -> 768 launch for 0.80 GB of X -- between the two fetches 9 MB stream through the XCD's 4 MB L2).
    // With `split` the two passes of a tile go to workgroups w and w + 8 instead: same XCD, same tile sequence, in step with
    // each other, so the second request for a line meets the first in L2.  Each workgroup then is a one-pass kernel on its
    // half of the rows: the row-indexed pointers are advanced by the half's first row and everything below sees npass = 1.
    int wg = (int)blockIdx.x, nwg = (int)gridDim.x;
    if (p.split) {
        const int h = (wg >> 3) & 1;
        wg = ((wg >> 4) << 3) | (wg & 7);
        nwg >>= 1;
        const int r0 = h * 64 * TH;                    // first row of this half
        p.wimg += h * p.img_per_pass;
        if (r0 < p.nbias) { p.bias += r0; p.nbias -= r0; } else { p.bias += p.nbias - 1; p.nbias = 1; }
        const long long roff = (long long)r0 * p.P;
        p.y += roff;
        if (p.aux_out) p.aux_out += roff;
        if (p.addend) p.addend += roff;
        if (p.aux_in) p.aux_in += roff;
        if (p.aff) p.aff += 2 * r0;
        if (p.rowstats) p.rowstats += 2 * r0;
        p.M -= r0;                                     // rows left from r0 on (the first half never looks past its 64 TH rows)
        p.npass = 1;
    }

    // Per-lane addressing of the X fragments and of the epilogue is recomputed from an opaque copy of the lane id where
    // it is used (once per tile): kept live across the MFMA loop it would be spilled, and a scratch reload inside the
    // epilogue waits behind the stores queued before it.
    auto opaque_lane = [&]() {
        int l = lane;
        asm volatile("" : "+v"(l));
        return l;
    };

    // ---- DMA issue: every wave-instruction is issued with all lanes (lanes outside the field or past K fetch from a
    //      zero block behind the weight image), so the vmcnt bookkeeping is static and LDS never keeps stale rows ----
    const char* w_lane = p.wimg + lane * 16 + wave * 1024;
    auto issue_group = [&](int group_in_image, int buf) {
        const char* src = w_lane + (long long)group_in_image * GROUP;
        char* dst = WB + buf * GROUP + wave * 1024;
#pragma unroll
        for (int q = 0; q < (NPG + 7) / 8; ++q)
            if (wave + 8 * q < NPG) dma16(src + q * 8192, dst + q * 8192);
    };
    // X region image: k row r at r * 256 bytes, its 16 chunks of 8 pixels rotated by 4 * (r & 3) so that the 4 rows x
    // 4 chunks a transposing read touches sit on 16 different 16-byte bank slots.  Piece n = k rows 4 n .. 4 n + 3.
    struct XTarget {
        const __hip_bfloat16* src;    // this lane's source of piece 0 (k row = lane >> 4)
        char* dst;                    // LDS region
        int krow0;                    // this lane's k row in piece 0
        bool lane_ok;                 // this lane's pixels are inside the field
        bool active;
    };
    // Work index -> pixel tile.  Workgroup w runs on XCD w % 8; with the plain order (tile = work index) neighbouring tiles
    // sit on different XCDs, and where a field row is not a multiple of 128 bytes (721 x 1440 bf16: 64 mod 128) the
    // 128-byte lines at a tile's edges are shared with the neighbours: each half is fetched / written back by another L2.
    // With xcd_runs the eight XCDs take contiguous runs of W / 8 tiles of each W-tile window (W = gridDim.x), so a shared
    // line meets both of its halves in one L2 (the forward / inverse FFT do the same with tile pairs, fft_split.h).
    auto phys_tile = [&](int t) {
        const int W = nwg, base = t - wg;
        if (!p.xcd_runs || base + W > ntiles) return t;        // the last, partial window keeps the plain order
        return base + (wg & 7) * (W >> 3) + (wg >> 3);
    };
    auto x_target = [&](int work, int phase, int region) {
        XTarget t;
        const int tile = phys_tile(work);
        const int b = tile / tiles_per_b;
        const int l = opaque_lane();
        const int x_chunk = ((l & 15) - 4 * ((l >> 4) & 3)) & 15;   // global chunk this lane fetches
        const long long n = (long long)(tile - b * tiles_per_b) * PN + x_chunk * 8;
        t.krow0 = phase * 16 * KSP + (l >> 4);
        t.src = p.x + ((long long)b * p.K + t.krow0) * p.P + n;
        t.dst = XS + region * REG_BYTES;
        t.lane_ok = n < p.P;
        t.active = true;
        return t;
    };
    // pieces 16 xg + 2 wave + {0, 1} of the target region; returns how many this wave issued
    auto issue_x = [&](const XTarget& t, int xg) {
        int cnt = 0;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = 16 * xg + 2 * wave + j;
            if (n < NXP) {
                const void* src = (t.krow0 + 4 * n < p.K && t.lane_ok) ? (const void*)(t.src + (long long)(4 * n) * p.P)
                                                                       : (const void*)p.zeros;
                dma16(src, t.dst + n * 1024);
                ++cnt;
            }
        }
        return cnt;
    };

    int tile = wg;
    int g = 0;           // running group counter: group g lives in ring buffer g % 3
    int gi = 0;          // groups issued so far
    int landed = 0;      // groups known to have landed (drained before the last epilogue)
    int si = 0;          // group-in-item index of the next group to issue
    int spass = 0;       // pass the next group to issue belongs to
    bool si_next = false;   // ... which belongs to the item after the current one
    int cx = 0;          // X pieces this wave issued after its last weight pieces
    int q = 0;           // running phase counter: phase q lives in X region q & (NREG - 1)
    // The work of a workgroup is a sequence of ITEMS (tile, pass): all passes of a tile back to back, then the next tile.
    auto issue_next_group = [&](bool item_after_exists) {
        if (si_next && !item_after_exists) return;
        if (!PCE_EXP(4)) issue_group(spass * ngroup_tile + si, gi % NBUF);
        ++gi;
        if (++si == ngroup_tile) {
            si = 0;
            si_next = true;
            if (++spass == p.npass) spass = 0;
        }
    };
    const uint32_t a_lane = lds_addr(WB) + mh * TH * 1024 + lane * 16;   // this wave's weight fragments inside a ring buffer
    bf16x8 af0[TH], af1[TH];
    // the bias of all passes, once per workgroup, behind the staging tiles (read by the epilogue of every tile)
    p.bias_lds = lds_addr(WB) + NBUF * GROUP + 8 * 2048;
    for (int i = tid; i < p.npass * 64 * TH; i += PT) lds_write_b32(p.bias_lds + 4 * i, p.bias[min(i, p.nbias - 1)]);
    wait_lgkm<0>();
    if (tile < ntiles) {
        const bool after = p.npass > 1 || tile + nwg < ntiles;
        issue_next_group(after);
        issue_next_group(after);
        for (int ph = 0; ph < NREG; ++ph) {
            const XTarget t = x_target(tile, ph, ph);
            for (int xg = 0; xg < XG; ++xg) issue_x(t, xg);
        }
        wait_vm0();      // the first item's X regions and weight groups: simplest to drain here
        landed = gi;
        block_sync();
        FragLoop<TH, 0, 0>::issue(a_lane, af0);      // step 0 of group 0
    }

#ifdef MK_PCE_STAMPS   // profiling build only (tools/pce_stamps.py): the stamps cost registers in the hot loop
    int stamp_n = 0;
    auto stamp = [&]() {
        if (p.dbg && blockIdx.x == 0 && tile == (int)gridDim.x * 3 && stamp_n < 64) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (lane == 0) p.dbg[wave * 64 + stamp_n] = t;
            ++stamp_n;
        }
    };
#else
    auto stamp = [&]() {};
#endif
    // per-row sums of the output (p.rowstats): accumulated per lane over the tiles of one batch item, one set per pass
    float rs1[2][TH], rs2[2][TH];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < TH; ++t) rs1[i][t] = rs2[i][t] = 0.f;
    int rs_b = -1;
    auto flush_rowstats = [&]() {       // every wave of the workgroup calls this at the same points
        if (!p.rowstats || rs_b < 0) return;
        const uint32_t stg = lds_addr(WB) + NBUF * GROUP;         // the 8 staging tiles (2 KB each): [wave][(t, k)][32 rows] floats
        const int l = opaque_lane();
        const int ml = l & 31;
#pragma unroll
        for (int i = 0; i < 2; ++i) {                             // one pass' set at a time: TH * 2 * 128 B <= 2 KB per wave
            if (i < p.npass) {
#pragma unroll
                for (int t = 0; t < TH; ++t) {
                    const float a = rs1[i][t] + __shfl_xor(rs1[i][t], 32), b = rs2[i][t] + __shfl_xor(rs2[i][t], 32);
                    if (l < 32) {
                        lds_write_b32(stg + wave * 2048 + (t * 2 + 0) * 128 + ml * 4, a);
                        lds_write_b32(stg + wave * 2048 + (t * 2 + 1) * 128 + ml * 4, b);
                    }
                }
                wait_lgkm<0>();
                block_sync();
                if (pg == 0 && l < 32) {
                    for (int t = 0; t < TH; ++t) {
                        const int m = i * 64 * TH + mh * 32 * TH + 32 * t + ml;
                        float a = 0.f, b = 0.f;
                        for (int w = 0; w < 4; ++w) {
                            const float a1 = lds_read_b32(stg + (mh * 4 + w) * 2048 + (t * 2 + 0) * 128 + ml * 4);
                            const float b1 = lds_read_b32(stg + (mh * 4 + w) * 2048 + (t * 2 + 1) * 128 + ml * 4);
                            wait_lgkm<0>();
                            a += a1;
                            b += b1;
                        }
                        if (m < p.M) {
                            atomicAdd(p.rowstats + ((long long)rs_b * p.Mb + m) * 2, (double)a);
                            atomicAdd(p.rowstats + ((long long)rs_b * p.Mb + m) * 2 + 1, (double)b);
                        }
                    }
                }
                block_sync();
            }
#pragma unroll
            for (int t = 0; t < TH; ++t) rs1[i][t] = rs2[i][t] = 0.f;
        }
    };
    for (; tile < ntiles; tile += nwg)
    for (int pass = 0; pass < p.npass; ++pass) {
        // the item after this one: the next pass over the same tile, or the first pass over the next tile
        const bool last_pass = pass + 1 == p.npass;
        const int next_tile = last_pass ? tile + nwg : tile;
        const bool item_after = next_tile < ntiles;
        const int ptile = phys_tile(tile);
        const int b = ptile / tiles_per_b;
        const long long n0 = (long long)(ptile - b * tiles_per_b) * PN;
        f32x16 acc[TH];
        stamp();   // 0: tile start
        if (si_next) si_next = false;               // the stream's "next item" is this item now
#pragma unroll
        for (int i = 0; i < TH; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        auto run_phase = [&](const int phase) {
            // ---- X phase: every wave has retired its pieces of this region (see the schedule above); the barrier makes
            //      them visible; pull this wave's pixel columns into registers ----
            const int region = q & (NREG - 1);
            block_sync();
            stamp();   // 1
            bf16x8 xf[KSP];
            {
                // (inline asm, not the builtin: with LDS-DMA in flight the compiler puts s_waitcnt vmcnt(0) in front of every
                // LDS read it can see -- here that drained the weight groups and the X pieces of the coming phases, HBM
                // latency included, at the start of every phase: 1,400 cycles in the stamps, three times per tile)
                const int l = opaque_lane();
                const int rowq = (l & 15) >> 2;
                const int ch = 4 * pg + 2 * ((l >> 4) & 1) + ((l & 3) >> 1);
                const uint32_t xfrag_lane = lds_addr(XS) + region * REG_BYTES + (8 * (l >> 5) + rowq) * XROW +
                                            ((ch + 4 * rowq) & 15) * 16 + (l & 1) * 8;
                if (!PCE_EXP(16)) {
                    u32x2 lo[KSP], hi[KSP];
                    [&]<int... SS>(std::integer_sequence<int, SS...>) {
                        ((lo[SS] = lds_read_tr16<SS * 16 * XROW>(xfrag_lane), hi[SS] = lds_read_tr16<SS * 16 * XROW + 4 * XROW>(xfrag_lane)), ...);
                    }(std::make_integer_sequence<int, KSP>{});
                    wait_lgkm<0>();
#pragma unroll
                    for (int s2 = 0; s2 < KSP; ++s2)
                        xf[s2] = __builtin_bit_cast(bf16x8, u32x4{lo[s2][0], lo[s2][1], hi[s2][0], hi[s2][1]});
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            block_sync();                                       // the region may be overwritten
            stamp();   // 2
            // fetched into this region while the phase computes: the phase NREG ahead (of this item, or of the next one)
            XTarget xt;
            xt.active = false;
            if (!PCE_EXP(8)) {
                if (phase + NREG < NPH) xt = x_target(tile, phase + NREG, region);
                else if (item_after) xt = x_target(next_tile, phase + NREG - NPH, region);
            }
            ++q;
#pragma unroll
            for (int gq = 0; gq < NG; ++gq) {
                const bool more1 = phase * NG + gq + 1 < ngroup_tile || item_after;   // group g + 1 exists
                stamp();   // 3 + 2 gq
                if (more1 && g + 1 >= landed) {                 // my pieces of group g + 1: all but my cx youngest operations
                    if (cx >= 2) wait_vm<2>();
                    else if (cx == 1) wait_vm<1>();
                    else wait_vm0();
                }
                if (!PCE_EXP(32)) block_sync();                  // group g + 1 visible; nobody reads group g - 1 any more
                stamp();   // 4 + 2 gq
                issue_next_group(item_after);                   // group g + 2 -> the buffer of group g - 1
                cx = (gq < XG && xt.active) ? issue_x(xt, gq) : 0;
                if (!PCE_EXP(2)) {
                    const uint32_t a = a_lane + (g % NBUF) * GROUP;
                    FragLoop<TH, 0, 0>::issue(a + SLOT, af1);                       // step 1 of group g
                    wait_lgkm<TH>();                                                // af0 (older) has arrived
#pragma unroll
                    for (int t = 0; t < TH; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[GS * gq], af0[t], acc[t], 0, 0, 0);   // D[pixel][m]
                    wait_lgkm<0>();                                                 // af1 has arrived
                    if (more1) FragLoop<TH, 0, 0>::issue(a_lane + ((g + 1) % NBUF) * GROUP, af0);   // step 0 of group g + 1
#pragma unroll
                    for (int t = 0; t < TH; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[GS * gq + 1], af1[t], acc[t], 0, 0, 0);
                }
                ++g;
            }
        };
        run_phase(0);
        if constexpr (NPH >= 2) run_phase(1);
        if constexpr (NPH >= 3) run_phase(2);
        if constexpr (NPH >= 6) {
            run_phase(3);
            run_phase(4);
            run_phase(5);
        }
        stamp();   // epilogue start
        // everything this wave has in flight (weight groups of the next item, late X pieces) lands before the epilogue's
        // stores are queued behind it: no later wait has to sit out a store's round trip
        wait_vm0();
        landed = gi;
        cx = 0;
        stamp();
        if (!PCE_EXP(1)) {
            EpiAddr ea;
            const int l = opaque_lane();
            const uint32_t stg = lds_addr(WB) + NBUF * GROUP + wave * 2048;    // wave-private [32 rows][32 px] bf16 tile
            const int ml = l & 31;
            ea.rot = (ml >> 1) & 3;
            ea.st_acc = stg + ml * 64 + (l >> 5) * 8;
            ea.row_lin = l >> 2;
            ea.px_lin = (l & 3) * 8;
            ea.st_lin = stg + ea.row_lin * 64 + (((l & 3) + (ea.row_lin >> 1)) & 3) * 16;
            const long long px0 = n0 + 32 * pg;
            if (p.rowstats && b != rs_b) {          // a new batch item: hand over the sums of the previous one
                flush_rowstats();
                rs_b = b;
            }
            int gmask = 0;
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) gmask |= (px0 + 8 * gg < p.P) ? (1 << gg) : 0;
            float ts1[TH], ts2[TH];
#pragma unroll
            for (int t = 0; t < TH; ++t) ts1[t] = ts2[t] = 0.f;
            pce_epilogue<TH, HAS_IN>(p, acc, ea, pass * 64 * TH + mh * 32 * TH, ml, (long long)b * p.Mb * p.P + px0,
                                     px0 + ea.px_lin < p.P, gmask, ts1, ts2, b * p.Mb);
            if (p.rowstats) {
                if (pass == 0) {
#pragma unroll
                    for (int t = 0; t < TH; ++t) { rs1[0][t] += ts1[t]; rs2[0][t] += ts2[t]; }
                } else {
#pragma unroll
                    for (int t = 0; t < TH; ++t) { rs1[1][t] += ts1[t]; rs2[1][t] += ts2[t]; }
                }
            }
        } else {
#pragma unroll
            for (int t = 0; t < TH; ++t) keep_alive(acc[t]);
        }
        stamp();   // epilogue end
    }
    flush_rowstats();
    // drain: nothing may be in flight into LDS when the workgroup's LDS is released
    wait_vm0();
}


// ---- weight image ----------------------------------------------------------------------------------------------
// element (pass, k16 step ks, row tile rt, lane, j)  <-  A[m][k],  m = pass*64*TH + rt*32 + (lane & 31),
// k = 16 ks + 8 (lane >> 5) + j;  A = W or W^T;  one slot = the 2 TH fragments of one k16 step; 32 zero elements at the end
template <typename T>
__device__ __forceinline__ unsigned short pce_pack_element(const T* __restrict__ w, int transpose, int M, int K, int ldw, int TH,
                                                           int steps_per_pass, long long core, long long idx) {
    if (idx >= core) return 0;
    const int j = (int)(idx & 7);
    const int lane = (int)((idx >> 3) & 63);
    long long f = idx >> 9;                 // fragment index
    const int rt = (int)(f % (2 * TH));
    f /= 2 * TH;
    const int pass = (int)(f / steps_per_pass), ks = (int)(f % steps_per_pass);
    const int m = pass * 64 * TH + rt * 32 + (lane & 31);
    const int k = 16 * ks + 8 * (lane >> 5) + j;
    float v = 0.f;
    if (m < M && k < K) v = (float)(transpose ? w[(long long)k * ldw + m] : w[(long long)m * ldw + k]);
    return f32_to_bf16_bits(v);
}

template <typename T>
__global__ void pce_pack_kernel(const T* __restrict__ w, int transpose, int M, int K, int ldw, int TH, int steps_per_pass,
                                unsigned short* __restrict__ img, long long core, long long total) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    img[idx] = pce_pack_element(w, transpose, M, K, ldw, TH, steps_per_pass, core, idx);
}

// Many images in one launch (mk_pce_pack_batch): descriptor e = 10 x int64 {weight pointer, dtype, transpose, M, K, ldw, TH,
// steps per pass, core elements, first element in the arena}; the images lie back to back, descriptor n holds the total
struct PackDesc {
    long long w, dtype, transpose, M, K, ldw, TH, spp, core, first;
};
__global__ void pce_pack_batch_kernel(const PackDesc* __restrict__ desc, int n, unsigned short* __restrict__ arena, long long total) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    int lo = 0, hi = n;                     // last descriptor whose first element is <= idx
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (desc[mid].first <= idx) lo = mid;
        else hi = mid;
    }
    const PackDesc d = desc[lo];
    const long long local = idx - d.first;
    arena[idx] = d.dtype == 0
        ? pce_pack_element(reinterpret_cast<const float*>(d.w), (int)d.transpose, (int)d.M, (int)d.K, (int)d.ldw, (int)d.TH, (int)d.spp, d.core, local)
        : pce_pack_element(reinterpret_cast<const __hip_bfloat16*>(d.w), (int)d.transpose, (int)d.M, (int)d.K, (int)d.ldw, (int)d.TH, (int)d.spp,
                           d.core, local);
}

// zero bias for layers without one (the epilogue's bias load is unconditional); first use must not be inside a capture
static const float* pce_zero_bias() {
    static float* z = [] {
        float* q = nullptr;
        if (hipMalloc(&q, 4096) != hipSuccess) return (float*)nullptr;
        (void)hipMemset(q, 0, 4096);
        return q;
    }();
    return z;
}

static unsigned long long* pce_dbg_buffer() {
    static unsigned long long* d = [] {
        unsigned long long* q = nullptr;
        if (getenv("MK_PCE_DBG")) {
            if (hipMalloc(&q, 8 * 64 * 8) != hipSuccess) return (unsigned long long*)nullptr;
            (void)hipMemset(q, 0, 8 * 64 * 8);
        }
        return q;
    }();
    return d;
}

struct PceCfg {
    int KSP, NPH, TH, npass;
};
static bool pce_config(int M, int K, PceCfg* c) {
    if (M <= 0 || K <= 0 || K > 768) return false;
    c->KSP = K > 64 ? 8 : (K > 32 ? 4 : 2);             // k16 steps per phase: phases of <= 128 k rows
    const int nph = mk::ceil_div(K, 128);
    c->NPH = nph > 3 ? 6 : nph;                         // built for 1, 2, 3 and 6 phases (4 and 5 pad to six)
    c->TH = M > 128 ? 6 : (M > 64 ? 2 : 1);
    c->npass = mk::ceil_div(M, 64 * c->TH);
    return c->npass <= 4;                               // M <= 1536: four passes of 384 rows (the bias copy in LDS is sized for four)
}
static long long pce_image_core_bytes(const PceCfg& c) { return (long long)c.npass * c.NPH * c.KSP * 2 * c.TH * 1024; }

static int pce_cu_count() {
    static const int ncu = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n > 0 ? n : 256;
    }();
    return ncu;
}

template <int KSP, int NPH, int TH, bool HAS_IN>
static int pce_launch(const PceParams& p, hipStream_t st) {
    constexpr int LDS = (NPH > 1 ? 2 : 1) * 16 * KSP * XROW + 3 * 2 * 2 * TH * 1024 + 8 * 2048 + 4 * 64 * TH * 4;   // X regions + three weight groups + 8 staging tiles + the bias of up to four passes
    static const bool once = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pce_kernel<KSP, NPH, TH, HAS_IN>),
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
    PceParams q = p;
    // MK_PCE_XCD_RUNS: 0 plain tile order, 1 (default) XCD runs where rows are not whole 128-byte lines, 2 always
    static const int runs_rule = [] { const char* e = getenv("MK_PCE_XCD_RUNS"); return e ? atoi(e) : 1; }();
    // MK_PCE_SPLIT: 0 both passes of a wide layer on one workgroup, 1 (default) on workgroup pairs w, w + 8 (see the kernel)
    static const int split_rule = [] { const char* e = getenv("MK_PCE_SPLIT"); return e ? atoi(e) : 1; }();
    q.split = split_rule && p.npass == 2 && grid == ncu && grid % 16 == 0 && p.ntiles >= grid;
    const long long wgs_per_pass = q.split ? grid / 2 : grid;
    q.xcd_runs = wgs_per_pass % 8 == 0 && (runs_rule == 2 || (runs_rule == 1 && (p.P * 2) % 128 != 0));
    hipLaunchKernelGGL((pce_kernel<KSP, NPH, TH, HAS_IN>), dim3((unsigned)grid), dim3(PT), LDS, st, q);
    return 0;
}

}  // namespace

extern "C" long long mk_pce_image_bytes(int M, int K) {
    PceCfg c;
    if (!pce_config(M, K, &c)) return 0;
    return pce_image_core_bytes(c) + 64;    // + the zero block masked DMA lanes read
}

extern "C" int mk_pce_pack(const void* w, int w_dtype, int transpose, int M, int K, int ldw, void* img, void* stream) {
    MK_REQUIRE(w && img, "null pointer");
    MK_REQUIRE(w_dtype == 0 || w_dtype == 1, "w_dtype must be 0 (fp32) or 1 (bf16)");
    PceCfg c;
    MK_REQUIRE(pce_config(M, K, &c), "unsupported shape (K <= 768, M <= 1536)");
    MK_REQUIRE(ldw >= (transpose ? M : K), "leading dimension too small");
    const long long total = (pce_image_core_bytes(c) + 64) / 2;
    const int spp = c.NPH * c.KSP;
    const long long nblk = (total + 255) / 256;
    if (w_dtype == 0)
        hipLaunchKernelGGL(pce_pack_kernel<float>, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, (const float*)w,
                           transpose, M, K, ldw, c.TH, spp, (unsigned short*)img, total - 32, total);
    else
        hipLaunchKernelGGL(pce_pack_kernel<__hip_bfloat16>, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream,
                           (const __hip_bfloat16*)w, transpose, M, K, ldw, c.TH, spp, (unsigned short*)img, total - 32, total);
    MK_LAUNCH_CHECK();
    return 0;
}

extern "C" int mk_pce_pack_layout(int M, int K, long long* out3) {
    MK_REQUIRE(out3, "null pointer");
    PceCfg c;
    MK_REQUIRE(pce_config(M, K, &c), "unsupported shape (K <= 768, M <= 1536)");
    out3[0] = c.TH;
    out3[1] = (long long)c.NPH * c.KSP;
    out3[2] = pce_image_core_bytes(c) / 2;
    return 0;
}

extern "C" int mk_pce_pack_batch(const void* desc_dev, int n, void* arena, long long total_elements, void* stream) {
    MK_REQUIRE(desc_dev && arena && n > 0 && total_elements > 0, "bad arguments");
    const long long nblk = (total_elements + 255) / 256;
    MK_REQUIRE(nblk < 2147483647LL, "arena too large");
    hipLaunchKernelGGL(pce_pack_batch_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, (const PackDesc*)desc_dev, n,
                       (unsigned short*)arena, total_elements);
    MK_LAUNCH_CHECK();
    return 0;
}

namespace {
__global__ void pce_zero_kernel(double* p, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0.0;
}
}  // namespace

extern "C" int mk_pce_gemm(const void* x, const void* wimg, void* y, const float* bias, const void* addend,
                           const void* aux_in, void* aux_out, int gelu, int batch, int M, int K, long long P, void* stream) {
    return mk_pce_gemm_ex(x, wimg, y, bias, addend, nullptr, aux_in, aux_out, gelu, nullptr, batch, M, K, P, stream);
}

extern "C" int mk_pce_gemm_ex(const void* x, const void* wimg, void* y, const float* bias, const void* addend,
                              const float* addend_affine, const void* aux_in, void* aux_out, int gelu, double* rowstats,
                              int batch, int M, int K, long long P, void* stream) {
    MK_REQUIRE(x && wimg && y, "null pointer");
    MK_REQUIRE(batch > 0 && M > 0 && K > 0 && P > 0, "bad sizes");
    MK_REQUIRE((P % 8) == 0, "P = H*W must be a multiple of 8 (16-byte row alignment)");
    MK_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)wimg & 15) == 0, "x and the weight image must be 16-byte aligned");
    PceCfg c;
    MK_REQUIRE(pce_config(M, K, &c), "unsupported shape (K <= 768, M <= 1536)");
    static const int pexp = [] { const char* e = getenv("MK_PCE_EXP"); return e ? atoi(e) : 0; }();
    MK_REQUIRE(!(addend && aux_in), "addend and aux_in are exclusive");
    MK_REQUIRE(!addend_affine || addend, "addend_affine needs an addend");
    MK_REQUIRE(((uintptr_t)addend_affine & 7) == 0, "addend_affine must be 8-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    if (rowstats) {
        MK_REQUIRE(c.npass <= 2, "row statistics are built for M <= 768");
        const long long n = 2LL * batch * M;
        hipLaunchKernelGGL(pce_zero_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, rowstats, n);
    }
    const float* zero_bias = bias ? nullptr : pce_zero_bias();
    MK_REQUIRE(bias || zero_bias, "cannot allocate the zero bias");
    const long long tiles_per_b = (P + PN - 1) / PN;
    MK_REQUIRE(tiles_per_b * batch < 2147483647LL, "too many pixel tiles");
    // M > 64 TH rows: passes of 64 TH rows over the same X tile inside one launch
    const char* zeros = (const char*)wimg + pce_image_core_bytes(c);
    {
        PceParams p;
        p.x = (const __hip_bfloat16*)x;
        p.wimg = (const char*)wimg;
        p.zeros = zeros;
        p.y = (__hip_bfloat16*)y;
        p.bias = bias ? bias : zero_bias;
        p.nbias = bias ? M : 1024;
        p.addend = (const __hip_bfloat16*)addend;
        p.aff = addend_affine;
        p.aux_in = (const __hip_bfloat16*)aux_in;
        p.aux_out = (__hip_bfloat16*)aux_out;
        p.rowstats = rowstats;
        p.gelu = gelu;
        p.M = M;
        p.K = K;
        p.B = batch;
        p.npass = c.npass;
        p.img_per_pass = (long long)c.NPH * c.KSP * 2 * c.TH * 1024;
        p.P = P;
        p.tiles_per_b = tiles_per_b;
        p.ntiles = tiles_per_b * batch;
        p.dbg = pce_dbg_buffer();
        p.exp = pexp;
        {   // MK_PCE_NT: 0 plain stores, 1 nontemporal always (default), n > 1: nontemporal when a field row has at least n pixels.
            // Measured (tools/pce_bench.py, alternating runs on one box): 384 -> 384 at 721x1440 0.493 -> 0.476 ms, with the skip
            // add at 240x480 0.076 -> 0.068, fc1 + bias + GELU + pre-activation at 240x480 0.163 -> 0.152, the full-resolution
            // fc1 / fc2 launches unchanged; the step 44.57 / 44.65 -> 44.46 / 44.49 ms.
            static const long long nt_rule = [] { const char* e = getenv("MK_PCE_NT"); return e ? atoll(e) : 1LL; }();
            p.nt = nt_rule == 1 || (nt_rule > 1 && P >= nt_rule);
        }
        p.xcd_runs = 0;     // set by pce_launch from the grid it picks
        p.split = 0;        // likewise
        p.Mb = M;
        bool done = false;
#define MK_PCE_CASE(KSP_, NPH_, TH_) \
        if (!done && c.KSP == KSP_ && c.NPH == NPH_ && c.TH == TH_) {                                         \
            if (p.addend || p.aux_in) pce_launch<KSP_, NPH_, TH_, true>(p, st);                               \
            else pce_launch<KSP_, NPH_, TH_, false>(p, st);                                                   \
            done = true;                                                                                        \
        }
        MK_PCE_CASE(2, 1, 1) MK_PCE_CASE(2, 1, 2) MK_PCE_CASE(2, 1, 6)
        MK_PCE_CASE(4, 1, 1) MK_PCE_CASE(4, 1, 2) MK_PCE_CASE(4, 1, 6)
        MK_PCE_CASE(8, 1, 1) MK_PCE_CASE(8, 1, 2) MK_PCE_CASE(8, 1, 6)
        MK_PCE_CASE(8, 2, 1) MK_PCE_CASE(8, 2, 2) MK_PCE_CASE(8, 2, 6)
        MK_PCE_CASE(8, 3, 1) MK_PCE_CASE(8, 3, 2) MK_PCE_CASE(8, 3, 6)
        MK_PCE_CASE(8, 6, 1) MK_PCE_CASE(8, 6, 2) MK_PCE_CASE(8, 6, 6)
#undef MK_PCE_CASE
        MK_REQUIRE(done, "no kernel instance for this shape");
        MK_LAUNCH_CHECK();
    }
    return 0;
}

// debug: s_memtime stamps of workgroup 0, fourth tile, of the last launch (MK_PCE_DBG=1): 8 waves x 64 stamps
extern "C" int mk_pce_debug_stamps(unsigned long long* out512) {
    MK_REQUIRE(out512, "null pointer");
    unsigned long long* d = pce_dbg_buffer();
    MK_REQUIRE(d, "MK_PCE_DBG is not set");
    MK_REQUIRE(hipMemcpy(out512, d, 8 * 64 * 8, hipMemcpyDeviceToHost) == hipSuccess, "copy failed");
    return 0;
}
