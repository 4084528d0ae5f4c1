// fp32-accurate GEMM engine on the bf16 matrix cores ("bf16x3") for the Legendre contraction (K2 / K3)
// and the dhconv spectral filter (K5) on gfx950.
//
// CDNA4 has no TF32 and its fp32 MFMA runs at 1/16 of the bf16 rate.  Every fp32 operand is therefore
// split EXACTLY into three bf16 pieces  x = h + m + l  (8 + 8 + 8 significand bits, by truncation) and a
// product is evaluated as the six piece products of weight >= 2^-16
//     a*b ~= ah*bh + ah*bm + am*bh + ah*bl + am*bm + al*bh          (dropped terms <= 3 * 2^-24 |a b|)
// with v_mfma_f32_32x32x16_bf16 accumulating in fp32: fp32-level accuracy at 16/6 = 2.7x the fp32-MFMA
// rate.  The 1e-5 parity budget of the spectral path is met with two orders of magnitude to spare
// (tests/test_kernels_gpu.py compares against the float64 oracle).
//
// One 256-thread workgroup (4 waves, 2 x 2) owns a 128 x 128 tile of C and walks the contraction in steps
// of 32.  LDS holds ONE stage: per operand 128 rows x [3 pieces][32 k] bf16 (192 B, pitch 208 B so the
// ds_read_b128 fragment reads and the ds_write_b128 staging writes are bank-conflict free); the next
// k-step is prefetched into registers while the MFMAs run and converted / written after a barrier
// (53.5 KB -> 3 workgroups per CU cover each other's staging phases).  Measured alternatives: a register ring
// 2-4 k-steps deep for the streamed operand (2 workgroups per CU) and a 512-thread 256 x 128 tile with two
// LDS stages (1 per CU) were both slower.  Alone, the load stream of the full-resolution analysis takes 0.22 ms
// (~43 GB/s per CU for its HBM / L2 mix) and the MFMA phase 0.19 ms; the two barriers per k-step serialise them inside
// a workgroup and the three workgroups per CU recover about two thirds of the overlap (0.31 ms).  Shrinking the panel
// bytes by a third (fp32 tiles, MK_X3_TABLE=f32) moves the total by 2 %, and a half-step pipeline (refill k 0..15 of
// the stage while the MFMAs read k 16..31, barriers that wait on nothing) left Legendre unchanged and cost dhconv
// 5-10 %: neither bytes nor the barrier placement is the limit.
// (Found in round 2: the stagers' gload used to finish with `valid ? loaded : 0` selects -- a use of the loaded value, so the
// compiler waited out the whole load latency right after issuing the loads, BEFORE the MFMAs of the current k-step, in all
// five kernels.  The masks now travel to sstore; the loads fly under the matrix work: dhconv forward 0.308 -> 0.269 ms.
// Tried on top: a ring two k-steps deep (MK_X3_DB=2: Legendre 0.29 -> 0.42 ms) and producer / consumer workgroups -- four
// staging waves, four multiplying waves, two LDS stages, one barrier per k-step, one workgroup per CU: 0.317 ms.)
//
// Operands whose contraction index is the slow memory axis (k-major rows, n contiguous) are transposed
// in the staging pass: a thread loads 8 consecutive k of two adjacent columns (float2 per row, 512 B per
// wave and row), splits them and writes one 16-byte [8 k] vector per piece and column.  Constant operands
// (the Legendre tables) are pre-split once into the exact LDS image, tile by tile, so staging them is a
// straight 24 KB copy.
#include "common.h"
#include "../../include/makani_amd.h"

#include <cstdint>
#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));   // native vector: HIP uint4 arrays end up in scratch

#ifndef MK_X3_DB
#define MK_X3_DB 1
#endif
constexpr int X3_DB = MK_X3_DB;               // register-ring depth of the streamed (B) operand, in k-steps
constexpr int XT = 256;                       // threads
constexpr int XM = 128, XN = 128, XK = 32;    // workgroup tile, k-step
constexpr int XPITCH = 208;                   // LDS row pitch: 192 data + 16 pad (13 x 16 B: odd)
constexpr int XROWB = 192;                    // bytes of one row per k-step: [3][32] bf16
constexpr int XIMG = XM * XPITCH + 128;       // one operand image (+128: offset of the odd half, see pair_off)
constexpr int X3_LDS = 2 * XIMG;              // 53,504 B

// LDS row offsets.  plain: row r at r * pitch.  pair: rows 2t, 2t+1 are written by one thread (lane t), so
// they live 64 rows (+128 B) apart -- 8 consecutive lanes then hit 8 consecutive rows (conflict-free
// stores) and a fragment read of 16 consecutive rows still covers all 64 banks.
__device__ __forceinline__ int plain_off(int r) { return r * XPITCH; }
__device__ __forceinline__ int pair_off(int r) { return ((r >> 1) + ((r & 1) << 6)) * XPITCH + ((r & 1) << 7); }

// exact three-way split; the bf16 pieces are the UPPER halves of the returned words
struct Split3 {
    uint32_t h, m, l;
};
__device__ __forceinline__ Split3 split3(float x) {
    Split3 s;
    s.h = __float_as_uint(x) & 0xFFFF0000u;
    const float r1 = x - __uint_as_float(s.h);
    s.m = __float_as_uint(r1) & 0xFFFF0000u;
    s.l = __float_as_uint(r1 - __uint_as_float(s.m));   // <= 8 significant bits left: truncation is exact
    return s;
}
// (upper half of e1) : (upper half of e0)
__device__ __forceinline__ uint32_t pack_hi(uint32_t e0, uint32_t e1) { return __builtin_amdgcn_perm(e1, e0, 0x07060302u); }

// split 8 consecutive-k values and store them as one 16-byte vector per piece at `dst` (+64 B per piece)
__device__ __forceinline__ void split_store8(const float (&v)[8], char* dst) {
    Split3 s[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] = split3(v[i]);
    uint4 h, m, l;
    h.x = pack_hi(s[0].h, s[1].h); h.y = pack_hi(s[2].h, s[3].h); h.z = pack_hi(s[4].h, s[5].h); h.w = pack_hi(s[6].h, s[7].h);
    m.x = pack_hi(s[0].m, s[1].m); m.y = pack_hi(s[2].m, s[3].m); m.z = pack_hi(s[4].m, s[5].m); m.w = pack_hi(s[6].m, s[7].m);
    l.x = pack_hi(s[0].l, s[1].l); l.y = pack_hi(s[2].l, s[3].l); l.z = pack_hi(s[4].l, s[5].l); l.w = pack_hi(s[6].l, s[7].l);
    *reinterpret_cast<uint4*>(dst) = h;
    *reinterpret_cast<uint4*>(dst + 64) = m;
    *reinterpret_cast<uint4*>(dst + 128) = l;
}

// Branch-free masked loads: a raw buffer load whose lane offset lies past the descriptor's range returns zeros, so an
// invalid lane simply gets the offset X3_OOB -- no exec-mask branch around the load, no `valid ? loaded : 0` select after it
// (a use of the loaded value: the compiler then waits out the load latency on the spot, before the MFMAs of the k-step) and
// no copy out of a conditionally loaded register (same effect).  The loads of a k-step are issued back to back and are
// waited for where the LDS-staging step reads them.  Offsets are bytes from the stager's base pointer (< 2^31).
typedef unsigned int x3_u4 __attribute__((ext_vector_type(4)));
typedef unsigned int x3_u2 __attribute__((ext_vector_type(2)));
constexpr unsigned X3_OOB = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t x3_rsrc(const float* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7FFFFFFF, 0x00020000);
}
__device__ __forceinline__ float4 x3_load16(__amdgpu_buffer_rsrc_t rs, unsigned off) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
}
__device__ __forceinline__ float2 x3_load8(__amdgpu_buffer_rsrc_t rs, unsigned off) {
    return __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, 0));
}

// ---------------------------------------------------------------------------
// Stagers: Regs, gload(kt, regs, tid), sstore(regs, image, tid), row_off(r)
// ---------------------------------------------------------------------------

__device__ __forceinline__ int quad_row(int lane);

// The constant operand (Legendre table) as fp32 tiles ([128 rows][32 k] floats = 16 KB per k-step), split into the bf16x3
// pieces while staging.  (A pre-split bf16x3 image -- 24 KB per k-step copied straight into LDS -- was built and measured
// 2 % slower in isolation and 0 - 0.2 ms per step slower, with one half more table bytes; removed in round 3.)
struct F32TileStager {
    const char* base;
    typedef float4 Regs[4];
    static constexpr int TILE_BYTES = XM * XK * 4;
    static __device__ __forceinline__ int row_off(int r) { return plain_off(r); }
    __device__ __forceinline__ void gload(int kt, Regs& r, int tid) const {
        const float* p = reinterpret_cast<const float*>(base + (long long)kt * TILE_BYTES);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int t = tid + q * XT, row = (t >> 6) * 16 + quad_row(t & 63);
            const float* pr = p + row * XK + (t & 3) * 8;
            r[2 * q] = *reinterpret_cast<const float4*>(pr);
            r[2 * q + 1] = *reinterpret_cast<const float4*>(pr + 4);
        }
    }
    __device__ __forceinline__ void sstore(const Regs& r, char* img, int tid) const {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int t = tid + q * XT, row = (t >> 6) * 16 + quad_row(t & 63);
            const float v[8] = {r[2 * q].x, r[2 * q].y, r[2 * q].z, r[2 * q].w, r[2 * q + 1].x, r[2 * q + 1].y, r[2 * q + 1].z, r[2 * q + 1].w};
            split_store8(v, img + row * XPITCH + (t & 3) * 16);
        }
    }
};

// Dead rows / columns of an image may hold anything: they only feed outputs that are never stored.  Indices
// past the contraction range must read as zero.

// k-major fp32 operand: element (k, c) = base[k * ldk + c]; tile rows are the 128 columns c (pairs 2t, 2t+1
// per lane t), valid for k_lo <= k < k_hi and c < cvalid (cvalid even).
struct TransStager {
    const float* base;
    long long ldk;
    int k_lo, k_hi, cvalid;
    typedef float2 Regs[8];
    static __device__ __forceinline__ int row_off(int r) { return pair_off(r); }
    __device__ __forceinline__ void gload(int kt, Regs& r, int tid) const {
        const int w = tid >> 6, c = (tid & 63) * 2;
        const int k0 = kt * XK + w * 8;
        // the descriptor is rebased to the first row of the k-step (a 64-bit pointer), so the 32-bit offsets only span the
        // 32 rows of the step: operands of any size (k-major Fourier rows of a large batch pass 2^31 bytes: 721 x 241 x 384
        // channels x 8 B = 534 MB per sample) stay addressable; the launchers require 32 * ldk * 4 < 2^31
        const __amdgpu_buffer_rsrc_t rs = x3_rsrc(base + (long long)kt * XK * ldk);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = k0 + i;
            const bool ok = k >= k_lo && k < k_hi && c < cvalid;
            r[i] = x3_load8(rs, ok ? (unsigned)(((long long)(w * 8 + i) * ldk + c) * 4) : X3_OOB);
        }
    }
    __device__ __forceinline__ void sstore(const Regs& r, char* img, int tid) const {
        const int w = tid >> 6, t = tid & 63;
        float a[8], b[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            a[i] = r[i].x;
            b[i] = r[i].y;
        }
        split_store8(a, img + t * XPITCH + w * 16);
        split_store8(b, img + (t + 64) * XPITCH + 128 + w * 16);
    }
};

// Lane -> (row within the wave's 16 rows, 16-byte chunk) for loaders that give a row to four consecutive lanes
// (128 contiguous bytes of global memory per row).  ds_write_b128 is serviced in groups of 8 consecutive lanes on
// 32 banks: the two rows of a group must lie 4 rows (4 * 52 = 16 banks mod 32) apart, not 1 (20 banks: the first
// chunk of the second row lands on the banks of the last chunk of the first -- measured as 33 % LDS conflict cycles
// in dhconv_dgrad, `profiles/r01_pmc_util.json`).
__device__ __forceinline__ int quad_row(int lane) { return ((lane >> 3) & 3) + 8 * (lane >> 5) + 4 * ((lane >> 2) & 1); }

// Row-major fp32 A operand, k contiguous: element (r, k) = base[r * ld + k], rows < rows, k < kvalid
// (kvalid a multiple of 4, rows 16-byte aligned).  Two (row, 8 k) tasks per thread.
struct RowStager {
    const float* base;
    long long ld;
    int rows, kvalid;
    typedef float4 Regs[4];
    static __device__ __forceinline__ int row_off(int r) { return plain_off(r); }
    __device__ __forceinline__ void gload(int kt, Regs& r, int tid) const {
        const __amdgpu_buffer_rsrc_t rs = x3_rsrc(base);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int t = tid + q * XT, row = (t >> 6) * 16 + quad_row(t & 63), k = kt * XK + (t & 3) * 8;
            const unsigned off = (unsigned)(((long long)row * ld + k) * 4);
#pragma unroll
            for (int h = 0; h < 2; ++h)
                r[2 * q + h] = x3_load16(rs, (row < rows && k + 4 * h < kvalid) ? off + 16 * h : X3_OOB);
        }
    }
    __device__ __forceinline__ void sstore(const Regs& r, char* img, int tid) const {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int t = tid + q * XT, row = (t >> 6) * 16 + quad_row(t & 63);
            const float v[8] = {r[2 * q].x, r[2 * q].y, r[2 * q].z, r[2 * q].w, r[2 * q + 1].x, r[2 * q + 1].y, r[2 * q + 1].z, r[2 * q + 1].w};
            if (row < rows) split_store8(v, img + row * XPITCH + (t & 3) * 16);
        }
    }
};

// Complex k-major B operand: element (kk, o) = (base[(kk * ldk + o) * 2], base[... + 1]), kk < kk_hi,
// o < ovalid.  Complex column o becomes the image rows n = 2o (real part of the product) and 2o + 1
// (imaginary part); complex row kk the contraction indices k = 2kk, 2kk + 1:
//   CONJ_B = false (dhconv forward, B = w):       row 2o: [ re, -im ]   row 2o+1: [ im,  re ]
//   CONJ_B = true  (dhconv wgrad,  B = gy, the conjugate sits on the A side): row 2o: [ re, im ]  row 2o+1: [ im, -re ]
template <bool CONJ_B>
struct CplxStager {
    const float* base;
    long long ldk;
    int kk_hi, ovalid;
    typedef float2 Regs[4];
    static __device__ __forceinline__ int row_off(int r) { return pair_off(r); }
    __device__ __forceinline__ void gload(int kt, Regs& r, int tid) const {
        const int w = __builtin_amdgcn_readfirstlane(tid >> 6), o = tid & 63;
        const int kk0 = kt * (XK / 2) + w * 4;
        const bool ook = o < ovalid;
        const __amdgpu_buffer_rsrc_t rs = x3_rsrc(base);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            r[i] = x3_load8(rs, (ook && kk0 + i < kk_hi) ? (unsigned)((((long long)(kk0 + i)) * ldk + o) * 8) : X3_OOB);
    }
    __device__ __forceinline__ void sstore(const Regs& r, char* img, int tid) const {
        const int w = tid >> 6, t = tid & 63;
        const bool ook = t < ovalid;            // columns past the end loaded column 0: zero them here
        float a[8], b[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float re = ook ? r[i].x : 0.f, im = ook ? r[i].y : 0.f;
            a[2 * i] = re;
            a[2 * i + 1] = CONJ_B ? im : -im;
            b[2 * i] = im;
            b[2 * i + 1] = CONJ_B ? -re : re;
        }
        split_store8(a, img + t * XPITCH + w * 16);
        split_store8(b, img + (t + 64) * XPITCH + 128 + w * 16);
    }
};

// dhconv dgrad B operand: gx[(i,c)] = sum_(o,d) gy[(o,d)] * B[(o,d)][(i,c)] with B = conj(w)^T.  Complex
// w[i][o] at base[(i * O + o) * 2]; image rows n = 2i (-> real part), 2i + 1 (-> imaginary part), contraction
// k = 2o + d:   row 2i: [ re, im ]   row 2i+1: [ -im, re ].   A thread loads 4 consecutive o of one i.
struct DgradStager {
    const float* base;
    int O, ivalid;
    typedef float4 Regs[2];
    static __device__ __forceinline__ int row_off(int r) { return pair_off(r); }
    __device__ __forceinline__ void gload(int kt, Regs& r, int tid) const {
        const int i = (tid >> 6) * 16 + quad_row(tid & 63), o = kt * (XK / 2) + (tid & 3) * 4;
        const __amdgpu_buffer_rsrc_t rs = x3_rsrc(base);
        const unsigned off = (unsigned)(((long long)i * O + o) * 8);
#pragma unroll
        for (int h = 0; h < 2; ++h) r[h] = x3_load16(rs, (i < ivalid && o + 2 * h < O) ? off + 16 * h : X3_OOB);
    }
    __device__ __forceinline__ void sstore(const Regs& r, char* img, int tid) const {
        const int i = (tid >> 6) * 16 + quad_row(tid & 63), c = tid & 3;
        const float a[8] = {r[0].x, r[0].y, r[0].z, r[0].w, r[1].x, r[1].y, r[1].z, r[1].w};
        const float b[8] = {-r[0].y, r[0].x, -r[0].w, r[0].z, -r[1].y, r[1].x, -r[1].w, r[1].z};
        split_store8(a, img + i * XPITCH + c * 16);
        split_store8(b, img + (i + 64) * XPITCH + 128 + c * 16);
    }
};

// dhconv wgrad A operand: gw[i][(o,d)] = sum_(r,c) A[i][(r,c)] * B[(r,c)][(o,d)] with A[i][(r,0)] = re x[r][i],
// A[i][(r,1)] = im x[r][i] (the conjugate is in CplxStager<true>'s signs).  Complex x[r][i] at
// base[(r * I + i) * 2]; image row i holds, per 4 rows r, [re, im] x 4.  Two columns i per thread.
struct WgradAStager {
    const float* base;
    long long I;
    int r_hi, ivalid;
    typedef float2 Regs[8];
    static __device__ __forceinline__ int row_off(int r) { return plain_off(r); }
    __device__ __forceinline__ void gload(int kt, Regs& r, int tid) const {
        const int w = __builtin_amdgcn_readfirstlane(tid >> 6), t = tid & 63;
        const int r0 = kt * (XK / 2) + w * 4;
        const __amdgpu_buffer_rsrc_t rs = x3_rsrc(base);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int i = t + 64 * e;
            const bool iok = i < ivalid;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                r[4 * e + j] = x3_load8(rs, (iok && r0 + j < r_hi) ? (unsigned)((((long long)(r0 + j)) * I + i) * 8) : X3_OOB);
        }
    }
    __device__ __forceinline__ void sstore(const Regs& r, char* img, int tid) const {
        const int w = tid >> 6, t = tid & 63;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float z = (t + 64 * e) < ivalid ? 1.f : 0.f;      // columns past the end loaded column 0: zero them here
            const float v[8] = {z * r[4 * e].x, z * r[4 * e].y, z * r[4 * e + 1].x, z * r[4 * e + 1].y,
                                z * r[4 * e + 2].x, z * r[4 * e + 2].y, z * r[4 * e + 3].x, z * r[4 * e + 3].y};
            split_store8(v, img + (t + 64 * e) * XPITCH + w * 16);
        }
    }
};

// One k-step (2 x k16) of a wave's 64 x 64 sub-tile from the LDS images: six piece products, smallest first,
// alternating between the two accumulators of a 32-row band.  A band whose rows are all past the valid
// extent gets no MFMA work (wave-uniform test).
__device__ __forceinline__ void x3_mfma_step(const char* As, const char* Bs, const int (&a_off)[2], const int (&b_off)[2],
                                             const bool (&live)[2], f32x16 (&acc)[2][2]) {
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        bf16x8 bf[2][3];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
#if defined(MK_X3_NOREAD)      // experiment: operands from registers, no LDS traffic in the step
                typedef int x3_i4 __attribute__((ext_vector_type(4)));
                x3_i4 fake = {b_off[b] + p, s, b, p};
                asm volatile("" : "+v"(fake));
                bf[b][p] = __builtin_bit_cast(bf16x8, fake);
#else
                bf[b][p] = *reinterpret_cast<const bf16x8*>(Bs + b_off[b] + p * 64 + s * 32);
#endif
            }
#pragma unroll
        for (int a = 0; a < 2; ++a)
            if (live[a]) {
                bf16x8 af[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) {
#if defined(MK_X3_NOREAD)
                    typedef int x3_i4 __attribute__((ext_vector_type(4)));
                    x3_i4 fake = {a_off[a] + p, s, a, p};
                    asm volatile("" : "+v"(fake));
                    af[p] = __builtin_bit_cast(bf16x8, fake);
#else
                    af[p] = *reinterpret_cast<const bf16x8*>(As + a_off[a] + p * 64 + s * 32);
#endif
                }
#pragma unroll
                for (int t = 0; t < 6; ++t)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
#if defined(MK_X3_SKIP25)      // experiment (wrong results): a quarter of the MFMAs gone -- what a 3-multiplication complex product could gain at best
                        if (s == 1 && b == 1) continue;
#endif
#if defined(MK_X3_NOMFMA)      // experiment: the LDS reads stay, the matrix instruction becomes one VALU op
                        acc[a][b][t] += __builtin_bit_cast(float4, af[PA[t]]).x * __builtin_bit_cast(float4, bf[b][PB[t]]).y;
#elif defined(MK_X3_AGPR)      // experiment: accumulators in the AccVGPR half of the register file
                        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[a][b]) : "v"(af[PA[t]]), "v"(bf[b][PB[t]]));
#else
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PA[t]], bf[b][PB[t]], acc[a][b], 0, 0, 0);
#endif
                    }
            }
    }
}

// ---------------------------------------------------------------------------
// The tile: C[128 x 128] (+)= A * B over k-steps [kt0, kt1)
// ---------------------------------------------------------------------------
// Accumulator (a, b) of wave (wr, wc) holds rows wr*64 + a*32 + [0,32) and the columns of parity b of
// wc*64 + [0,64) (column 2j + b in MFMA column j), so a lane owns adjacent column pairs and the epilogue
// writes 8-byte values, 256 B per row and wave.
// The B operand streams from HBM: it is prefetched DB k-steps ahead through a ring of register sets (the
// loop is unrolled DB times so ring slots are compile-time); the A operand (L2-resident panels) one ahead.
// EPI: 0 = store (nontemporal), 1 = C += tile (read-modify-write by the one workgroup that owns the tile), 2 = atomic adds
// (several workgroups contract disjoint k ranges into one tile), 3 = store act(tile + bias[row]) with `rowbias` pointing at the
// tile's first row (null: no bias) and `act` != 0 the exact (erf) GELU: conv + bias + activation of layers.py:158-206 in one launch
template <int DB, class AS, class BS, int EPI = 0>
__device__ __forceinline__ void x3_tile(const AS& as, const BS& bs, int kt0, int kt1, int rvalid, int cvalid, float* cbase,
                                        long long ldc, char* lds, int exp = 0, const float* rowbias = nullptr, int act = 0) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: liveness tests stay scalar
    const int wr = wave >> 1, wc = wave & 1;
    const int fi = lane & 31, kg = lane >> 5;
    char* As = lds;
    char* Bs = lds + XIMG;

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    // The accumulators live in the AccVGPR half of the register file (the empty asm makes the compiler select the AGPR form
    // of the MFMAs, as the vendor GEMM libraries do).  With arch-VGPR accumulators this kernel corrupted LDS-exchange kernels
    // that shared its CUs -- rocFFT's and this package's FFT rows, 16 lanes x one register at a time -- whenever another stream
    // or another process ran them at the same moment (tools/ab/share_stress.py torch_fft@mk_dhconv: 300 of 300 rocFFT results
    // wrong next to the VGPR form, 0 of 300 next to this one; profiles/r03_share_stress.txt).
#if !defined(MK_X3_VGPR_ACC)
    asm volatile("" : "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(acc[1][0]), "+a"(acc[1][1]));
#endif

    bool live[2];
    int a_off[2], b_off[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        a_off[a] = AS::row_off(wr * 64 + a * 32 + fi) + kg * 16;
        b_off[a] = BS::row_off(wc * 64 + 2 * fi + a) + kg * 16;
        live[a] = (wr * 64 + a * 32 < rvalid) && (wc * 64 < cvalid);
    }

    typename AS::Regs ra;
    typename BS::Regs rb[DB];
    if (kt0 < kt1) {
        as.gload(kt0, ra, tid);
        bs.gload(kt0, rb[0], tid);
        as.sstore(ra, As, tid);
        bs.sstore(rb[0], Bs, tid);
        __syncthreads();
        if (kt0 + 1 < kt1) as.gload(kt0 + 1, ra, tid);
#pragma unroll
        for (int j = 0; j < DB; ++j)
            if (kt0 + 1 + j < kt1) bs.gload(kt0 + 1 + j, rb[j], tid);
    }
    for (int ktb = kt0; ktb < kt1; ktb += DB) {
#pragma unroll
        for (int j = 0; j < DB; ++j) {   // ring slot j holds k-step kt + 1
            const int kt = ktb + j;
            if (kt >= kt1) break;
            if (!(exp & 4)) x3_mfma_step(As, Bs, a_off, b_off, live, acc);
            if (kt + 1 < kt1) {
                __syncthreads();
                if (!(exp & 2)) {
                    as.sstore(ra, As, tid);
                    bs.sstore(rb[j], Bs, tid);
                }
                __syncthreads();
                if (!(exp & 1)) {
                    if (kt + 2 < kt1 && !(exp & 8)) as.gload(kt + 2, ra, tid);
                    if (kt + 1 + DB < kt1 && !(exp & 16)) bs.gload(kt + 1 + DB, rb[j], tid);
                }
            }
        }
    }
    // C/D map: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const int col = wc * 64 + 2 * fi;
    if (col < cvalid) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wr * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
                if (row < rvalid) {
                    if constexpr (EPI == 1) {
                        float2* d = reinterpret_cast<float2*>(cbase + (long long)row * ldc + col);
                        const float2 o = *d;
                        *d = make_float2(o.x + acc[a][0][r], o.y + acc[a][1][r]);
                        continue;
                    } else if constexpr (EPI == 2) {
                        atomicAdd(cbase + (long long)row * ldc + col, acc[a][0][r]);
                        if (col + 1 < cvalid) atomicAdd(cbase + (long long)row * ldc + col + 1, acc[a][1][r]);
                        continue;
                    } else if constexpr (EPI == 3) {
                        const float bv = rowbias ? rowbias[row] : 0.f;
                        float v0 = acc[a][0][r] + bv, v1 = acc[a][1][r] + bv;
                        if (act) {
                            v0 = 0.5f * v0 * (1.f + erff(v0 * 0.70710678118654752440f));
                            v1 = 0.5f * v1 * (1.f + erff(v1 * 0.70710678118654752440f));
                        }
                        *reinterpret_cast<float2*>(cbase + (long long)row * ldc + col) = make_float2(v0, v1);
                        continue;
                    }
#ifndef MK_X3_PLAIN_STORE
                    // nontemporal: the streamed output does not push the re-used panels (Legendre tile images, dhconv operands)
                    // out of L2.  Isolated launches, same box: Legendre 0.120 / 0.107 -> 0.109 / 0.094 ms at 240 latitudes,
                    // 0.288 / 0.289 -> 0.283 / 0.272 at 721; dhconv wgrad 0.211 -> 0.203; dhconv forward / dgrad unchanged.
                    // (Nontemporal LOADS of the streamed operand were measured too: 8-13 % slower.)
                    typedef float x3_f2 __attribute__((ext_vector_type(2)));
                    x3_f2 v2;
                    v2[0] = acc[a][0][r];
                    v2[1] = acc[a][1][r];
                    __builtin_nontemporal_store(v2, reinterpret_cast<x3_f2*>(cbase + (long long)row * ldc + col));
#else
                    *reinterpret_cast<float2*>(cbase + (long long)row * ldc + col) = make_float2(acc[a][0][r], acc[a][1][r]);
#endif
                }
            }
    }
}

// block -> (batch, tile_m, tile_n): all tiles of one batch index on one XCD (blockIdx % 8), back to back
struct TileId {
    int batch, tm, tn;
    bool valid;
};
__device__ __forceinline__ TileId decode_block(int nbatch, int tiles_m, int tiles_n) {
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q = bid >> 3;
    const int T = tiles_m * tiles_n;
    TileId t;
    t.batch = (q / T) * 8 + xcd;
    const int r = q % T;
    t.tm = r / tiles_n;
    t.tn = r - t.tm * tiles_n;
    t.valid = t.batch < nbatch;
    return t;
}
static inline long long grid_blocks(int nbatch, int tiles_m, int tiles_n) {
    return (long long)mk::ceil_div(nbatch, 8) * 8 * tiles_m * tiles_n;
}

static int x3_exp() {
    static const int v = [] { const char* e = getenv("MK_X3_EXP"); return e ? atoi(e) : 0; }();
    return v;
}

// ---------------------------------------------------------------------------
// Legendre analysis / synthesis
// ---------------------------------------------------------------------------
struct LegX3Params {
    const float* src;
    const char* tab;   // pre-split table (mk_legendre_x3_split)
    float* dst;
    int K, L, Mloc, m_off, N2;
    int RT;   // analysis layout: 128-row tiles per m (rows l = m + 128 rt + r);  synthesis layout: k tiles
    int KC;   // analysis layout: 32-k chunks;                                   synthesis layout: 32-l chunks
    int tiles_n;
    int kmajor;   // layout of the Fourier rows: 0 = xf[m][k][:], 1 = xf[k][m][:] (latitude major, distributed SHT)
    int exp;  // ablation switches (MK_X3_EXP): 1 no prefetch loads, 2 no staging stores, 4 no MFMAs -- wrong results
};

// c[l][m][:] = sum_k W[m][l][k] xf[m][k][:]   (rows l >= m only)
template <class AS>
__global__ __launch_bounds__(XT, 3) void legendre_fwd_x3_kernel(LegX3Params p) {
    extern __shared__ __attribute__((aligned(16))) char lds_x3[];
    const TileId t = decode_block(p.Mloc, p.RT, p.tiles_n);
    if (!t.valid) return;
    const int m = t.batch, mg = p.m_off + m;
    const int l0 = mg + t.tm * XM;
    if (l0 >= p.L) return;
    const int n0 = t.tn * XN;
    AS as;
    as.base = p.tab + ((long long)mg * p.RT + t.tm) * p.KC * AS::TILE_BYTES;
    TransStager bs;
    bs.base = p.src + (p.kmajor ? (long long)m * p.N2 : (long long)m * p.K * p.N2) + n0;
    bs.ldk = p.kmajor ? (long long)p.Mloc * p.N2 : (long long)p.N2;
    bs.k_lo = 0;
    bs.k_hi = p.K;
    bs.cvalid = p.N2 - n0;
    x3_tile<X3_DB>(as, bs, 0, p.KC, p.L - l0, p.N2 - n0, p.dst + ((long long)l0 * p.Mloc + m) * p.N2 + n0,
            (long long)p.Mloc * p.N2, lds_x3, p.exp);
}

// xf[m][k][:] = sum_{l >= m} P[m][l][k] c[l][m][:]
template <class AS>
__global__ __launch_bounds__(XT, 3) void legendre_inv_x3_kernel(LegX3Params p) {
    extern __shared__ __attribute__((aligned(16))) char lds_x3[];
    const TileId t = decode_block(p.Mloc, p.RT, p.tiles_n);
    if (!t.valid) return;
    const int m = t.batch, mg = p.m_off + m;
    const int k0 = t.tm * XM;
    if (k0 >= p.K) return;
    const int n0 = t.tn * XN;
    AS as;
    as.base = p.tab + ((long long)mg * p.RT + t.tm) * p.KC * AS::TILE_BYTES;
    TransStager bs;
    bs.base = p.src + (long long)m * p.N2 + n0;
    bs.ldk = (long long)p.Mloc * p.N2;
    bs.k_lo = mg;
    bs.k_hi = p.L;
    bs.cvalid = p.N2 - n0;
    const int kt0 = mg >> 5;
    float* cb = p.kmajor ? p.dst + ((long long)k0 * p.Mloc + m) * p.N2 + n0 : p.dst + ((long long)m * p.K + k0) * p.N2 + n0;
    x3_tile<X3_DB>(as, bs, kt0 < p.KC ? kt0 : p.KC, p.KC, p.K - k0, p.N2 - n0, cb,
               p.kmajor ? (long long)p.Mloc * p.N2 : (long long)p.N2, lds_x3, p.exp);
}

// ---------------------------------------------------------------------------
// dhconv forward / dgrad / wgrad (same contracts as the fp32 kernels of gemm.hip; cin, cout even)
// ---------------------------------------------------------------------------
struct DhX3Params {
    const float* a;   // x (fwd, wgrad) or gy (dgrad)
    const float* b;   // w (fwd, dgrad) or gy (wgrad)
    float* dst;
    int Lloc, Mloc, B, I, O, l_off, m_off, tiles_m, tiles_n, exp;
};

__device__ __forceinline__ int dh_rows(const DhX3Params& p, int l) {
    int nm = p.l_off + l - p.m_off + 1;  // local modes with global m <= global l
    nm = nm < 0 ? 0 : (nm > p.Mloc ? p.Mloc : nm);
    return nm * p.B;
}

// y[l][r][:] = x[l][r][:] * w[l]   (rows r = (m, b) with m <= l)
__global__ __launch_bounds__(XT, 3) void dhconv_fwd_x3_kernel(DhX3Params p) {
    extern __shared__ __attribute__((aligned(16))) char lds_x3[];
    const TileId t = decode_block(p.Lloc, p.tiles_m, p.tiles_n);
    if (!t.valid) return;
    const int l = p.Lloc - 1 - t.batch;  // heaviest degrees first
    const int R = dh_rows(p, l);
    const int r0 = t.tm * XM;
    if (r0 >= R) return;
    const int n0 = t.tn * XN;
    const long long rowbase = (long long)l * p.Mloc * p.B + r0;
    RowStager as;
    as.base = p.a + rowbase * 2 * p.I;
    as.ld = 2 * p.I;
    as.rows = R - r0;
    as.kvalid = 2 * p.I;
    CplxStager<false> bs;
    bs.base = p.b + ((long long)l * p.I * p.O + n0 / 2) * 2;
    bs.ldk = p.O;
    bs.kk_hi = p.I;
    bs.ovalid = p.O - n0 / 2;
    x3_tile<X3_DB>(as, bs, 0, (2 * p.I + XK - 1) / XK, R - r0, 2 * p.O - n0, p.dst + rowbase * 2 * p.O + n0, 2LL * p.O, lds_x3,
                   p.exp);
}

// gx[l][r][:] = gy[l][r][:] * conj(w[l])^T
__global__ __launch_bounds__(XT, 3) void dhconv_dgrad_x3_kernel(DhX3Params p) {
    extern __shared__ __attribute__((aligned(16))) char lds_x3[];
    const TileId t = decode_block(p.Lloc, p.tiles_m, p.tiles_n);
    if (!t.valid) return;
    const int l = p.Lloc - 1 - t.batch;
    const int R = dh_rows(p, l);
    const int r0 = t.tm * XM;
    if (r0 >= R) return;
    const int n0 = t.tn * XN;
    const long long rowbase = (long long)l * p.Mloc * p.B + r0;
    RowStager as;
    as.base = p.a + rowbase * 2 * p.O;
    as.ld = 2 * p.O;
    as.rows = R - r0;
    as.kvalid = 2 * p.O;
    DgradStager bs;
    bs.base = p.b + ((long long)l * p.I + n0 / 2) * p.O * 2;
    bs.O = p.O;
    bs.ivalid = p.I - n0 / 2;
    x3_tile<X3_DB>(as, bs, 0, (2 * p.O + XK - 1) / XK, R - r0, 2 * p.I - n0, p.dst + rowbase * 2 * p.I + n0, 2LL * p.I, lds_x3,
               p.exp);
}

// gw[l][i][:] = sum_r conj(x[l][r][i]) gy[l][r][:]
__global__ __launch_bounds__(XT, 3) void dhconv_wgrad_x3_kernel(DhX3Params p) {
    extern __shared__ __attribute__((aligned(16))) char lds_x3[];
    const TileId t = decode_block(p.Lloc, p.tiles_m, p.tiles_n);
    if (!t.valid) return;
    const int l = p.Lloc - 1 - t.batch;
    const int R = dh_rows(p, l);   // contraction length (may be 0: the gradient of that degree is zero)
    const int i0 = t.tm * XM;
    if (i0 >= p.I) return;
    const int n0 = t.tn * XN;
    const long long rowbase = (long long)l * p.Mloc * p.B;
    WgradAStager as;
    as.base = p.a + (rowbase * p.I + i0) * 2;
    as.I = p.I;
    as.r_hi = R;
    as.ivalid = p.I - i0;
    CplxStager<true> bs;
    bs.base = p.b + (rowbase * p.O + n0 / 2) * 2;
    bs.ldk = p.O;
    bs.kk_hi = R;
    bs.ovalid = p.O - n0 / 2;
    x3_tile<X3_DB>(as, bs, 0, (2 * R + XK - 1) / XK, p.I - i0, 2 * p.O - n0, p.dst + ((long long)l * p.I + i0) * 2 * p.O + n0,
               2LL * p.O, lds_x3, p.exp);
}

// table [M][L][KP] fp32 -> pre-split image.  One thread per (block, row, kk).
__global__ void legendre_x3_split_kernel(const float* __restrict__ tab, void* __restrict__ out, int K, int KP, int L,
                                         int M, int RT, int KC, int inverse, long long total) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int kk = (int)(idx & 31);
    const int r = (int)((idx >> 5) & 127);
    long long blk = idx >> 12;
    const int kc = (int)(blk % KC);
    blk /= KC;
    const int rt = (int)(blk % RT);
    const int m = (int)(blk / RT);
    int l, k;
    if (!inverse) {
        l = m + rt * XM + r;
        k = kc * XK + kk;
    } else {
        k = rt * XM + r;
        l = kc * XK + kk;
    }
    float v = 0.f;
    if (l < L && k < K) v = tab[((long long)m * L + l) * KP + k];
    reinterpret_cast<float*>(out)[idx] = v;   // [block][128 rows][32 k] fp32
}

static void x3_layout(int nlat, int lmax, int inverse, int* RT, int* KC) {
    if (!inverse) {
        *RT = mk::ceil_div(lmax, XM);
        *KC = mk::ceil_div(nlat, XK);
    } else {
        *RT = mk::ceil_div(nlat, XM);
        *KC = mk::ceil_div(lmax, XK);
    }
}

}  // namespace

extern "C" long long mk_legendre_x3_bytes(int nlat, int lmax, int mmax, int inverse) {
    if (nlat <= 0 || lmax <= 0 || mmax <= 0) return 0;
    int RT, KC;
    x3_layout(nlat, lmax, inverse, &RT, &KC);
    return (long long)mmax * RT * KC * F32TileStager::TILE_BYTES;
}

extern "C" int mk_legendre_x3_split(const float* tab, void* out, int nlat, int lmax, int mmax, int inverse, void* stream) {
    MK_REQUIRE(tab && out, "null pointer");
    MK_REQUIRE(nlat > 0 && lmax > 0 && mmax > 0, "bad sizes");
    int RT, KC;
    x3_layout(nlat, lmax, inverse, &RT, &KC);
    const long long total = (long long)mmax * RT * KC * XM * XK;
    const long long nblk = (total + 255) / 256;
    MK_REQUIRE(nblk < 2147483647LL, "grid too large");
    hipLaunchKernelGGL(legendre_x3_split_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, tab,
                       out, nlat, mk_legendre_kpad(nlat), lmax, mmax, RT, KC, inverse, total);
    MK_LAUNCH_CHECK();
    return 0;
}

static int legendre_x3_launch(bool fwd, const float* src, const void* tab, float* dst, int bc, int nlat, int lmax,
                              int mmax_loc, int m_off, int kmajor, hipStream_t st) {
    LegX3Params p;
    p.kmajor = kmajor;
    p.src = src;
    p.tab = (const char*)tab;
    p.dst = dst;
    p.K = nlat;
    p.L = lmax;
    p.Mloc = mmax_loc;
    p.m_off = m_off;
    p.N2 = 2 * bc;
    x3_layout(nlat, lmax, fwd ? 0 : 1, &p.RT, &p.KC);
    p.tiles_n = mk::ceil_div(p.N2, XN);
    p.exp = x3_exp();
    const long long nblk = grid_blocks(mmax_loc, p.RT, p.tiles_n);
    if (nblk >= 2147483647LL) return -1;
    // TransStager: 32-bit byte offsets inside one 32-row k-step of the data operand (row stride Mloc * N2 floats at most)
    if (33LL * p.Mloc * p.N2 * 4 >= (1LL << 31)) return -2;
    const dim3 grid((unsigned)nblk), blk(XT);
    if (fwd) hipLaunchKernelGGL(legendre_fwd_x3_kernel<F32TileStager>, grid, blk, X3_LDS, st, p);
    else hipLaunchKernelGGL(legendre_inv_x3_kernel<F32TileStager>, grid, blk, X3_LDS, st, p);
    return 0;
}

extern "C" int mk_legendre_fwd_x3_ex(const float* xf, const void* tab_x3, float* c, int bc, int nlat, int lmax,
                                     int mmax_loc, int m_off, int mmax_glob, int xf_layout, void* stream) {
    MK_REQUIRE(xf && tab_x3 && c, "null pointer");
    MK_REQUIRE(bc > 0 && nlat > 0 && lmax > 0 && mmax_loc > 0, "bad sizes");
    MK_REQUIRE(m_off >= 0 && m_off + mmax_loc <= mmax_glob, "mode shard out of range");
    MK_REQUIRE(xf_layout == 0 || xf_layout == 1, "xf_layout must be 0 ([M][K][BC]) or 1 ([K][M][BC])");
    MK_REQUIRE(legendre_x3_launch(true, xf, tab_x3, c, bc, nlat, lmax, mmax_loc, m_off, xf_layout, (hipStream_t)stream) == 0,
               "operand too large: grid over 2^31 blocks, or 33 * mmax_loc * 2 * bc * 4 bytes (one k-step of the data operand) over 2^31");
    MK_LAUNCH_CHECK();
    return 0;
}

extern "C" int mk_legendre_fwd_x3(const float* xf, const void* tab_x3, float* c, int bc, int nlat, int lmax, int mmax_loc,
                                  int m_off, int mmax_glob, void* stream) {
    return mk_legendre_fwd_x3_ex(xf, tab_x3, c, bc, nlat, lmax, mmax_loc, m_off, mmax_glob, 0, stream);
}

extern "C" int mk_legendre_inv_x3_ex(const float* c, const void* tab_x3, float* xf, int bc, int nlat, int lmax,
                                     int mmax_loc, int m_off, int mmax_glob, int xf_layout, void* stream) {
    MK_REQUIRE(xf && tab_x3 && c, "null pointer");
    MK_REQUIRE(bc > 0 && nlat > 0 && lmax > 0 && mmax_loc > 0, "bad sizes");
    MK_REQUIRE(m_off >= 0 && m_off + mmax_loc <= mmax_glob, "mode shard out of range");
    MK_REQUIRE(xf_layout == 0 || xf_layout == 1, "xf_layout must be 0 ([M][K][BC]) or 1 ([K][M][BC])");
    MK_REQUIRE(legendre_x3_launch(false, c, tab_x3, xf, bc, nlat, lmax, mmax_loc, m_off, xf_layout, (hipStream_t)stream) == 0,
               "operand too large: grid over 2^31 blocks, or 33 * mmax_loc * 2 * bc * 4 bytes (one k-step of the data operand) over 2^31");
    MK_LAUNCH_CHECK();
    return 0;
}

extern "C" int mk_legendre_inv_x3(const float* c, const void* tab_x3, float* xf, int bc, int nlat, int lmax, int mmax_loc,
                                  int m_off, int mmax_glob, void* stream) {
    return mk_legendre_inv_x3_ex(c, tab_x3, xf, bc, nlat, lmax, mmax_loc, m_off, mmax_glob, 0, stream);
}

static int dh_x3_check(const void* a, const void* b, const void* c, int lloc, int mloc, int batch, int cin, int cout,
                       int l_off, int m_off) {
    MK_REQUIRE(a && b && c, "null pointer");
    MK_REQUIRE(lloc > 0 && mloc > 0 && batch > 0 && cin > 0 && cout > 0, "bad sizes");
    MK_REQUIRE(l_off >= 0 && m_off >= 0, "negative shard offset");
    MK_REQUIRE(cin % 2 == 0 && cout % 2 == 0, "the bf16x3 dhconv kernels need even channel counts (use the fp32 kernels)");
    return 0;
}

extern "C" int mk_dhconv_fwd_x3(const float* x, const float* w, float* y, int lloc, int mloc, int batch, int cin,
                                int cout, int l_off, int m_off, void* stream) {
    if (int e = dh_x3_check(x, w, y, lloc, mloc, batch, cin, cout, l_off, m_off)) return e;
    DhX3Params p{x, w, y, lloc, mloc, batch, cin, cout, l_off, m_off, mk::ceil_div(mloc * batch, XM),
                 mk::ceil_div(2 * cout, XN), x3_exp()};
    const long long nblk = grid_blocks(lloc, p.tiles_m, p.tiles_n);
    MK_REQUIRE(nblk < 2147483647LL, "grid too large");
    hipLaunchKernelGGL(dhconv_fwd_x3_kernel, dim3((unsigned)nblk), dim3(XT), X3_LDS, (hipStream_t)stream, p);
    MK_LAUNCH_CHECK();
    return 0;
}

extern "C" int mk_dhconv_dgrad_x3(const float* gy, const float* w, float* gx, int lloc, int mloc, int batch, int cin,
                                  int cout, int l_off, int m_off, void* stream) {
    if (int e = dh_x3_check(gy, w, gx, lloc, mloc, batch, cin, cout, l_off, m_off)) return e;
    DhX3Params p{gy, w, gx, lloc, mloc, batch, cin, cout, l_off, m_off, mk::ceil_div(mloc * batch, XM),
                 mk::ceil_div(2 * cin, XN), x3_exp()};
    const long long nblk = grid_blocks(lloc, p.tiles_m, p.tiles_n);
    MK_REQUIRE(nblk < 2147483647LL, "grid too large");
    hipLaunchKernelGGL(dhconv_dgrad_x3_kernel, dim3((unsigned)nblk), dim3(XT), X3_LDS, (hipStream_t)stream, p);
    MK_LAUNCH_CHECK();
    return 0;
}

extern "C" int mk_dhconv_wgrad_x3(const float* x, const float* gy, float* gw, int lloc, int mloc, int batch, int cin,
                                  int cout, int l_off, int m_off, void* stream) {
    if (int e = dh_x3_check(x, gy, gw, lloc, mloc, batch, cin, cout, l_off, m_off)) return e;
    DhX3Params p{x, gy, gw, lloc, mloc, batch, cin, cout, l_off, m_off, mk::ceil_div(cin, XM),
                 mk::ceil_div(2 * cout, XN), x3_exp()};
    const long long nblk = grid_blocks(lloc, p.tiles_m, p.tiles_n);
    MK_REQUIRE(nblk < 2147483647LL, "grid too large");
    hipLaunchKernelGGL(dhconv_wgrad_x3_kernel, dim3((unsigned)nblk), dim3(XT), X3_LDS, (hipStream_t)stream, p);
    MK_LAUNCH_CHECK();
    return 0;
}


// ---------------------------------------------------------------------------
// 1x1 convolutions on fp32 fields (nn.Conv2d(.., 1) of MLP / EncoderDecoder / skips outside autocast: layers.py:86-216,
// sfnonet.py:207,379,463) on the same bf16x3 engine: fp32-accurate products without a vendor GEMM.
//   mode 0:  C[b] = A B[b]          A [M][K] row-major (K a multiple of 4), B[b] [K][N] k-major (the NCHW field, N = H*W even)
//   mode 1:  C[b] += A B[b]         (the skip connection folded into the GEMM: C holds the addend)
//   mode 2:  C += sum_b A[b] B[b]^T  A[b] [M][Kp], B[b] [N][Kp] both row-major over the contraction (the pixels): the weight
//            gradient, the pixels cut into slabs of `kslab` k-steps, one workgroup per (tile, slab), fp32 atomics into C (zeroed by the caller)
// ---------------------------------------------------------------------------
namespace {
struct ConvX3Params {
    const float* a;
    const float* b;
    float* c;
    long long lda, ldb, ldc, sa, sb, sc;
    int M, K, N, nwork, tiles_m, tiles_n, nslab, kslab;
    const float* bias;      // mode 3: per-row bias or null
    int act;                // mode 3: 1 = exact GELU
};
template <int MODE>
__global__ __launch_bounds__(XT, 3) void conv_x3_kernel(ConvX3Params p) {
    extern __shared__ __attribute__((aligned(16))) char lds_x3[];
    // MODE 2: work item = (batch item, pixel slab), all tiles of gW of one item on one XCD.  MODE 0 / 1: work item = (batch item,
    // 128-pixel tile), its M / 128 row tiles back to back on one XCD -- they share the x tile, the second to sixth find it in
    // that XCD's L2 (with the batch index as the only work index a batch of one ran on ONE of the eight XCDs: 8x slower)
    TileId t = decode_block(p.nwork, p.tiles_m, MODE == 2 ? p.tiles_n : 1);
    if (!t.valid) return;
    if constexpr (MODE != 2) {
        const int w = t.batch;
        t.batch = w / p.tiles_n;
        t.tn = w - t.batch * p.tiles_n;
    }
    const int m0 = t.tm * XM, n0 = t.tn * XN;
    if constexpr (MODE == 2) {
        const int b = t.batch / p.nslab, slab = t.batch - b * p.nslab;
        const int kt0 = slab * p.kslab, ktn = (p.K + XK - 1) / XK;
        const int kt1 = kt0 + p.kslab < ktn ? kt0 + p.kslab : ktn;
        RowStager as, bs;
        as.base = p.a + b * p.sa + (long long)m0 * p.lda;
        as.ld = p.lda;
        as.rows = p.M - m0;
        as.kvalid = p.K;
        bs.base = p.b + b * p.sb + (long long)n0 * p.ldb;
        bs.ld = p.ldb;
        bs.rows = p.N - n0;
        bs.kvalid = p.K;
        x3_tile<X3_DB, RowStager, RowStager, 2>(as, bs, kt0, kt1, p.M - m0, p.N - n0, p.c + (long long)m0 * p.ldc + n0, p.ldc, lds_x3);
    } else {
        RowStager as;
        as.base = p.a + (long long)m0 * p.lda;
        as.ld = p.lda;
        as.rows = p.M - m0;
        as.kvalid = (p.K + 3) / 4 * 4;         // whole 16-byte groups: the caller pads the rows of A with zeros
        TransStager bs;
        bs.base = p.b + t.batch * p.sb + n0;
        bs.ldk = p.ldb;
        bs.k_lo = 0;
        bs.k_hi = p.K;
        bs.cvalid = p.N - n0;
        x3_tile<X3_DB, RowStager, TransStager, MODE>(as, bs, 0, (p.K + XK - 1) / XK, p.M - m0, p.N - n0,
                                                     p.c + t.batch * p.sc + (long long)m0 * p.ldc + n0, p.ldc, lds_x3, 0,
                                                     (MODE == 3 && p.bias) ? p.bias + m0 : nullptr, p.act);
    }
}
}  // namespace

static int conv_x3_launch(const float* a, long long lda, const float* b, long long ldb, float* c, long long ldc, int M, int K,
                          long long N, int batch, long long sa, long long sb, long long sc, int mode, const float* bias, int act,
                          void* stream) {
    MK_REQUIRE(a && b && c, "null pointer");
    MK_REQUIRE(M > 0 && K > 0 && N > 0 && batch > 0, "bad sizes");
    MK_REQUIRE(mode >= 0 && mode <= 3, "mode must be 0 (store), 1 (accumulate), 2 (weight gradient) or 3 (store with bias / GELU)");
    MK_REQUIRE((((uintptr_t)a | (uintptr_t)b | (uintptr_t)c) & 15) == 0, "operands must be 16-byte aligned");
    ConvX3Params p;
    p.a = a; p.b = b; p.c = c;
    p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.sa = sa; p.sb = sb; p.sc = sc;
    p.M = M; p.K = K;
    p.bias = bias; p.act = act;
    MK_REQUIRE(N < 2147483647LL, "N too large");
    p.N = (int)N;
    p.tiles_m = mk::ceil_div(M, XM);
    p.tiles_n = mk::ceil_div((int)N, XN);
    p.nslab = 1;
    p.kslab = 0;
    if (mode == 2) {
        // A [M][K], B [N][K]: rows start on 16-byte boundaries, whole 4-element groups
        MK_REQUIRE(lda % 4 == 0 && ldb % 4 == 0 && K % 4 == 0 && sa % 4 == 0 && sb % 4 == 0, "weight gradient: row strides and the contraction length must be multiples of 4");
        const int ktn = mk::ceil_div(K, XK);
        // enough workgroups to fill the chip: ~6 per CU
        long long want = 1536 / ((long long)p.tiles_m * p.tiles_n * batch);
        if (want < 1) want = 1;
        if (want > ktn) want = ktn;
        p.kslab = mk::ceil_div(ktn, (int)want);
        p.nslab = mk::ceil_div(ktn, p.kslab);
        p.nwork = batch * p.nslab;
    } else {
        MK_REQUIRE(lda % 4 == 0 && lda >= (K + 3) / 4 * 4, "A: the row stride must be a multiple of 4 and cover K rounded up to 4 (zero padded)");
        MK_REQUIRE(N % 2 == 0 && ldb % 2 == 0 && ldc % 2 == 0 && sb % 2 == 0 && sc % 2 == 0, "B / C: even row lengths and strides");
        MK_REQUIRE(33LL * ldb * 4 < (1LL << 31), "B row stride too large for the 32-bit offsets of one k-step");
        MK_REQUIRE((long long)batch * p.tiles_n < 2147483647LL, "too many pixel tiles");
        p.nwork = batch * p.tiles_n;
    }
    const long long nblk = grid_blocks(p.nwork, p.tiles_m, mode == 2 ? p.tiles_n : 1);
    MK_REQUIRE(nblk < 2147483647LL, "grid too large");
    const dim3 grid((unsigned)nblk), blk(XT);
    hipStream_t st = (hipStream_t)stream;
    if (mode == 0) hipLaunchKernelGGL(conv_x3_kernel<0>, grid, blk, X3_LDS, st, p);
    else if (mode == 1) hipLaunchKernelGGL(conv_x3_kernel<1>, grid, blk, X3_LDS, st, p);
    else if (mode == 2) hipLaunchKernelGGL(conv_x3_kernel<2>, grid, blk, X3_LDS, st, p);
    else hipLaunchKernelGGL(conv_x3_kernel<3>, grid, blk, X3_LDS, st, p);
    MK_LAUNCH_CHECK();
    return 0;
}

extern "C" int mk_conv1x1_x3(const float* a, long long lda, const float* b, long long ldb, float* c, long long ldc, int M, int K,
                             long long N, int batch, long long sa, long long sb, long long sc, int mode, void* stream) {
    MK_REQUIRE(mode >= 0 && mode <= 2, "mode must be 0 (store), 1 (accumulate) or 2 (weight gradient)");
    return conv_x3_launch(a, lda, b, ldb, c, ldc, M, K, N, batch, sa, sb, sc, mode, nullptr, 0, stream);
}

// C[b] = act(A B[b] + bias): the fp32 convolution with its bias add and (act = 1) exact GELU in the epilogue -- what
// `nn.Conv2d(cin, cout, 1, bias=True)` + `nn.GELU()` (layers.py:95-99, 158-206) compute, in one pass over the output.
extern "C" int mk_conv1x1_x3_bias_act(const float* a, long long lda, const float* b, long long ldb, float* c, long long ldc, int M,
                                      int K, long long N, int batch, long long sb, long long sc, const float* bias, int act,
                                      void* stream) {
    MK_REQUIRE(act == 0 || act == 1, "act must be 0 (none) or 1 (exact GELU)");
    return conv_x3_launch(a, lda, b, ldb, c, ldc, M, K, N, batch, 0, sb, sc, 3, bias, act, stream);
}
