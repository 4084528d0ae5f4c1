// fp32 MFMA tile engine for the Legendre contraction (K2 / K3) and the dhconv
// spectral filter (K5) on gfx950.
//
// Every op here is a batch of real GEMMs C = A*B whose operands live in the
// private layouts of include/makani_amd.h.  One 256-thread workgroup (4 waves as
// 2 x 2) owns a 64 x 128 tile of C and walks the contraction dimension in steps of
// 32 through double-buffered LDS.  Both operands are staged k-major
// (As[k][row], Bs[k][col]) so that the v_mfma_f32_32x32x2_f32 fragments
// (A[i = lane&31][k = lane>>5], B[k = lane>>5][j = lane&31]) are read from consecutive
// LDS words: conflict free.  Exact fp32 arithmetic (MFMA f32 == fmaf chain), which
// is what the 1e-5 parity budget needs; no bf16 down-cast anywhere.
//
// The complex dhconv contraction is expressed as a real GEMM with the 2x2 real
// block structure folded into the B-operand staging (the loader writes the
// (re, im) row and its (-im, re) partner), so the MFMA loop is the same for all
// five ops.  Structural zeros are skipped: Legendre tiles start at l = m, dhconv
// tiles stop at m = l.
//
// Block -> tile mapping is XCD aware: all tiles of one batch index (one m, or one l)
// are dealt to the same XCD (blockIdx % 8) back to back, so the operand panels they
// share stay in that XCD's L2.
#include "common.h"
#include "../../include/makani_amd.h"

#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kThreads = 256;
constexpr int TM = 64, TN = 128, TK = 32;
constexpr int WM = 32, WN = 64;  // per-wave tile (waves 2 x 2)
constexpr int NT = WN / 32;
constexpr int LDA_T = TM + 1;  // transposing loader (b32 scatter, conflict free)
constexpr int LDA_D = TM;      // direct loader (vector stores)
constexpr int LDB_D = TN;
constexpr int LDB_T = TN + 2;  // transposing b64 scatter
constexpr int LDA_MAX = LDA_T, LDB_MAX = LDB_T;
constexpr int A_WORDS = TK * LDA_MAX, B_WORDS = TK * LDB_MAX;

__device__ __forceinline__ int mk_ceil_div_dev(int a, int b) { return (a + b - 1) / b; }

template <int VEC> struct VecT;
template <> struct VecT<4> { using type = float4; };
template <> struct VecT<2> { using type = float2; };

template <int VEC>
__device__ __forceinline__ void gload_vec(float (&dst)[VEC], const float* p, bool ok) {
    if (ok) {
        const typename VecT<VEC>::type v = *reinterpret_cast<const typename VecT<VEC>::type*>(p);
        const float* f = reinterpret_cast<const float*>(&v);
#pragma unroll
        for (int i = 0; i < VEC; ++i) dst[i] = f[i];
    } else {
#pragma unroll
        for (int i = 0; i < VEC; ++i) dst[i] = 0.f;
    }
}

template <int VEC>
__device__ __forceinline__ void sstore_vec(float* lds, const float (&src)[VEC]) {
    typename VecT<VEC>::type v;
    float* f = reinterpret_cast<float*>(&v);
#pragma unroll
    for (int i = 0; i < VEC; ++i) f[i] = src[i];
    *reinterpret_cast<typename VecT<VEC>::type*>(lds) = v;
}

// ---------------------------------------------------------------------------
// Tile loaders.  Each has: NV vectors per thread, gload(kt, regs), sstore(regs, lds)
// and LD (LDS row stride in words).
// ---------------------------------------------------------------------------

// Operand tile [TK][COLS], global rows indexed by k, contiguous along the tile's
// row/col index:  elem(k, c) = base[k * ldk + c],  valid if k < kvalid && c < cvalid.
template <int COLS, int VEC, int LD_>
struct DirectLoader {
    static constexpr int LD = LD_;
    static constexpr int VECW = VEC;
    static constexpr int VPR = COLS / VEC;                 // vectors per k-row
    static constexpr int NV = TK * VPR / kThreads;         // vectors per thread
    const float* base;  // element (k = 0, c = 0) of this tile row/col block, k = 0 of the whole K range
    long long ldk;
    int kvalid, cvalid;  // remaining extents from this tile's origin (k counted over the whole range)
    __device__ __forceinline__ void gload(int kt, float (&r)[NV][VEC], int tid) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = tid + i * kThreads;
            const int kk = v / VPR, c = (v - kk * VPR) * VEC;
            const int k = kt * TK + kk;
            gload_vec<VEC>(r[i], base + (long long)k * ldk + c, k < kvalid && c < cvalid);
        }
    }
    __device__ __forceinline__ void sstore(const float (&r)[NV][VEC], float* lds, int tid) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = tid + i * kThreads;
            const int kk = v / VPR, c = (v - kk * VPR) * VEC;
            sstore_vec<VEC>(lds + kk * LD + c, r[i]);
        }
    }
};

// Operand tile [TK][ROWS], global rows indexed by the tile's row index, contiguous
// along k:  elem(k, r) = base[r * ldr + k],  valid if r < rvalid && k < kvalid.
template <int ROWS, int VEC>
struct TransLoader {
    static constexpr int LD = ROWS + 1;
    static constexpr int VECW = VEC;
    static constexpr int VPR = TK / VEC;                  // vectors per global row
    static constexpr int NV = ROWS * VPR / kThreads;
    const float* base;
    long long ldr;
    int rvalid, kvalid;
    __device__ __forceinline__ void gload(int kt, float (&r)[NV][VEC], int tid) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = tid + i * kThreads;
            const int row = v / VPR, kq = (v - row * VPR) * VEC;
            const int k = kt * TK + kq;
            gload_vec<VEC>(r[i], base + (long long)row * ldr + k, row < rvalid && k < kvalid);
        }
    }
    __device__ __forceinline__ void sstore(const float (&r)[NV][VEC], float* lds, int tid) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = tid + i * kThreads;
            const int row = v / VPR, kq = (v - row * VPR) * VEC;
#pragma unroll
            for (int j = 0; j < VEC; ++j) lds[(kq + j) * LD + row] = r[i][j];
        }
    }
};

// dhconv B operand (forward and wgrad): the k-tile holds TK/2 complex rows; global
// row q (complex, COLS real words contiguous) expands to real rows 2q and 2q+1:
//   MODE 0 (fwd,   B' = [[Wre, Wim], [-Wim, Wre]]):  row 2q = (re, im),  row 2q+1 = (-im,  re)
//   MODE 1 (wgrad, B  = [[Gre, Gim], [ Gim,-Gre]]):  row 2q = (re, im),  row 2q+1 = ( im, -re)
template <int COLS, int VEC, int MODE>
struct ComplexRowLoader {
    static constexpr int LD = LDB_D;
    static constexpr int VECW = VEC;
    static constexpr int VPR = COLS / VEC;
    static constexpr int NV = (TK / 2) * VPR / kThreads;
    const float* base;  // complex row 0 of the contraction range, col origin of this tile
    long long ldq;      // words between consecutive complex rows
    int qvalid, cvalid;
    __device__ __forceinline__ void gload(int kt, float (&r)[NV][VEC], int tid) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = tid + i * kThreads;
            const int qq = v / VPR, c = (v - qq * VPR) * VEC;
            const int q = kt * (TK / 2) + qq;
            gload_vec<VEC>(r[i], base + (long long)q * ldq + c, q < qvalid && c < cvalid);
        }
    }
    __device__ __forceinline__ void sstore(const float (&r)[NV][VEC], float* lds, int tid) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = tid + i * kThreads;
            const int qq = v / VPR, c = (v - qq * VPR) * VEC;
            sstore_vec<VEC>(lds + (2 * qq) * LD + c, r[i]);
            float s[VEC];
#pragma unroll
            for (int j = 0; j < VEC; j += 2) {
                if (MODE == 0) {
                    s[j] = -r[i][j + 1];
                    s[j + 1] = r[i][j];
                } else {
                    s[j] = r[i][j + 1];
                    s[j + 1] = -r[i][j];
                }
            }
            sstore_vec<VEC>(lds + (2 * qq + 1) * LD + c, s);
        }
    }
};

// dhconv dgrad B operand: B''[2o+s][2i+t] from W[i][o] (complex, o contiguous):
//   [2o][2i] = re, [2o][2i+1] = -im, [2o+1][2i] = im, [2o+1][2i+1] = re.
// Tile: TK/2 complex o's (contraction) x TN/2 complex i's (columns).  CPV complex
// values per global load (2 -> float4, 1 -> float2).
template <int CPV>
struct ConjTransLoader {
    static constexpr int LD = LDB_T;
    static constexpr int VEC = 2 * CPV;
    static constexpr int VECW = VEC;
    static constexpr int VPR = (TK / 2) / CPV;             // vectors per global row (one i)
    static constexpr int NV = (TN / 2) * VPR / kThreads;
    const float* base;  // W[i0][o = 0] of this l
    long long ldi;      // words between consecutive i rows (2 * O)
    int ivalid, ovalid;
    __device__ __forceinline__ void gload(int kt, float (&r)[NV][VEC], int tid) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = tid + i * kThreads;
            const int ii = v / VPR, oq = (v - ii * VPR) * CPV;
            const int o = kt * (TK / 2) + oq;
            gload_vec<VEC>(r[i], base + (long long)ii * ldi + 2 * o, ii < ivalid && o < ovalid);
        }
    }
    __device__ __forceinline__ void sstore(const float (&r)[NV][VEC], float* lds, int tid) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = tid + i * kThreads;
            const int ii = v / VPR, oq = (v - ii * VPR) * CPV;
#pragma unroll
            for (int j = 0; j < CPV; ++j) {
                const float re = r[i][2 * j], im = r[i][2 * j + 1];
                float* p0 = lds + (2 * (oq + j)) * LD + 2 * ii;
                *reinterpret_cast<float2*>(p0) = make_float2(re, -im);
                *reinterpret_cast<float2*>(p0 + LD) = make_float2(im, re);
            }
        }
    }
};

// dhconv wgrad A operand: A[i][2r+s] = X[r][i].s -> As[2r+s][i]; global row r holds
// complex X[r][i] contiguous over i.  Tile: TK/2 rows r x TM complex columns i.
template <int CPV>
struct DeinterleaveLoader {
    static constexpr int LD = LDA_D;
    static constexpr int VEC = 2 * CPV;
    static constexpr int VECW = VEC;
    static constexpr int VPR = TM / CPV;
    static constexpr int NV = (TK / 2) * VPR / kThreads;
    const float* base;  // X[r = 0][i0]
    long long ldr;
    int rvalid, ivalid;
    __device__ __forceinline__ void gload(int kt, float (&r)[NV][VEC], int tid) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = tid + i * kThreads;
            const int rr = v / VPR, ic = (v - rr * VPR) * CPV;
            const int row = kt * (TK / 2) + rr;
            gload_vec<VEC>(r[i], base + (long long)row * ldr + 2 * ic, row < rvalid && ic < ivalid);
        }
    }
    __device__ __forceinline__ void sstore(const float (&r)[NV][VEC], float* lds, int tid) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = tid + i * kThreads;
            const int rr = v / VPR, ic = (v - rr * VPR) * CPV;
            float re[CPV], im[CPV];
#pragma unroll
            for (int j = 0; j < CPV; ++j) {
                re[j] = r[i][2 * j];
                im[j] = r[i][2 * j + 1];
            }
            if constexpr (CPV == 2) {
                *reinterpret_cast<float2*>(lds + (2 * rr) * LD + ic) = make_float2(re[0], re[1]);
                *reinterpret_cast<float2*>(lds + (2 * rr + 1) * LD + ic) = make_float2(im[0], im[1]);
            } else {
                lds[(2 * rr) * LD + ic] = re[0];
                lds[(2 * rr + 1) * LD + ic] = im[0];
            }
        }
    }
};

// ---------------------------------------------------------------------------
// main loop + epilogue
// ---------------------------------------------------------------------------
__constant__ int g_exp = 0;   // tuning experiments only (MK_GEMM_EXP); 0 in production

struct Epilogue {
    float* base;       // C element (row 0, col 0) of this tile
    long long ldc;     // words between consecutive C rows
    int rvalid, cvalid;
};

template <class ALoad, class BLoad>
__device__ __forceinline__ void gemm_tile(const ALoad& al, const BLoad& bl, int nk, const Epilogue& ep, float* lds) {
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;
    const int li = lane & 31, lk = lane >> 5;
    const bool wave_live = (__builtin_amdgcn_readfirstlane(wm0) < ep.rvalid) &&
                           (__builtin_amdgcn_readfirstlane(wn0) < ep.cvalid);
    float* As = lds;                   // [2][A_WORDS]
    float* Bs = lds + 2 * A_WORDS;     // [2][B_WORDS]

    f32x16 acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;

    // register staging for the next k-tile
    float rA[ALoad::NV][ALoad::VECW];
    float rB[BLoad::NV][BLoad::VECW];

    if (nk > 0) {
        al.gload(0, rA, tid);
        bl.gload(0, rB, tid);
        al.sstore(rA, As, tid);
        bl.sstore(rB, Bs, tid);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < nk;
        if (more && !(g_exp & 1)) {
            al.gload(kt + 1, rA, tid);
            bl.gload(kt + 1, rB, tid);
        }
        const float* a_ = As + cur * A_WORDS + lk * ALoad::LD + wm0 + li;
        const float* b_ = Bs + cur * B_WORDS + lk * BLoad::LD + wn0 + li;
        // ragged tiles: a wave whose 32 rows (or 64 columns) are all past the valid extent does no MFMA
        // work (it still stages operands and meets the barriers) -> the matrix pipe only sees real rows
        if (wave_live) {
            // fragments of k-step ks+1 are fetched from LDS before the MFMAs of step ks issue
            float a_c = a_[0], b_c[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) b_c[n] = b_[n * 32];
#pragma unroll
            for (int ks = 0; ks < TK / 2; ++ks) {
                float a_n = 0.f, b_n[NT];
                if (ks + 1 < TK / 2) {
                    a_n = a_[(2 * ks + 2) * ALoad::LD];
#pragma unroll
                    for (int n = 0; n < NT; ++n) b_n[n] = b_[(2 * ks + 2) * BLoad::LD + n * 32];
                }
                __builtin_amdgcn_sched_barrier(0);  // keep the prefetch reads ahead of this step's MFMAs
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_c, b_c[n], acc[n], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                a_c = a_n;
#pragma unroll
                for (int n = 0; n < NT; ++n) b_c[n] = b_n[n];
            }
        }
        if (more && !(g_exp & 2)) {
            al.sstore(rA, As + (cur ^ 1) * A_WORDS, tid);
            bl.sstore(rB, Bs + (cur ^ 1) * B_WORDS, tid);
        }
        if (!(g_exp & 4)) __syncthreads();
    }
    // epilogue: acc reg r -> row (r&3) + 8*(r>>2) + 4*(lane>>5), col lane&31
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int col = wn0 + n * 32 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wm0 + (r & 3) + 8 * (r >> 2) + 4 * lk;
            if (row < ep.rvalid && col < ep.cvalid) ep.base[(long long)row * ep.ldc + col] = acc[n][r];
        }
    }
}


// ---------------------------------------------------------------------------
// block -> (batch, tile_m, tile_n), XCD aware
// ---------------------------------------------------------------------------
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
static inline unsigned grid_blocks(int nbatch, int tiles_m, int tiles_n) {
    return (unsigned)(mk::ceil_div(nbatch, 8) * 8 * tiles_m * tiles_n);
}

// ---------------------------------------------------------------------------
// Legendre analysis / synthesis
// ---------------------------------------------------------------------------
struct LegParams {
    const float* src;
    const float* tab;
    float* dst;
    int K, KP, L, Mloc, m_off, N2, tiles_m, tiles_n;
};

template <int VEC>
__global__ __launch_bounds__(kThreads) void legendre_fwd_kernel(LegParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const TileId t = decode_block(p.Mloc, p.tiles_m, p.tiles_n);
    if (!t.valid) return;
    const int m = t.batch, mg = p.m_off + m;
    const int l0 = mg + t.tm * TM;
    if (l0 >= p.L) return;
    const int n0 = t.tn * TN;
    TransLoader<TM, 4> al;
    al.base = p.tab + ((long long)mg * p.L + l0) * p.KP;
    al.ldr = p.KP;
    al.rvalid = p.L - l0;
    al.kvalid = p.KP;
    DirectLoader<TN, VEC, LDB_D> bl;
    bl.base = p.src + (long long)m * p.K * p.N2 + n0;
    bl.ldk = p.N2;
    bl.kvalid = p.K;
    bl.cvalid = p.N2 - n0;
    Epilogue ep;
    ep.base = p.dst + ((long long)l0 * p.Mloc + m) * p.N2 + n0;
    ep.ldc = (long long)p.Mloc * p.N2;
    ep.rvalid = p.L - l0;
    ep.cvalid = p.N2 - n0;
    gemm_tile(al, bl, mk_ceil_div_dev(p.K, TK), ep, lds);
}

template <int VEC>
__global__ __launch_bounds__(kThreads) void legendre_inv_kernel(LegParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const TileId t = decode_block(p.Mloc, p.tiles_m, p.tiles_n);
    if (!t.valid) return;
    const int m = t.batch, mg = p.m_off + m;
    const int k0 = t.tm * TM;
    if (k0 >= p.K) return;
    const int n0 = t.tn * TN;
    const int nl = p.L - mg;  // contraction length (may be <= 0)
    DirectLoader<TM, 4, LDA_D> al;
    al.base = p.tab + ((long long)mg * p.L + (nl > 0 ? mg : 0)) * p.KP + k0;
    al.ldk = p.KP;
    al.kvalid = nl;
    al.cvalid = p.KP - k0;
    DirectLoader<TN, VEC, LDB_D> bl;
    bl.base = p.src + ((long long)(nl > 0 ? mg : 0) * p.Mloc + m) * p.N2 + n0;
    bl.ldk = (long long)p.Mloc * p.N2;
    bl.kvalid = nl;
    bl.cvalid = p.N2 - n0;
    Epilogue ep;
    ep.base = p.dst + ((long long)m * p.K + k0) * p.N2 + n0;
    ep.ldc = p.N2;
    ep.rvalid = p.K - k0;
    ep.cvalid = p.N2 - n0;
    gemm_tile(al, bl, nl > 0 ? mk_ceil_div_dev(nl, TK) : 0, ep, lds);
}

// ---------------------------------------------------------------------------
// dhconv forward / dgrad / wgrad
// ---------------------------------------------------------------------------
struct DhParams {
    const float* a;   // x (fwd, wgrad) or gy (dgrad)
    const float* b;   // w (fwd, dgrad) or gy (wgrad)
    float* dst;
    int Lloc, Mloc, B, I, O, l_off, m_off, tiles_m, tiles_n;
};

__device__ __forceinline__ int dh_rows(const DhParams& p, int l) {
    int nm = p.l_off + l - p.m_off + 1;  // local modes with global m <= global l
    nm = nm < 0 ? 0 : (nm > p.Mloc ? p.Mloc : nm);
    return nm * p.B;
}

template <int VEC>
__global__ __launch_bounds__(kThreads) void dhconv_fwd_kernel(DhParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const TileId t = decode_block(p.Lloc, p.tiles_m, p.tiles_n);
    if (!t.valid) return;
    const int l = p.Lloc - 1 - t.batch;  // heaviest degrees first
    const int R = dh_rows(p, l);
    const int r0 = t.tm * TM;
    if (r0 >= R) return;
    const int n0 = t.tn * TN;
    const long long rowbase = (long long)l * p.Mloc * p.B + r0;
    TransLoader<TM, VEC> al;
    al.base = p.a + rowbase * 2 * p.I;
    al.ldr = 2 * p.I;
    al.rvalid = R - r0;
    al.kvalid = 2 * p.I;
    ComplexRowLoader<TN, VEC, 0> bl;
    bl.base = p.b + (long long)l * p.I * 2 * p.O + n0;
    bl.ldq = 2 * p.O;
    bl.qvalid = p.I;
    bl.cvalid = 2 * p.O - n0;
    Epilogue ep;
    ep.base = p.dst + rowbase * 2 * p.O + n0;
    ep.ldc = 2 * p.O;
    ep.rvalid = R - r0;
    ep.cvalid = 2 * p.O - n0;
    gemm_tile(al, bl, mk_ceil_div_dev(2 * p.I, TK), ep, lds);
}

template <int VEC, int CPV>
__global__ __launch_bounds__(kThreads) void dhconv_dgrad_kernel(DhParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const TileId t = decode_block(p.Lloc, p.tiles_m, p.tiles_n);
    if (!t.valid) return;
    const int l = p.Lloc - 1 - t.batch;
    const int R = dh_rows(p, l);
    const int r0 = t.tm * TM;
    if (r0 >= R) return;
    const int n0 = t.tn * TN, i0 = n0 / 2;
    const long long rowbase = (long long)l * p.Mloc * p.B + r0;
    TransLoader<TM, VEC> al;
    al.base = p.a + rowbase * 2 * p.O;
    al.ldr = 2 * p.O;
    al.rvalid = R - r0;
    al.kvalid = 2 * p.O;
    ConjTransLoader<CPV> bl;
    bl.base = p.b + ((long long)l * p.I + i0) * 2 * p.O;
    bl.ldi = 2 * p.O;
    bl.ivalid = p.I - i0;
    bl.ovalid = p.O;
    Epilogue ep;
    ep.base = p.dst + rowbase * 2 * p.I + n0;
    ep.ldc = 2 * p.I;
    ep.rvalid = R - r0;
    ep.cvalid = 2 * p.I - n0;
    gemm_tile(al, bl, mk_ceil_div_dev(2 * p.O, TK), ep, lds);
}

template <int VEC, int CPV>
__global__ __launch_bounds__(kThreads) void dhconv_wgrad_kernel(DhParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const TileId t = decode_block(p.Lloc, p.tiles_m, p.tiles_n);
    if (!t.valid) return;
    const int l = p.Lloc - 1 - t.batch;
    const int R = dh_rows(p, l);
    const int i0 = t.tm * TM;
    if (i0 >= p.I) return;
    const int n0 = t.tn * TN;
    const long long rowbase = (long long)l * p.Mloc * p.B;
    DeinterleaveLoader<CPV> al;
    al.base = p.a + rowbase * 2 * p.I + 2 * i0;
    al.ldr = 2 * p.I;
    al.rvalid = R;
    al.ivalid = p.I - i0;
    ComplexRowLoader<TN, VEC, 1> bl;
    bl.base = p.b + rowbase * 2 * p.O + n0;
    bl.ldq = 2 * p.O;
    bl.qvalid = R;
    bl.cvalid = 2 * p.O - n0;
    Epilogue ep;
    ep.base = p.dst + ((long long)l * p.I + i0) * 2 * p.O + n0;
    ep.ldc = 2 * p.O;
    ep.rvalid = p.I - i0;
    ep.cvalid = 2 * p.O - n0;
    gemm_tile(al, bl, mk_ceil_div_dev(2 * R, TK), ep, lds);
}

constexpr size_t kLdsBytes = sizeof(float) * 2 * (A_WORDS + B_WORDS);

}  // namespace

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
static void apply_exp() {
    static int last = -1;
    const char* e = getenv("MK_GEMM_EXP");
    const int v = e ? atoi(e) : 0;
    if (v != last) {
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_exp), &v, sizeof(int));
        last = v;
    }
}

static int legendre_launch(bool fwd, const float* src, const float* tab, float* dst, int bc, int nlat, int lmax,
                           int mmax_loc, int m_off, int mmax_glob, hipStream_t st) {
    apply_exp();
    LegParams p;
    p.src = src;
    p.tab = tab;
    p.dst = dst;
    p.K = nlat;
    p.KP = mk_legendre_kpad(nlat);
    p.L = lmax;
    p.Mloc = mmax_loc;
    p.m_off = m_off;
    p.N2 = 2 * bc;
    p.tiles_m = fwd ? mk::ceil_div(lmax, TM) : mk::ceil_div(nlat, TM);
    p.tiles_n = mk::ceil_div(p.N2, TN);
    (void)mmax_glob;
    const long long nblk = (long long)mk::ceil_div(mmax_loc, 8) * 8 * p.tiles_m * p.tiles_n;
    if (nblk >= 2147483647LL) return -1;
    const dim3 grid(grid_blocks(mmax_loc, p.tiles_m, p.tiles_n));
    const bool v4 = (p.N2 % 4) == 0;
    if (fwd) {
        if (v4)
            hipLaunchKernelGGL(legendre_fwd_kernel<4>, grid, dim3(kThreads), kLdsBytes, st, p);
        else
            hipLaunchKernelGGL(legendre_fwd_kernel<2>, grid, dim3(kThreads), kLdsBytes, st, p);
    } else {
        if (v4)
            hipLaunchKernelGGL(legendre_inv_kernel<4>, grid, dim3(kThreads), kLdsBytes, st, p);
        else
            hipLaunchKernelGGL(legendre_inv_kernel<2>, grid, dim3(kThreads), kLdsBytes, st, p);
    }
    return 0;
}

extern "C" int mk_legendre_fwd(const float* xf, const float* tab, float* c, int bc, int nlat, int lmax,
                               int mmax_loc, int m_off, int mmax_glob, void* stream) {
    MK_REQUIRE(xf && tab && c, "null pointer");
    MK_REQUIRE(bc > 0 && nlat > 0 && lmax > 0 && mmax_loc > 0, "bad sizes");
    MK_REQUIRE(m_off >= 0 && m_off + mmax_loc <= mmax_glob, "mode shard out of range");
    MK_REQUIRE(legendre_launch(true, xf, tab, c, bc, nlat, lmax, mmax_loc, m_off, mmax_glob, (hipStream_t)stream) == 0,
               "grid too large");
    MK_LAUNCH_CHECK();
    return 0;
}

extern "C" int mk_legendre_inv(const float* c, const float* tab, float* xf, int bc, int nlat, int lmax,
                               int mmax_loc, int m_off, int mmax_glob, void* stream) {
    MK_REQUIRE(xf && tab && c, "null pointer");
    MK_REQUIRE(bc > 0 && nlat > 0 && lmax > 0 && mmax_loc > 0, "bad sizes");
    MK_REQUIRE(m_off >= 0 && m_off + mmax_loc <= mmax_glob, "mode shard out of range");
    MK_REQUIRE(legendre_launch(false, c, tab, xf, bc, nlat, lmax, mmax_loc, m_off, mmax_glob, (hipStream_t)stream) == 0,
               "grid too large");
    MK_LAUNCH_CHECK();
    return 0;
}

static int dh_check(const void* a, const void* b, const void* c, int lloc, int mloc, int batch, int cin, int cout,
                    int l_off, int m_off) {
    apply_exp();
    MK_REQUIRE(a && b && c, "null pointer");
    MK_REQUIRE(lloc > 0 && mloc > 0 && batch > 0 && cin > 0 && cout > 0, "bad sizes");
    MK_REQUIRE(l_off >= 0 && m_off >= 0, "negative shard offset");
    return 0;
}

extern "C" int mk_dhconv_fwd(const float* x, const float* w, float* y, int lloc, int mloc, int batch, int cin,
                             int cout, int l_off, int m_off, void* stream) {
    if (int e = dh_check(x, w, y, lloc, mloc, batch, cin, cout, l_off, m_off)) return e;
    DhParams p{x, w, y, lloc, mloc, batch, cin, cout, l_off, m_off, mk::ceil_div(mloc * batch, TM),
               mk::ceil_div(2 * cout, TN)};
    const dim3 grid(grid_blocks(lloc, p.tiles_m, p.tiles_n));
    if (cin % 2 == 0 && cout % 2 == 0)
        hipLaunchKernelGGL(dhconv_fwd_kernel<4>, grid, dim3(kThreads), kLdsBytes, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL(dhconv_fwd_kernel<2>, grid, dim3(kThreads), kLdsBytes, (hipStream_t)stream, p);
    MK_LAUNCH_CHECK();
    return 0;
}

extern "C" int mk_dhconv_dgrad(const float* gy, const float* w, float* gx, int lloc, int mloc, int batch, int cin,
                               int cout, int l_off, int m_off, void* stream) {
    if (int e = dh_check(gy, w, gx, lloc, mloc, batch, cin, cout, l_off, m_off)) return e;
    DhParams p{gy, w, gx, lloc, mloc, batch, cin, cout, l_off, m_off, mk::ceil_div(mloc * batch, TM),
               mk::ceil_div(2 * cin, TN)};
    const dim3 grid(grid_blocks(lloc, p.tiles_m, p.tiles_n));
    if (cin % 2 == 0 && cout % 2 == 0)
        hipLaunchKernelGGL((dhconv_dgrad_kernel<4, 2>), grid, dim3(kThreads), kLdsBytes, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL((dhconv_dgrad_kernel<2, 1>), grid, dim3(kThreads), kLdsBytes, (hipStream_t)stream, p);
    MK_LAUNCH_CHECK();
    return 0;
}

extern "C" int mk_dhconv_wgrad(const float* x, const float* gy, float* gw, int lloc, int mloc, int batch, int cin,
                               int cout, int l_off, int m_off, void* stream) {
    if (int e = dh_check(x, gy, gw, lloc, mloc, batch, cin, cout, l_off, m_off)) return e;
    DhParams p{x, gy, gw, lloc, mloc, batch, cin, cout, l_off, m_off, mk::ceil_div(cin, TM),
               mk::ceil_div(2 * cout, TN)};
    const dim3 grid(grid_blocks(lloc, p.tiles_m, p.tiles_n));
    if (cin % 2 == 0 && cout % 2 == 0)
        hipLaunchKernelGGL((dhconv_wgrad_kernel<4, 2>), grid, dim3(kThreads), kLdsBytes, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL((dhconv_wgrad_kernel<2, 1>), grid, dim3(kThreads), kLdsBytes, (hipStream_t)stream, p);
    MK_LAUNCH_CHECK();
    return 0;
}
