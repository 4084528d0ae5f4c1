// Split real FFT for the two production longitudes, N = 480 * S (S = 1: 480, S = 3: 1440).
// Included inside the anonymous namespace of fft.hip (uses Dft<>, cmul, ...).
//
// A real row of N = 480*S points is S interleaved sub-sequences x[S*n + s] of 480 reals.  Each is
// transformed as a 240-point complex FFT (480 reals packed in pairs) with the radix plan 16 x 15,
// and because only modes m <= 240 are kept the radix-S decimation-in-time combine
//     X[m] = sum_s W_N^{s m} F_s[m],   m <= 240
// needs exactly the 241 non-redundant modes of every F_s: nothing is computed that is thrown away.
//
// Workgroup = 384 threads = 24 sub-rows x 16 lanes (S = 3: 8 rows; S = 1: 24 rows).  A sub-row is
// owned by 16 lanes of ONE wave, pass A (radix 16) and pass B (radix 15, in place) run back to back
// with no workgroup barrier: LDS operations of a wave execute in program order.  Only the
// global <-> LDS staging and the mode split / combine cross waves (2 barriers per tile).
//
// LDS image: sub-row sr at float2 offset sr*272 + 4*(sr/S); pass A output index i is stored at
// i + (i >> 4) (16-element blocks padded by one) -> conflict-free radix-16 scatter and radix-15
// gathers; the 4*(row) rotation makes the 8-rows-per-mode accesses of the split step conflict-free.

constexpr int SH = 240;        // complex points per sub-FFT
constexpr int SP = 272;        // sub-row stride (float2), = 16 mod 32
constexpr int SNSUB = 24;      // sub-rows per workgroup
constexpr int STHREADS = 384;  // 16 lanes per sub-row
constexpr int SLDS_F2 = SNSUB * SP + 96;

// Schedule index -> tile.  Tiles t and t + 1 (adjacent channel groups of one latitude) share the 128-byte
// lines of xf[m][k][:] (a tile covers 64 or 192 bytes per mode): run them on the same XCD (schedule index
// mod 8) back to back, so each line is fetched from / merged for HBM once instead of once per XCD.
__device__ __forceinline__ int pair_tile(int i) {
    const int xcd = i & 7, seq = i >> 3;
    return ((((seq >> 1) << 3) + xcd) << 1) + (seq & 1);
}

__device__ __forceinline__ int sub_base(int sr, int S) { return sr * SP + 4 * (sr / S); }
__device__ __forceinline__ int phi(int i) { return i + (i >> 4); }
// an opaque copy of a value: stops the compiler from hoisting per-tile index arithmetic out of the
// persistent tile loop (which costs ~100 VGPRs of precomputed offsets and an occupancy level)
__device__ __forceinline__ int opaque(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

// value of an output element as it is stored (fp32 rows: itself; bf16 rows: rounded): what the statistics below must see
template <typename T> __device__ __forceinline__ float as_stored(float v);
template <> __device__ __forceinline__ float as_stored<float>(float v) { return v; }
template <> __device__ __forceinline__ float as_stored<__hip_bfloat16>(float v) { return __bfloat162float(__float2bfloat16(v)); }

// radix-16 then radix-15 on the sub-row owned by this lane group; tw15[r] = exp(-2 pi i k r / 240).
// SUMS (inverse transform only): the finished sub-row IS 480 consecutive-stride output reals (x[2n] = Re z, x[2n+1] = -Im z);
// their sum and sum of squares, taken on the values as stored, go to red[2 sr], red[2 sr + 1] -- the row statistics of the
// instance norm that reads the field next, for free while the row is in registers.
template <bool SUMS = false, typename TOut = float>
__device__ __forceinline__ void split_passes(float2* lds, int tid, int S, const float2 (&tw15)[15], float* red = nullptr) {
    const int sr = tid >> 4, j = tid & 15;
    float2* base = lds + sub_base(sr, S);
    if (j < 15) {
        float2 v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = base[j + 15 * r];
        Dft<16>::run(v);
#pragma unroll
        for (int r = 0; r < 16; ++r) base[17 * j + r] = v[r];
    }
    // no barrier: the whole sub-row lives in this wave, DS ops of a wave are ordered
    float2 u[15];
#pragma unroll
    for (int r = 0; r < 15; ++r) u[r] = base[j + 17 * r];
#pragma unroll
    for (int r = 1; r < 15; ++r) u[r] = cmul(u[r], tw15[r]);
    Dft<15>::run(u);
#pragma unroll
    for (int r = 0; r < 15; ++r) base[j + 17 * r] = u[r];
    if constexpr (SUMS) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int r = 0; r < 15; ++r) {
            const float a = as_stored<TOut>(u[r].x), b = as_stored<TOut>(u[r].y);
            s1 += a - b;
            s2 = fmaf(a, a, fmaf(b, b, s2));
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {      // the 16 lanes of the sub-row
            s1 += __shfl_xor(s1, o, 16);
            s2 += __shfl_xor(s2, o, 16);
        }
        if (j == 0 && red != nullptr) {
            red[2 * sr] = s1;
            red[2 * sr + 1] = s2;
        }
    }
}

template <typename T> struct InVec;
template <> struct InVec<float> {
    static constexpr int E = 4;
    static __device__ __forceinline__ void unpack(const uint4& v, float (&o)[4]) {
        o[0] = __uint_as_float(v.x); o[1] = __uint_as_float(v.y); o[2] = __uint_as_float(v.z); o[3] = __uint_as_float(v.w);
    }
    static __device__ __forceinline__ void load(const float* p, float (&o)[4]) {
        const float4 v = *reinterpret_cast<const float4*>(p);
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    }
};
template <> struct InVec<__hip_bfloat16> {
    static constexpr int E = 8;
    static __device__ __forceinline__ void unpack(const uint4& v, float (&o)[8]) {
        const unsigned int w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            o[2 * i] = __uint_as_float(w[i] << 16);
            o[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    }
    static __device__ __forceinline__ void load(const __hip_bfloat16* p, float (&o)[8]) {
        const uint4 v = *reinterpret_cast<const uint4*>(p);
        const unsigned int w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            o[2 * i] = __uint_as_float(w[i] << 16);
            o[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    }
};

template <int S, typename TIn>
__global__ __launch_bounds__(STHREADS) void rfft_split_kernel(const TIn* __restrict__ x, float2* __restrict__ xf,
                                                              const float2* __restrict__ tw, int BC, int K, int M,
                                                              float scale0, float scale_m, float scale_h, int exp, XfLayout xl) {
    constexpr int N = 480 * S, G = SNSUB / S, HH = N / 2, E = InVec<TIn>::E;
    // staging unit: S consecutive vectors of E reals = E reals (E/2 complex) of every sub-sequence
    constexpr int GPR = N / (S * E), NGRP = G * GPR, IT = (NGRP + STHREADS - 1) / STHREADS;
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    const int tid = threadIdx.x;
    const int ntile = (BC + G - 1) / G;
    const int tile = pair_tile(blockIdx.x);
    if (tile >= ntile * K) return;
    const int k = tile / ntile;
    const int bc0 = (tile - k * ntile) * G;
    const XfChan xc = xf_chan(xl, bc0, BC);

    // The row loads only ISSUE here -- raw 16-byte buffer loads (a lane with nothing to fetch gets an offset past the
    // descriptor's range and reads zeros), unpacked to floats where they are staged into LDS.  The earlier form converted
    // each vector right after its conditional load: a use of the loaded value, so the compiler put `s_waitcnt vmcnt(0)`
    // behind every one of the 4-6 row loads of a thread and the HBM latency was paid 4-6 times per tile, in series.
    typedef unsigned int rf_u4 __attribute__((ext_vector_type(4)));
    uint4 raw[IT][S];
    {
        const TIn* tile_base = x + ((size_t)bc0 * K + k) * N;         // offsets below stay < 24 rows x K x N elements
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<TIn*>(tile_base), 0, 0x7FFFFFFF, 0x00020000);
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int v = tid + it * STHREADS;
            const int g = v / GPR, p = v - g * GPR;
            const bool ok = (v < NGRP) && (bc0 + g < BC) && !(exp & 1);
#pragma unroll
            for (int c = 0; c < S; ++c) {
                const unsigned off = ok ? (unsigned)((((size_t)g * K) * N + (size_t)(p * S + c) * E) * sizeof(TIn)) : 0x80000000u;
                raw[it][c] = __builtin_bit_cast(uint4, (rf_u4)__builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
            }
        }
    }
    float2 tw15[15];
#pragma unroll
    for (int r = 0; r < 15; ++r) tw15[r] = tw[((tid & 15) * r) * S];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int v = tid + it * STHREADS;
        if (v < NGRP) {
            const int g = v / GPR, p = v - g * GPR;
            float vals[S][E];
#pragma unroll
            for (int c = 0; c < S; ++c) InVec<TIn>::unpack(raw[it][c], vals[c]);
            // flat element f = c*E + e of the group is real index S*E*p + f = S*n + s: s = f % S, n = E*p + f / S
#pragma unroll
            for (int sq = 0; sq < S; ++sq) {
                float4* dst = reinterpret_cast<float4*>(lds + sub_base(g * S + sq, S) + (E / 2) * p);
#pragma unroll
                for (int h = 0; h < E / 4; ++h) {
                    float o[4];
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        const int f = (4 * h + a) * S + sq;   // n - E*p = 4h + a
                        o[a] = vals[f / E][f % E];
                    }
                    dst[h] = make_float4(o[0], o[1], o[2], o[3]);
                }
            }
        }
    }
    __syncthreads();
    if (!(exp & 2)) split_passes(lds, tid, S, tw15);
    // exp(-2 pi i s' (tid / G) / N), s' = S, 1, 2: the split step's twiddles of this thread's first item -- issued here, where
    // the pass twiddles are dead (register peak) and the barrier below covers part of the latency
    float2 wfirst[S];
    wfirst[0] = tw[HH + S * (tid / G)];
#pragma unroll
    for (int s = 1; s < S; ++s) wfirst[s] = tw[HH + s * (tid / G)];
    __syncthreads();
    if (exp & 8) return;

    // mode split + radix-S combine, modes m and 240 - m together (they share Z[m], Z[240 - m]):
    //   F[m]     = e - i p,  F[240-m] = conj(e) - i conj(p),  e = (a + conj b)/2, p = w_m (a - conj b)/2
    //   X[m]     = sum_s W^{s m} F_s[m],   X[240-m] = sum_s conj(W^{s m}) e^{-i pi s/S} F_s[240-m]
    const float2* tw2 = tw + HH;  // exp(-2 pi i m / N)
    constexpr int NPAIR = SH / 2 + 1;  // mp = 0 .. 120
    // Thread (mp0, g) = (tid / G, tid % G) handles the mode pairs mp0, mp0 + STEP, ... of channel g (STEP = 384 / G is a
    // whole number: 16 or 48).  Its twiddles exp(-2 pi i s mp / N) come from ONE exact table read per s (for mp0, issued
    // with the row loads at the top of the kernel) and one rotation by an exact, thread-uniform table value per item --
    // the per-item table reads this replaces were dependent global loads in the middle of the loop (latency exposed 8 times
    // per tile), and the item -> (mp, g) map cost an integer division each.
    constexpr int STEP = STHREADS / G, PIT = (NPAIR + STEP - 1) / STEP;
    static_assert(STHREADS % G == 0, "items of a thread must keep their channel");
    const int mp0 = tid / G, g = tid - mp0 * G;
    if (bc0 + g < BC) {
        float2* dst0 = xf + xc.pbase + ((size_t)mp0 * xl.sm + (size_t)k * xl.sk) * xc.BCx + (xc.bcx0 + g);              // mode mp0
        float2* dst1 = xf + xc.pbase + ((size_t)(SH - mp0) * xl.sm + (size_t)k * xl.sk) * xc.BCx + (xc.bcx0 + g);       // partner
        const ptrdiff_t dstep = (ptrdiff_t)STEP * xl.sm * xc.BCx;
#pragma unroll
        for (int it = 0; it < PIT; ++it) {
            const int mp = mp0 + STEP * it;
            if (mp < NPAIR) {
                const int m1 = SH - mp;                 // partner mode (240 for mp = 0)
                const int p0 = phi(mp), p1 = phi(mp == 0 ? 0 : m1);
                const float2 wsub = it ? cmul(wfirst[0], tw2[S * STEP * it]) : wfirst[0];      // exp(-2 pi i mp / 480)
                float2 acc0 = make_float2(0.f, 0.f), acc1 = make_float2(0.f, 0.f);
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    const float2* zs = lds + sub_base(g * S + s, S);
                    const float2 a = zs[p0];
                    const float2 b = zs[p1];
                    const float2 e = make_float2(0.5f * (a.x + b.x), 0.5f * (a.y - b.y));
                    const float2 d = make_float2(0.5f * (a.x - b.x), 0.5f * (a.y + b.y));
                    const float2 pp = cmul(wsub, d);
                    float2 f0 = make_float2(e.x + pp.y, e.y - pp.x);
                    float2 f1 = make_float2(e.x - pp.y, -e.y - pp.x);
                    if (s > 0) {
                        const float2 w = it ? cmul(wfirst[s], tw2[s * STEP * it]) : wfirst[s];       // exp(-2 pi i s mp / N)
                        constexpr float kc[3] = {1.f, 0.5f, -0.5f}, ks[3] = {0.f, -0.86602540378443864676f, -0.86602540378443864676f};
                        static_assert(S == 1 || S == 3, "combine constants are for S = 3");
                        f0 = cmul(f0, w);
                        f1 = cmul(f1, cmul(make_float2(w.x, -w.y), make_float2(kc[s], ks[s])));
                    }
                    acc0 = cadd(acc0, f0);
                    acc1 = cadd(acc1, f1);
                }
                if (exp & 4) {   // ablation: keep the arithmetic alive, store (almost) nothing
                    if (acc0.x + acc1.x != 12345.678f) continue;
                }
                if (mp < M) {
                    const float sc = (mp == 0) ? scale0 : scale_m;
                    dst0[(ptrdiff_t)it * dstep] = make_float2(sc * acc0.x, sc * acc0.y);
                }
                if (m1 != mp && m1 < M) {
                    const float sc = (m1 == HH) ? scale_h : scale_m;
                    dst1[-(ptrdiff_t)it * dstep] = make_float2(sc * acc1.x, sc * acc1.y);
                }
            }
        }
    }
}

template <typename T> struct OutVec;
template <> struct OutVec<float> {
    static constexpr int E = 4;
    static __device__ __forceinline__ void store(float* p, const float (&o)[4]) {
        *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
    }
};
template <> struct OutVec<__hip_bfloat16> {
    static constexpr int E = 8;
    static __device__ __forceinline__ void store(__hip_bfloat16* p, const float (&o)[8]) {
        __hip_bfloat16 h[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) h[i] = __float2bfloat16(o[i]);
        *reinterpret_cast<uint4*>(p) = *reinterpret_cast<const uint4*>(h);      // (nontemporal here: measured no change)
    }
};

// WIDE: mode offsets of one tile may exceed 2^31 bytes (mode-major Fourier rows of a very large batch): 64-bit pointer
// arithmetic per gather instead of 32-bit buffer offsets.
// PERSIST = 1: workgroups walk the tiles with the next tile's mode gather in flight under the current tile's passes (32 prefetch
// registers: 167 VGPRs, two workgroups per CU).  PERSIST = 0: one tile per workgroup, nothing prefetched, the pass twiddles loaded
// after the merge step -- registers for five waves per SIMD, i.e. three workgroups per CU covering each other like the forward kernel.
template <int S, typename TOut, bool WIDE, int PERSIST, bool SUMS = false>
__global__ __launch_bounds__(STHREADS, PERSIST ? 3 : 5) void irfft_split_kernel(const float2* __restrict__ xf, TOut* __restrict__ x,
                                                               const float2* __restrict__ tw, int BC, int K, int M,
                                                               float scale0, float scale_m, float scale_h, XfLayout xl,
                                                               double* __restrict__ rowsums) {
    constexpr int N = 480 * S, G = SNSUB / S, HH = N / 2;
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    float* red = reinterpret_cast<float*>(lds + SLDS_F2);      // [24 sub-rows][sum, sum of squares] (rowsums only)
    const int tid = threadIdx.x;
    const int ntile = (BC + G - 1) / G;
    const int total = ntile * K;
    const float2* tw2 = tw + HH;
    // persistent: this workgroup walks the schedule indices blockIdx.x, + gridDim.x, ... (tiles via pair_tile)
    const int total16 = (total + 15) & ~15;
    auto next_valid = [&](int i) {
        while (i < total16 && pair_tile(i) >= total) i += (int)gridDim.x;
        return i;
    };
    int sched = PERSIST ? next_valid(blockIdx.x) : (int)blockIdx.x;
    if (sched >= total16) return;
    int tile = pair_tile(sched);
    if (!PERSIST && tile >= total) return;

    float2 tw15[15];
    auto load_tw15 = [&]() {
#pragma unroll
        for (int r = 0; r < 15; ++r) tw15[r] = tw[((tid & 15) * r) * S];
    };
    if constexpr (PERSIST != 0) load_tw15();

    float2 wfirst[S];        // exp(-2 pi i s' (tid / G) / N), s' = S, 1, 2: merge-step twiddles of this thread's first item
    wfirst[0] = tw2[S * (tid / G)];
#pragma unroll
    for (int s = 1; s < S; ++s) wfirst[s] = tw2[s * (tid / G)];

    // merge step, modes j and 240 - j together (both need X[j] and X[240 - j]):
    //   Y_s[m] = X[m] W^{-s m};  e = Ya + conj Yb, t = w_j (Ya - conj Yb), w_j = exp(+2 pi i j / 480)
    //   Z[j] = e + i t,  Z[240-j] = conj(e) + i conj(t); stored conjugated (inverse = conj . forward . conj)
    constexpr int NPAIR = SH / 2 + 1;
    constexpr int PIT = (G * NPAIR + STHREADS - 1) / STHREADS;
    float2 xa[PIT], xb[PIT];
    // Thread (jp0, g) = (tid / G, tid % G) owns the mode pairs jp0, jp0 + STEP, ... of channel g (STEP = 384 / G: 16 or 48),
    // in the gathers and in the merge step alike.
    constexpr int STEP = STHREADS / G;
    static_assert(STHREADS % G == 0 && PIT == (NPAIR + STEP - 1) / STEP, "items of a thread keep their channel");
    auto gather = [&](int t, int tl) {      // the 8-byte mode gathers of tile t (in flight while the previous tile computes)
        const int k = t / ntile, bc0 = (t - k * ntile) * G;
        const XfChan xc = xf_chan(xl, bc0, BC);
        const int jp0 = tl / G, g = tl - jp0 * G;
        const bool chan_ok = bc0 + g < BC;
        // Every load is issued unconditionally and nothing here consumes a loaded value: a conditional load into a pre-zeroed
        // register became load + copy, and the copy made the compiler wait for the whole gather right here instead of after
        // the FFT passes of the current tile.  The merge step masks.
        if constexpr (!WIDE) {
            // Raw buffer loads on a tile-uniform descriptor with 32-bit lane offsets that advance by one add per item; a lane
            // with nothing to fetch gets an offset past the range and reads zeros.  (The pointer form below cost a 64-bit
            // multiply chain -- four quarter-rate integer multiplies -- and an exec-mask branch per 8-byte load: 16 loads per
            // thread and tile, a fifth of the kernel's VALU cycles.)
            typedef unsigned int gf_u2 __attribute__((ext_vector_type(2)));
            const float2* tbase = xf + xc.pbase + (size_t)k * xl.sk * xc.BCx + xc.bcx0;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2*>(tbase), 0, 0x7FFFFFFF, 0x00020000);
            const unsigned mstride = (unsigned)xl.sm * (unsigned)xc.BCx * 8u;        // bytes from one mode to the next
            unsigned offa = (unsigned)g * 8u + (unsigned)jp0 * mstride;
            unsigned offb = (unsigned)g * 8u + (unsigned)(SH - jp0) * mstride;
            const unsigned dstepb = (unsigned)STEP * mstride;
#pragma unroll
            for (int it = 0; it < PIT; ++it) {
                const int jp = jp0 + STEP * it, j1 = SH - jp;
                const bool oa = chan_ok && jp < NPAIR && jp < M, ob = chan_ok && jp < NPAIR && j1 < M;
                xa[it] = __builtin_bit_cast(float2, (gf_u2)__builtin_amdgcn_raw_buffer_load_b64(rs, oa ? offa : 0x80000000u, 0, 0));
                xb[it] = __builtin_bit_cast(float2, (gf_u2)__builtin_amdgcn_raw_buffer_load_b64(rs, ob ? offb : 0x80000000u, 0, 0));
                offa += dstepb;
                offb -= dstepb;
            }
        } else {
            const float2* base = xf + xc.pbase + (size_t)k * xl.sk * xc.BCx + (xc.bcx0 + g);
#pragma unroll
            for (int it = 0; it < PIT; ++it) {
                const int jp = jp0 + STEP * it, j1 = SH - jp;
                const bool oa = chan_ok && jp < NPAIR && jp < M, ob = chan_ok && jp < NPAIR && j1 < M;
                xa[it] = *(oa ? base + (size_t)jp * xl.sm * xc.BCx : xf);
                xb[it] = *(ob ? base + (size_t)j1 * xl.sm * xc.BCx : xf);
            }
        }
    };
    gather(tile, tid);
  for (;;) {
    const int tl = opaque(tid);
    const int k = tile / ntile, bc0 = (tile - k * ntile) * G;
    const int jp0 = tl / G, g = tl - jp0 * G;
    const bool chan_ok = bc0 + g < BC;
#pragma unroll
    for (int it = 0; it < PIT; ++it) {
        const int jp = jp0 + STEP * it;
        if (jp >= NPAIR) continue;
        const int j1 = SH - jp;
        const float sa = (chan_ok && jp < M) ? ((jp == 0) ? scale0 : scale_m) : 0.f;      // the gather's masks (see there)
        const float sb = (chan_ok && j1 < M) ? ((j1 == HH) ? scale_h : scale_m) : 0.f;
        const float2 a0 = make_float2(sa * xa[it].x, sa * xa[it].y);
        const float2 b0 = make_float2(sb * xb[it].x, sb * xb[it].y);
        // twiddles: one exact table value per thread (its first item, loaded once per launch) rotated by an exact,
        // thread-uniform table value per item -- no dependent table reads inside the loop
        float2 wsub = it ? cmul(wfirst[0], tw2[S * STEP * it]) : wfirst[0];
        wsub.y = -wsub.y;  // exp(+2 pi i jp / 480)
#pragma unroll
        for (int s = 0; s < S; ++s) {
            float2 ya = a0, yb = b0;
            if (s > 0) {
                constexpr float kc[3] = {1.f, 0.5f, -0.5f}, ks[3] = {0.f, 0.86602540378443864676f, 0.86602540378443864676f};
                const float2 w = it ? cmul(wfirst[s], tw2[s * STEP * it]) : wfirst[s];      // W^{s jp}
                ya = cmul(ya, make_float2(w.x, -w.y));          // X[j]  W^{-s j}
                yb = cmul(yb, cmul(w, make_float2(kc[s], ks[s])));  // X[j1] W^{-s (240 - j)} = X[j1] e^{+i pi s/3} W^{s j}
            }
            if (jp == 0) {
                ya.y = 0.f;                 // imaginary part of the zero mode is ignored
                yb.y = 0.f;                 // Nyquist of the 480-point sub-transform is real ...
                if (S > 1) yb.x *= 2.f;     // ... and an interior mode of the long one: both conjugate halves
            }
            const float2 e = make_float2(ya.x + yb.x, ya.y - yb.y);
            const float2 d = make_float2(ya.x - yb.x, ya.y + yb.y);
            const float2 t = cmul(wsub, d);
            float2* zs = lds + sub_base(g * S + s, S);
            zs[jp] = make_float2(e.x - t.y, -(e.y + t.x));
            if (jp != 0 && j1 != jp) zs[j1] = make_float2(e.x + t.y, e.y - t.x);
        }
    }
    if constexpr (PERSIST == 0) load_tw15();      // issued before the barrier: the latency sits under the wait for the other waves
    __syncthreads();
    const int next = PERSIST ? next_valid(sched + (int)gridDim.x) : total16;
    if (next < total16) gather(pair_tile(next), opaque(tid));
    if constexpr (SUMS) split_passes<true, TOut>(lds, tl, S, tw15, red);
    else split_passes(lds, tl, S, tw15);
    __syncthreads();
    if (SUMS && tl < 2 * G) {      // one fp64 atomic per (channel, statistic) and tile: the row's share of the field sums
        const int gch = tl >> 1, t = tl & 1;
        if (bc0 + gch < BC) {
            float v = 0.f;
#pragma unroll
            for (int sq = 0; sq < S; ++sq) v += red[2 * (gch * S + sq) + t];
            atomicAdd(&rowsums[2 * (size_t)(bc0 + gch) + t], (double)v);
        }
    }

    // copy-out: groups of EO*S reals = EO reals (EO/2 complex, contiguous in the padded image) of every
    // sub-sequence, de-conjugated; EO = 4 (fp32 rows) or 8 (bf16 rows) -> 16-byte stores
    constexpr int EO = OutVec<TOut>::E;
    constexpr int GPR = N / (EO * S), NGRP = G * GPR, OIT = (NGRP + STHREADS - 1) / STHREADS;
#pragma unroll
    for (int it = 0; it < OIT; ++it) {
        const int v = tl + it * STHREADS;
        const int g = v / GPR, p = v - g * GPR;
        if (v < NGRP && bc0 + g < BC) {
            float r[S][EO];
#pragma unroll
            for (int sq = 0; sq < S; ++sq) {
                const float2* zs = lds + sub_base(g * S + sq, S) + phi((EO / 2) * p);
#pragma unroll
                for (int h = 0; h < EO / 2; ++h) {
                    const float2 z = zs[h];
                    r[sq][2 * h] = z.x;
                    r[sq][2 * h + 1] = -z.y;
                }
            }
            TOut* dst = x + ((size_t)(bc0 + g) * K + k) * N + (size_t)p * EO * S;
#pragma unroll
            for (int c = 0; c < S; ++c) {
                float o[EO];
#pragma unroll
                for (int e = 0; e < EO; ++e) {
                    const int f = c * EO + e;      // real index S*EO*p + f = S*n + s
                    o[e] = r[f % S][f / S];
                }
                OutVec<TOut>::store(dst + c * EO, o);
            }
        }
    }
    if (next >= total16) break;
    sched = next;
    tile = pair_tile(next);
    __syncthreads();   // the image is rewritten by the next tile's merge step
  }
}

// Inverse kernel: one workgroup per tile (three per CU covering each other, as in the forward kernel).  Two persistent
// workgroups per CU with the next tile's mode gather prefetched in registers measured 15-21 % slower (0.595 vs 0.505 ms at
// 721 x 1440, 0.094 vs 0.074 ms at 240 x 480) and were removed; PERSIST = 1 survives only as the loop of the WIDE variant.
static inline unsigned split_grid(long long tiles) { return (unsigned)((tiles + 15) / 16 * 16); }

// ablation switches for the forward kernel (MK_FFT_EXP): 1 no row loads, 2 no FFT passes, 4 no mode stores,
// 8 stop after the passes -- wrong results, timing experiments only
static inline int fft_exp() {
    static const int v = [] { const char* e = getenv("MK_FFT_EXP"); return e ? atoi(e) : 0; }();
    return v;
}

template <int S>
int launch_rfft_split(const void* x, int x_dtype, float* xf, const float* tw, int bc, int nlat, int mmax, float s0,
                      float sm, float sh, hipStream_t st) {
    constexpr int G = SNSUB / S;
    // one workgroup per tile, rounded up to whole XCD pair groups.  Persistence was measured twice: with a prefetch of
    // the next rows it costs 70 VGPRs; as a plain loop over tiles at 2-4 workgroups per CU it is 13-33 % slower than this
    // launch-per-tile form although wave launches alone account for 0.14 of the 0.59 ms (MK_FFT_EXP=15).
    const dim3 grid((unsigned)((mk::ceil_div(bc, G) * nlat + 15) / 16 * 16));
    const size_t lds = sizeof(float2) * SLDS_F2;
    if (x_dtype == 0)
        hipLaunchKernelGGL((rfft_split_kernel<S, float>), grid, dim3(STHREADS), lds, st, (const float*)x, (float2*)xf,
                           (const float2*)tw, bc, nlat, mmax, s0, sm, sh, fft_exp(), g_xl);
    else
        hipLaunchKernelGGL((rfft_split_kernel<S, __hip_bfloat16>), grid, dim3(STHREADS), lds, st,
                           (const __hip_bfloat16*)x, (float2*)xf, (const float2*)tw, bc, nlat, mmax, s0, sm, sh, fft_exp(), g_xl);
    return 0;
}

template <int S>
int launch_irfft_split(const float* xf, void* x, int x_dtype, const float* tw, int bc, int nlat, int mmax, float s0,
                       float sm, float sh, hipStream_t st) {
    constexpr int G = SNSUB / S;
    const dim3 grid(split_grid(mk::ceil_div(bc, G) * (long long)nlat));
    const size_t lds = sizeof(float2) * SLDS_F2 + 2 * SNSUB * sizeof(float);
    // 32-bit mode offsets inside a tile's buffer descriptor: (240 modes + one row of channels) * 8 bytes must stay below 2^31
    const long long bcx = g_xl.Cp ? (long long)g_xl.Bn * g_xl.Cp : (long long)bc;
    const bool wide = ((long long)SH * g_xl.sm + 1) * bcx * 8 >= (1LL << 31);
#define MK_IRFFT_LAUNCH(T, W, PS, SM)                                                                                  \
    hipLaunchKernelGGL((irfft_split_kernel<S, T, W, PS, SM>), grid, dim3(STHREADS), lds, st, (const float2*)xf, (T*)x,    \
                       (const float2*)tw, bc, nlat, mmax, s0, sm, sh, g_xl, g_rowsums)
#define MK_IRFFT_PICK(T)                                                                                               \
    if (g_rowsums) {                                                                                                   \
        if (wide) MK_IRFFT_LAUNCH(T, true, 1, true); else MK_IRFFT_LAUNCH(T, false, 0, true);                          \
    } else {                                                                                                           \
        if (wide) MK_IRFFT_LAUNCH(T, true, 1, false); else MK_IRFFT_LAUNCH(T, false, 0, false);                        \
    }
    if (x_dtype == 0) {
        MK_IRFFT_PICK(float)
    } else {
        MK_IRFFT_PICK(__hip_bfloat16)
    }
#undef MK_IRFFT_PICK
#undef MK_IRFFT_LAUNCH
    return 0;
}

static inline bool fft_legacy() {
    static const bool v = [] {
        const char* e = getenv("MK_FFT_LEGACY");
        return e && e[0] == '1';
    }();
    return v;
}
