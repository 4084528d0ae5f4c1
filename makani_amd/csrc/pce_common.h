// Device helpers shared by the pixel-column engine kernels (pce.hip: one 1x1 convolution per launch; pce_mlp.hip: the
// fused conv + GELU + conv pair): GELU on the VALU budget of an epilogue, LDS-DMA issue, LDS accesses as inline asm
// with hand-counted waits (the compiler drains vmcnt(0) in front of every LDS access it can see while an LDS-DMA is
// in flight), raw barriers.
#pragma once
#include <hip/hip_bf16.h>
#include <cstdint>

namespace pce {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// ---- GELU (exact erf form, makani uses nn.GELU()) on the VALU budget of an epilogue --------------------------
// Phi(x) through erfc(|x|/sqrt2) with the Abramowitz-Stegun 7.1.26 rational-exponential form (|error| < 1.5e-7
// absolute on erf): 2 transcendentals + ~12 FMAs instead of ~30 instructions of erff().  The results are rounded
// to bf16 (2^-9 relative) right after.
struct PhiPair {
    float Phi, phi;   // standard normal CDF and PDF at x
};
__device__ __forceinline__ PhiPair normal_cdf_pdf(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float q = fmaf(1.061405429f, t, -1.453152027f);
    q = fmaf(q, t, 1.421413741f);
    q = fmaf(q, t, -0.284496736f);
    q = fmaf(q, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);   // exp(-x^2/2)
    const float half_erfc = 0.5f * q * t * e;                               // 0.5 erfc(|x|/sqrt2) = Phi(-|x|)
    PhiPair r;
    r.Phi = x < 0.f ? half_erfc : 1.0f - half_erfc;
    r.phi = 0.3989422804014327f * e;
    return r;
}
__device__ __forceinline__ float gelu_f(float x) { return x * normal_cdf_pdf(x).Phi; }
__device__ __forceinline__ float gelu_grad_f(float x) {
    const PhiPair c = normal_cdf_pdf(x);
    return fmaf(x, c.phi, c.Phi);
}

__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) { return __uint_as_float((uint32_t)b << 16); }
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float v) {
    const __hip_bfloat16 h = __float2bfloat16(v);
    return *reinterpret_cast<const unsigned short*>(&h);
}

// ---- LDS-DMA and hand-scheduled LDS reads ----------------------------------------------------------------------
// The compiler treats an outstanding LDS-DMA as a pending write to ALL of LDS and drains vmcnt(0) in front of the
// next LDS read it can see; the weight-fragment reads therefore go through inline asm with hand-counted waits.
__device__ __forceinline__ void dma16(const void* gptr, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gptr,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
template <int OFF>
__device__ __forceinline__ bf16x8 lds_read_frag(uint32_t addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return __builtin_bit_cast(bf16x8, v);
}
template <int OFF>
__device__ __forceinline__ u32x4 lds_read_b128(uint32_t addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
template <int OFF>
__device__ __forceinline__ void lds_write_b128(uint32_t addr, u32x4 v) {
    asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
// low / high 16 bits of a VGPR
template <int OFF>
__device__ __forceinline__ void lds_write_b16_lo(uint32_t addr, uint32_t v) {
    asm volatile("ds_write_b16 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void lds_write_b16_hi(uint32_t addr, uint32_t v) {
    asm volatile("ds_write_b16_d16_hi %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
template <int OFF>
__device__ __forceinline__ u32x2 lds_read_tr16(uint32_t addr) {
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
__device__ __forceinline__ void lds_write_b32(uint32_t addr, float v) {
    asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ float lds_read_b32(uint32_t addr) {
    float v;
    asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    const bf16x2 t = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, t);
}
template <int N>
__device__ __forceinline__ void wait_lgkm() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// raw s_barrier (no vmcnt(0) fence: LDS-DMA may stay in flight across it) between compiler-level memory fences
__device__ __forceinline__ void block_sync() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void keep_alive(const f32x16& v) { asm volatile("" ::"v"(v)); }
__device__ __forceinline__ uint32_t lds_addr(const char* p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int OFF>
__device__ __forceinline__ void lds_write_b64(uint32_t addr, u32x2 v) {
    asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
__device__ __forceinline__ u32x2 lds_read_b64(uint32_t addr) {
    u32x2 v;
    asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}


// returnless LDS float add (several lanes may hit one address: the LDS serialises them)
__device__ __forceinline__ void lds_add_f32(uint32_t addr, float v) {
    asm volatile("ds_add_f32 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
template <int OFF>
__device__ __forceinline__ void lds_add_f32_off(uint32_t addr, float v) {
    asm volatile("ds_add_f32 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ float lds_read_b32_off(uint32_t addr) {
    float v;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
// 16-byte global load the compiler does not track: no s_waitcnt is inserted for it, the CALLER orders its use behind
// its own vmcnt wait (the compiler's waitcnt pass turns a load carried across a loop iteration into vmcnt(0), which
// would drain the LDS-DMA ring)
__device__ __forceinline__ u32x4 gload16_untracked(const void* p) {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    return v;
}
// s_waitcnt vmcnt(n) for a wave-uniform run-time n (smaller immediates are stricter: n is clamped to 15)
__device__ __forceinline__ void wait_vm_upto(int n) {
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
        case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
    }
}
}  // namespace pce
