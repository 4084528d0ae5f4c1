// LDS-DMA and hand-scheduled LDS access helpers shared by the output-stationary kernels (gfx950).
//
// hipcc's waitcnt pass treats an outstanding LDS-DMA (global_load_lds) as a pending write to ALL of LDS and drains
// vmcnt(0) in front of every LDS read it can see, so kernels that keep DMA in flight across their LDS reads issue
// those reads as inline asm with hand-counted lgkmcnt and use the raw barrier below.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace mkdma {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(const void* gptr, char* lds_wave_base) {       // 64 lanes x 16 B -> base + 16 lane
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gptr,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
template <int OFF>
__device__ __forceinline__ u32x4 lds_read_b128(uint32_t addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
template <int N>
__device__ __forceinline__ void wait_lgkm() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// raw s_barrier (no vmcnt(0) fence: LDS-DMA may stay in flight across it) between compiler-level memory fences
__device__ __forceinline__ void block_sync() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ uint32_t lds_addr(const char* p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}

}  // namespace mkdma
