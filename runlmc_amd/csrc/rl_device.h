// Device-side basics shared by every kernel: complex fp64 helpers, the LDS
// carve macro and the launch macro.
//
// The product build is hipcc --offload-arch=gfx950.  RL_EMU is defined only by
// tests/emu (a thread-level emulator that runs this same source on the host
// for debugging and sanitizers); nothing shipped is built that way.
#pragma once

#if defined(RL_EMU)
#include "rl_emu.h"
#define RL_SMEM(name) unsigned char* name = rl_emu_smem()
#define RL_LAUNCH(kern, grid, block, smem, stream, ...) \
    rl_emu_launch((grid), (block), (smem), [=]() { (kern)(__VA_ARGS__); })
#define RL_BACKEND_NAME "emu-host"
#else
#include <hip/hip_runtime.h>
// all LDS lives in ONE dynamic array whose base is 16-byte aligned
// (cdna_hip_programming.md guideline 17)
#define RL_SMEM(name)                                                        \
    extern __shared__ __attribute__((aligned(16))) unsigned char name##_lds[]; \
    unsigned char* name = name##_lds
#define RL_LAUNCH(kern, grid, block, smem, stream, ...) \
    hipLaunchKernelGGL(kern, (grid), (block), (smem), (stream), __VA_ARGS__)
#define RL_BACKEND_NAME "hip-gfx950"
#endif

#include <stdint.h>

// A pointer to data that NO thread of the running kernel writes, read at a wave-uniform
// address: the constant address space makes the compiler fetch it through the scalar
// unit (s_load into SGPRs).  Needed where a kernel also stores through pointers the
// compiler cannot tell apart from this one (struct members): there it falls back to one
// vector load per lane and drains the vector-memory counter around it.
#if defined(RL_EMU)
typedef const double* rl_kconst;
#define RL_KCONST(p) (p)
#else
typedef const double __attribute__((address_space(4)))* rl_kconst;
#define RL_KCONST(p) ((rl_kconst)(unsigned long long)(p))
#endif


// One element of a row block addressed as  wave-uniform base + per-lane byte offset  (both
// < 2^31): a buffer access -- the base lives in scalar registers, ONE offset register serves
// every access of a loop over bases, and an offset at or past `nbytes` reads 0 / stores
// nothing (the hardware's range check: no clamped index, no masked store).
#if defined(RL_EMU)
__device__ __forceinline__ double rl_row_load(const double* base, unsigned nbytes, unsigned off) {
    return off < nbytes ? *reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + off) : 0.0;
}
__device__ __forceinline__ void rl_row_store(double* base, unsigned nbytes, unsigned off, double v) {
    if (off < nbytes) *reinterpret_cast<double*>(reinterpret_cast<char*>(base) + off) = v;
}
#else
typedef unsigned int rl_u32x2 __attribute__((ext_vector_type(2)));
#define RL_BUFFER_WORD3 0x00020000       // raw buffer, 32-bit data format (gfx90a .. gfx950)
__device__ __forceinline__ double rl_row_load(const double* base, unsigned nbytes, unsigned off) {
    const __amdgpu_buffer_rsrc_t r =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(base), 0, (int)nbytes, RL_BUFFER_WORD3);
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 0));
}
__device__ __forceinline__ void rl_row_store(double* base, unsigned nbytes, unsigned off, double v) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)nbytes, RL_BUFFER_WORD3);
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(rl_u32x2, v), r, (int)off, 0, 0);
}
#endif

// RL_TIMING (experiment builds only: python -m runlmc_amd.build --timing):
// kernels stamp s_memtime at their phase boundaries into a global buffer that
// rl_debug_timing() reads back -- the only way to see where a latency-bound
// kernel's microseconds go.
#if defined(RL_TIMING) && !defined(RL_EMU)
__device__ long long rl_timing_buf[256];
#define RL_STAMP(slot)                                                           \
    do {                                                                         \
        if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) \
            rl_timing_buf[slot] = wall_clock64();                                \
    } while (0)
// the same for a workgroup in the middle of a launch (steady state: the chip is
// full and the memory system loaded), chosen by its x / y block index
#define RL_STAMP_AT(slot, bx, by)                                                \
    do {                                                                         \
        if (blockIdx.x == (bx) && blockIdx.y == (by) && blockIdx.z == 0 && threadIdx.x == 0) \
            rl_timing_buf[slot] = wall_clock64();                                \
    } while (0)
// the same under any condition (e.g. one thread of one workgroup at its n-th tile)
#define RL_STAMP_IF(slot, cond)                                                  \
    do {                                                                         \
        if (cond) rl_timing_buf[slot] = wall_clock64();                          \
    } while (0)
// census of concurrently resident workgroups of a launch (slot: current count,
// slot + 1: the most seen): ENTER first thing in the kernel, LEAVE last
#define RL_CENSUS_ENTER(slot)                                                    \
    do {                                                                         \
        if (threadIdx.x == 0) {                                                  \
            const long long now = atomicAdd((unsigned long long*)&rl_timing_buf[slot], 1ull) + 1; \
            atomicMax((unsigned long long*)&rl_timing_buf[(slot) + 1], (unsigned long long)now);   \
        }                                                                        \
    } while (0)
#define RL_CENSUS_LEAVE(slot)                                                    \
    do {                                                                         \
        if (threadIdx.x == 0) atomicAdd((unsigned long long*)&rl_timing_buf[slot], ~0ull); \
    } while (0)
#else
#define RL_STAMP(slot) do { } while (0)
#define RL_STAMP_AT(slot, bx, by) do { } while (0)
#define RL_STAMP_IF(slot, cond) do { } while (0)
#define RL_CENSUS_ENTER(slot) do { } while (0)
#define RL_CENSUS_LEAVE(slot) do { } while (0)
#endif

struct __attribute__((aligned(16))) cplx {
    double x, y;
};

__device__ __forceinline__ cplx c_make(double x, double y) {
    cplx r;
    r.x = x;
    r.y = y;
    return r;
}
__device__ __forceinline__ cplx c_add(cplx a, cplx b) { return c_make(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cplx c_sub(cplx a, cplx b) { return c_make(a.x - b.x, a.y - b.y); }
// a * b
__device__ __forceinline__ cplx c_mul(cplx a, cplx b) {
    return c_make(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
// a * conj(b)
__device__ __forceinline__ cplx c_mulc(cplx a, cplx b) {
    return c_make(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}
// multiply by -i (forward quarter turn) / +i
__device__ __forceinline__ cplx c_mul_mi(cplx a) { return c_make(a.y, -a.x); }
__device__ __forceinline__ cplx c_mul_pi(cplx a) { return c_make(-a.y, a.x); }
__device__ __forceinline__ cplx c_scale(cplx a, double s) { return c_make(a.x * s, a.y * s); }

// q = w / d for small w via a precomputed magic = floor(2^32 / d) + 1
// (exact while w * d < 2^32); the host passes `magic`
__device__ __forceinline__ unsigned fast_div(unsigned w, unsigned magic) {
#if defined(RL_EMU)
    return (unsigned)(((unsigned long long)w * magic) >> 32);
#else
    return __umulhi(w, magic);
#endif
}

#define RL_MAX_PASSES 8
// Radix schedule of one power-of-two FFT; passed to kernels by value.
struct FftPlan {
    int n;
    int npass;
    int radix[RL_MAX_PASSES];
};

