// Thread-level emulator for the HIP kernels under runlmc_amd/csrc.
//
// TEST INFRASTRUCTURE ONLY.  This header lets the *same* kernel source that
// hipcc compiles for gfx950 be compiled by g++ and executed on host threads,
// one fiber per GPU thread, with __syncthreads() as a real barrier, so that
// index arithmetic, barrier placement and out-of-bounds accesses can be
// debugged (and run under ASan/UBSan) in a container that has no GPU.  It is
// not a fallback: runlmc_amd never loads a library built with it unless a
// test hands it the path explicitly, and no number in bench.py comes from it.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#ifndef __restrict__
#define __restrict__ __restrict
#endif

struct rl_emu_uint3 { unsigned x, y, z; };
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

extern thread_local rl_emu_uint3 threadIdx, blockIdx, blockDim, gridDim;

void rl_emu_syncthreads();
unsigned char* rl_emu_smem();
void rl_emu_launch(dim3 grid, dim3 block, size_t smem_bytes,
                   const std::function<void()>& body);

#define __syncthreads() rl_emu_syncthreads()
// workgroups run on several host threads: real atomics and fences
static inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }

// --- the sliver of the HIP runtime API the host code uses -----------------
typedef int hipError_t;
typedef void* hipStream_t;
typedef struct rl_emu_event* hipEvent_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost,
                     hipMemcpyDeviceToDevice, hipMemcpyDefault };

hipError_t hipMalloc(void** p, size_t bytes);
hipError_t hipFree(void* p);
hipError_t hipMemGetInfo(size_t* free_bytes, size_t* total_bytes);
enum { hipHostMallocDefault = 0 };
hipError_t hipHostMalloc(void** p, size_t bytes, unsigned flags);
hipError_t hipHostFree(void* p);
hipError_t hipMemcpy(void* dst, const void* src, size_t n, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t n,
                          hipMemcpyKind k, hipStream_t s);
hipError_t hipMemset(void* p, int v, size_t n);
hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t s);
hipError_t hipSetDevice(int d);
hipError_t hipGetDevice(int* d);
hipError_t hipGetDeviceCount(int* n);
struct hipDeviceProp_t {
    int multiProcessorCount;
};
hipError_t hipGetDeviceProperties(hipDeviceProp_t* prop, int device);
hipError_t hipDeviceSynchronize();
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipGetLastError();
hipError_t hipPeekAtLastError();
const char* hipGetErrorString(hipError_t e);
hipError_t hipEventCreate(hipEvent_t* e);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
hipError_t hipFuncSetAttribute(const void* f, int attr, int value);
// streams + graph capture: while a capture is open, launches and async copies
// are recorded (not run); hipGraphLaunch runs the recording.
typedef struct rl_emu_graph* hipGraph_t;
typedef struct rl_emu_graph* hipGraphExec_t;
enum { hipStreamNonBlocking = 1 };
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal, hipStreamCaptureModeThreadLocal,
                            hipStreamCaptureModeRelaxed };
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamBeginCapture(hipStream_t s, hipStreamCaptureMode mode);
hipError_t hipStreamEndCapture(hipStream_t s, hipGraph_t* graph);
enum hipStreamCaptureStatus { hipStreamCaptureStatusNone, hipStreamCaptureStatusActive };
hipError_t hipStreamIsCapturing(hipStream_t s, hipStreamCaptureStatus* status);
enum { hipEventDisableTiming = 2 };
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned flags);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);
hipError_t hipGraphInstantiate(hipGraphExec_t* exec, hipGraph_t graph, void*, void*, size_t);
hipError_t hipGraphLaunch(hipGraphExec_t exec, hipStream_t s);
hipError_t hipGraphExecDestroy(hipGraphExec_t exec);
hipError_t hipGraphDestroy(hipGraph_t graph);
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
