// Fiber-based workgroup emulator (TEST INFRASTRUCTURE ONLY, see rl_emu.h).
//
// One OS thread runs one workgroup at a time.  Every GPU thread of the
// workgroup is a ucontext fiber; the scheduler resumes fibers round-robin and
// a fiber returns to it either by finishing or by reaching __syncthreads().
// After one sweep every live fiber sits at the same barrier, which is the
// barrier's semantics.  Workgroups of one launch are spread over a few OS
// threads.
#include "rl_emu.h"

#include <ucontext.h>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

thread_local rl_emu_uint3 threadIdx, blockIdx, blockDim, gridDim;

namespace {

constexpr size_t kStackBytes = 96 * 1024;

struct Worker {
    ucontext_t sched;
    std::vector<ucontext_t> ctx;
    unsigned char* stacks = nullptr;
    size_t stacks_cap = 0;
    ~Worker() { std::free(stacks); }
    std::vector<char> done;
    std::vector<unsigned char> smem;
    const std::function<void()>* body = nullptr;
    int current = -1;
};

thread_local Worker* tl_worker = nullptr;

void fiber_entry() {
    Worker* w = tl_worker;
    (*w->body)();
    w->done[w->current] = 1;
    swapcontext(&w->ctx[w->current], &w->sched);
}

void run_block(Worker& w, dim3 block, size_t smem_bytes) {
    const unsigned nthr = block.x * block.y * block.z;
    if (w.ctx.size() < nthr) w.ctx.resize(nthr);
    if (w.stacks_cap < size_t(nthr) * kStackBytes) {
        std::free(w.stacks);
        w.stacks_cap = size_t(nthr) * kStackBytes;
        w.stacks = static_cast<unsigned char*>(std::malloc(w.stacks_cap));
    }
    w.done.assign(nthr, 0);
    w.smem.assign(smem_bytes + 64, 0xCD);  // poison: uninitialised LDS reads show up
    for (unsigned t = 0; t < nthr; ++t) {
        getcontext(&w.ctx[t]);
        w.ctx[t].uc_stack.ss_sp = w.stacks + size_t(t) * kStackBytes;
        w.ctx[t].uc_stack.ss_size = kStackBytes;
        w.ctx[t].uc_link = &w.sched;
        makecontext(&w.ctx[t], fiber_entry, 0);
    }
    unsigned remaining = nthr;
    while (remaining) {
        for (unsigned t = 0; t < nthr; ++t) {
            if (w.done[t]) continue;
            threadIdx.x = t % block.x;
            threadIdx.y = (t / block.x) % block.y;
            threadIdx.z = t / (block.x * block.y);
            w.current = int(t);
            swapcontext(&w.sched, &w.ctx[t]);
            if (w.done[t]) --remaining;
        }
    }
}

int g_device = 0;

}  // namespace

void rl_emu_syncthreads() {
    Worker* w = tl_worker;
    swapcontext(&w->ctx[w->current], &w->sched);
}

unsigned char* rl_emu_smem() {
    // 16-byte aligned view of the block's LDS image
    auto p = reinterpret_cast<uintptr_t>(tl_worker->smem.data());
    return reinterpret_cast<unsigned char*>((p + 15) & ~uintptr_t(15));
}

// Persistent pool: worker threads (and their fiber stacks) live for the whole
// process; a launch publishes a job and every worker pulls block indices from
// one atomic counter.
namespace {

struct Job {
    dim3 grid, block;
    size_t smem = 0;
    const std::function<void()>* body = nullptr;
    size_t nblocks = 0;
    std::atomic<size_t> next{0};
};

struct Pool {
    std::mutex mu;
    std::condition_variable cv_start, cv_done;
    std::vector<std::thread> threads;
    Job* job = nullptr;
    uint64_t generation = 0;
    unsigned busy = 0;
    bool stop = false;

    static void drain(Job& j, Worker& w) {
        w.body = j.body;
        tl_worker = &w;
        gridDim = {j.grid.x, j.grid.y, j.grid.z};
        blockDim = {j.block.x, j.block.y, j.block.z};
        for (;;) {
            size_t b = j.next.fetch_add(1);
            if (b >= j.nblocks) break;
            blockIdx.x = unsigned(b % j.grid.x);
            blockIdx.y = unsigned((b / j.grid.x) % j.grid.y);
            blockIdx.z = unsigned(b / (size_t(j.grid.x) * j.grid.y));
            run_block(w, j.block, j.smem);
        }
        tl_worker = nullptr;
    }

    void worker_main() {
        Worker w;
        uint64_t seen = 0;
        for (;;) {
            Job* j;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_start.wait(lk, [&] { return stop || generation != seen; });
                if (stop) return;
                seen = generation;
                j = job;
            }
            drain(*j, w);
            {
                std::lock_guard<std::mutex> lk(mu);
                if (--busy == 0) cv_done.notify_all();
            }
        }
    }

    void ensure(unsigned n) {
        while (threads.size() < n) threads.emplace_back([this] { worker_main(); });
    }

    ~Pool() {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv_start.notify_all();
        for (auto& t : threads) t.join();
    }
};

Pool& pool() {
    static Pool p;
    return p;
}

}  // namespace

struct rl_emu_graph {
    std::vector<std::function<void()>> nodes;
    int refs = 1;
};
static rl_emu_graph* g_capture = nullptr;

static void run_launch(dim3 grid, dim3 block, size_t smem_bytes,
                       const std::function<void()>& body);

void rl_emu_launch(dim3 grid, dim3 block, size_t smem_bytes,
                   const std::function<void()>& body) {
    if (g_capture) {
        std::function<void()> copy = body;
        g_capture->nodes.push_back(
            [=]() { run_launch(grid, block, smem_bytes, copy); });
        return;
    }
    run_launch(grid, block, smem_bytes, body);
}

static void run_launch(dim3 grid, dim3 block, size_t smem_bytes,
                       const std::function<void()>& body) {
    Job j;
    j.grid = grid;
    j.block = block;
    j.smem = smem_bytes;
    j.body = &body;
    j.nblocks = size_t(grid.x) * grid.y * grid.z;
    if (j.nblocks == 0) return;
    static thread_local Worker main_worker;
    unsigned helpers = std::thread::hardware_concurrency();
    helpers = helpers > 1 ? helpers - 1 : 0;
    if (helpers > 7) helpers = 7;
    if (j.nblocks < 4) helpers = 0;
    Pool& p = pool();
    if (helpers) {
        std::lock_guard<std::mutex> lk(p.mu);
        p.ensure(helpers);
        p.job = &j;
        p.busy = unsigned(p.threads.size());
        ++p.generation;
    }
    if (helpers) p.cv_start.notify_all();
    Pool::drain(j, main_worker);
    if (helpers) {
        std::unique_lock<std::mutex> lk(p.mu);
        p.cv_done.wait(lk, [&] { return p.busy == 0; });
        p.job = nullptr;
    }
}

// --- runtime API stand-ins --------------------------------------------------
hipError_t hipMalloc(void** p, size_t bytes) {
    *p = nullptr;
    if (bytes == 0) bytes = 16;
    if (posix_memalign(p, 256, bytes) != 0) return hipErrorOutOfMemory;
    std::memset(*p, 0xAB, bytes);  // device memory is not zero-initialised
    return hipSuccess;
}
hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
// (host memory stands in for the device's: a fixed, generous answer)
hipError_t hipMemGetInfo(size_t* free_bytes, size_t* total_bytes) {
    if (free_bytes) *free_bytes = (size_t)16 << 30;
    if (total_bytes) *total_bytes = (size_t)32 << 30;
    return hipSuccess;
}
hipError_t hipHostMalloc(void** p, size_t bytes, unsigned) { return hipMalloc(p, bytes); }
hipError_t hipHostFree(void* p) { return hipFree(p); }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) {
    std::memmove(d, s, n);
    return hipSuccess;
}
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k,
                          hipStream_t) {
    if (g_capture) {
        g_capture->nodes.push_back([=]() { std::memmove(d, s, n); });
        return hipSuccess;
    }
    return hipMemcpy(d, s, n, k);
}
hipError_t hipMemset(void* p, int v, size_t n) {
    std::memset(p, v, n);
    return hipSuccess;
}
hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) {
    return hipMemset(p, v, n);
}
hipError_t hipSetDevice(int d) { g_device = d; return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = g_device; return hipSuccess; }
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t* prop, int) {
    prop->multiProcessorCount = 2;   // few "compute units": exercises the persistent loops
    return hipSuccess;
}
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipGetLastError() { return hipSuccess; }
hipError_t hipPeekAtLastError() { return hipSuccess; }
const char* hipGetErrorString(hipError_t e) {
    return e == hipSuccess ? "hipSuccess" : "emulated HIP error";
}

struct rl_emu_event { std::chrono::steady_clock::time_point t; };
hipError_t hipEventCreate(hipEvent_t* e) { *e = new rl_emu_event; return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) {
    e->t = std::chrono::steady_clock::now();
    return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}
hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }

hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) {
    *s = reinterpret_cast<hipStream_t>(uintptr_t(0x51));
    return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) {
    if (g_capture) return hipErrorInvalidValue;
    g_capture = new rl_emu_graph;
    return hipSuccess;
}
hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* graph) {
    *graph = g_capture;
    g_capture = nullptr;
    return *graph ? hipSuccess : hipErrorInvalidValue;
}
hipError_t hipStreamIsCapturing(hipStream_t, hipStreamCaptureStatus* status) {
    *status = g_capture ? hipStreamCaptureStatusActive : hipStreamCaptureStatusNone;
    return hipSuccess;
}
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
// every emulated stream runs its work at launch time, in order: nothing to wait for
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipGraphInstantiate(hipGraphExec_t* exec, hipGraph_t graph, void*, void*, size_t) {
    graph->refs += 1;
    *exec = graph;
    return hipSuccess;
}
hipError_t hipGraphLaunch(hipGraphExec_t exec, hipStream_t) {
    for (auto& node : exec->nodes) node();
    return hipSuccess;
}
static void graph_unref(rl_emu_graph* g) {
    if (g && --g->refs == 0) delete g;
}
hipError_t hipGraphExecDestroy(hipGraphExec_t exec) { graph_unref(exec); return hipSuccess; }
hipError_t hipGraphDestroy(hipGraph_t graph) { graph_unref(graph); return hipSuccess; }
