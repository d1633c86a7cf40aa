// capi_util.hpp -- small host-side helpers shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <time.h>
#include <math.h>
#include "../../include/sim5gpu.h"
#include "kernels.hpp"
#include "s5_config.hpp"

namespace s5 {

using namespace s5abi;

extern thread_local char g_err[512];
extern DiskConsts g_disk;

void set_error(const char* what, hipError_t e);
int  have_device();
DiskConsts make_disk_consts(double M, double a, double mdot, double alpha = 0.1);
// A flux-table block handed out by attach_flux_table stays PINNED until the FluxPinScope of the calling entry point ends -- by
// then the launches that read it have been enqueued, and a block is only ever freed after a device synchronisation.  A pinned
// block is never freed, whatever the cache retires meanwhile on another thread (ADVICE r4 / VERDICT r5 item 9: until round 5
// the only protection was the distance of 512 newer models between a block's retirement and its release).
struct FluxPinScope {
    FluxPinScope();
    ~FluxPinScope();
    FluxPinScope(const FluxPinScope&) = delete;
    FluxPinScope& operator=(const FluxPinScope&) = delete;
    void* held;                                             // std::vector<std::pair<int, const double*>>*: (device, block)
    void* outer;                                            // the scope this one is nested in, if any
};
int  attach_flux_table(DiskConsts& d);                      // (inside a FluxPinScope)
int  attach_K_table(ImageParams& p);                     // capi_core.hip: universal K(m) table of the fast image kernels (device)                    // capi_core.hip: radial profile table of the fast variant (device)
size_t release_flux_tables();                              // capi_core.hip: every cached flux-table block of every device
void disk_set_mdot(DiskConsts& d, double mdot);
int  disk_lumi(const DiskConsts& d, double* lumi);          // capi_batch.hip: Simpson rule, integrand on the device
int  fill_image_params(const sim5gpu_image_desc* desc, ImageParams& p, bool need_disk = true);

#define S5_HIP(call)                                                          \
    do {                                                                      \
        hipError_t e_ = (call);                                               \
        if (e_ != hipSuccess) { s5::set_error(#call, e_); return SIM5GPU_E_HIP; } \
    } while (0)

// Device staging memory of the batch entry points.  A per-thread grow-only arena replaces a
// hipMalloc/hipFree pair per argument (each tens of microseconds): the SIM5 scalar API of
// sim5_amd/host/sim5lib.c calls these entry points with n = 1, millions of times.  Buffers are carved
// in call order and the arena is rewound when the last buffer of a call is destroyed.
struct Arena {
    char* base = nullptr;
    size_t cap = 0, used = 0;
    size_t demand = 0;          // bytes asked for by the call in progress, including buffers that did not fit
    size_t want = 0;            // largest demand of any call so far: the arena grows to it between calls
    int live = 0;
    int dev = -1;               // device the block lives on; a thread that switches device gets a new block
    void release() { if (base) (void)hipFree(base); base = nullptr; cap = 0; used = 0; }
    void* take(size_t bytes)
    {
        bytes = (bytes + 255) & ~size_t(255);
        if (live == 0) {                                       // first buffer of a call: the block may be replaced
            int cur = 0;
            (void)hipGetDevice(&cur);
            if (cur != dev) { release(); dev = cur; }
            demand = 0;
            size_t need = want > bytes ? want : bytes;
            if (need > cap) {
                release();
                size_t grow = (size_t)1 << 20;
                while (grow < need) grow *= 2;
                if (hipMalloc((void**)&base, grow) != hipSuccess) { base = nullptr; cap = 0; demand += bytes; return nullptr; }
                cap = grow;
            }
        }
        demand += bytes;
        if (used + bytes > cap) return nullptr;                // cannot move buffers that are in use: one-off hipMalloc
        void* p = base + used;
        used += bytes; ++live;
        return p;
    }
    // a buffer of the call that did not come from the block (fallback allocation) has been released
    void call_done() { if (demand > want) want = demand; }
    void give() { if (--live == 0) { used = 0; call_done(); } }

    // Small batches (the n = 1 calls of the SIM5 scalar API, sim5_amd/host/sim5lib.c) are staged in page-locked host
    // memory that the GPU reads and writes in place over the bus: a call is then one launch and one synchronisation
    // instead of a blocking hipMemcpy per argument (measured through tests/c/shim_probe.c: 373 -> see INTEGRATION.md
    // us per ray).  Any device of the process can address the block, so it is not tied to the current device.
    char* pin = nullptr;
    size_t pin_cap = 0, pin_used = 0;
    int pin_live = 0;
    // (a look-ahead batch of the scalar shim -- a row of rays with their 240-byte records -- is staged here too)
    static constexpr size_t PIN_BLOCK = 4 << 20, PIN_LIMIT = 512 << 10;
    void* take_pinned(size_t bytes)
    {
        if (bytes > PIN_LIMIT) return nullptr;
        bytes = (bytes + 63) & ~size_t(63);
        if (!pin) {
            if (pin_cap == (size_t)-1) return nullptr;                       // tried before, not available
            if (hipHostMalloc((void**)&pin, PIN_BLOCK, hipHostMallocDefault) != hipSuccess) { pin = nullptr; pin_cap = (size_t)-1; return nullptr; }
            pin_cap = PIN_BLOCK;
        }
        if (pin_used + bytes > pin_cap) return nullptr;
        void* p = pin + pin_used;
        pin_used += bytes; ++pin_live;
        return p;
    }
    void give_pinned() { if (--pin_live == 0) pin_used = 0; }
    ~Arena() { release(); if (pin) (void)hipHostFree(pin); }   // thread exit: the blocks go back
};
Arena& arena();

// sin and cos of an inclination as the REFERENCE BINARY forms them: gcc merges the sin(i) and cos(i) of geodesic_init_inf (ref
// src/sim5kerr-geod.c:73-77) into ONE call of glibc's sincos() (read off the disassembly of the reference library built here with the
// reference's own flags), and sincos() and cos() are different routines that differ in the last bit for some arguments
// -- which reaches q and, on the central column of an odd-width image, the class of a pixel (found by the randomised campaign of
// round 5: a = 0.9999, i = 40.2 deg).  Every host-side sin i / cos i of the library comes from here.
inline void reference_sincos(double x, double& s, double& c) { ::sincos(x, &s, &c); }

// The stream of the calling HOST THREAD (capi_core.hip): every batch entry point launches on it and waits for it alone, so
// host threads that call the per-ray functions concurrently (ref README.md:16,202: "thread-safe", the OpenMP'd caller of
// SURVEY 8(d)) neither serialise on the legacy null stream nor wait for each other's kernels in a device-wide
// synchronisation.  Non-blocking (no implicit ordering with the null stream), one per thread and device, made on first use.
hipStream_t thread_stream();

// RAII device buffer for the host-array (batch) entry points
template <typename T>
struct DevBuf {
    T* ptr = nullptr;
    size_t n = 0;
    bool failed = false;
    bool from_arena = false;
    bool pinned = false;            // ptr is page-locked host memory the kernel accesses in place
    void alloc()
    {
        if (!n) return;
        ptr = (T*)arena().take_pinned(n * sizeof(T));
        if (ptr) { pinned = true; return; }
        ptr = (T*)arena().take(n * sizeof(T));
        if (ptr) { from_arena = true; return; }
        if (hipMalloc((void**)&ptr, n * sizeof(T)) != hipSuccess) { ptr = nullptr; failed = true; }
    }
    explicit DevBuf(size_t count) : n(count) { alloc(); }
    DevBuf(const T* host, size_t count) : n(host ? count : 0)
    {
        alloc();
        if (ptr && pinned) memcpy(ptr, host, n * sizeof(T));
        else if (ptr && hipMemcpy(ptr, host, n * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) failed = true;
    }
    ~DevBuf()
    {
        if (!ptr) return;
        if (pinned) arena().give_pinned();
        else if (from_arena) arena().give(); else { (void)hipFree(ptr); arena().call_done(); }
    }
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    bool ok() const { return !failed; }
    hipError_t to_host(T* host) const
    {
        if (!n || !host || !ptr) return hipSuccess;
        if (pinned) { memcpy(host, ptr, n * sizeof(T)); return hipSuccess; }      // the kernel has been waited for (run_map)
        return hipMemcpy(host, ptr, n * sizeof(T), hipMemcpyDeviceToHost);
    }
};

// ---------------------------------------------------------------------------------------------------------------------------
// the batch entry points' kernel runner: one lane per element
// ---------------------------------------------------------------------------------------------------------------------------
template <typename F>
__global__ __launch_bounds__(256) void map_rays(size_t n, F body)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) body(i);
}

// the same for ONE workgroup's worth of rays, announcing its end itself: every thread makes its stores visible to the host,
// the workgroup meets, thread 0 raises a word in page-locked host memory.  The host watches that word instead of asking the
// runtime for the stream's state: the end-of-kernel signal, its interrupt-less polling through the runtime and the
// queue's bookkeeping are off the caller's critical path (they complete behind the next call's set-up)
template <typename F>
__global__ __launch_bounds__(256) void map_rays_flag(size_t n, F body, int* done)
{
    const size_t i = threadIdx.x;
    if (i < n) body(i);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// wait for the word a small launch raises at its end (map_rays_flag below, k_chain.hip); now and then the runtime is asked
// too, so that a launch that died -- it will never raise the word -- ends the wait with its error.  The wait is bounded in
// WALL-CLOCK time (S5_DONE_TIMEOUT_S with the stream idle and the word still down) and its expiry has an error of its own.
constexpr double S5_DONE_TIMEOUT_S = 2.0;
inline double wall_seconds()
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
inline void cpu_relax()
{
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#endif
}
inline hipError_t wait_done_word(int* done, hipStream_t stream)
{
    hipError_t e = hipSuccess;
    double idle_since = -1.0;
    for (unsigned spin = 1; !__atomic_load_n(done, __ATOMIC_ACQUIRE); ++spin) {
        cpu_relax();
        if ((spin & 255u) == 0u) {
            e = hipStreamQuery(stream);
            if (e == hipErrorNotReady) { e = hipSuccess; idle_since = -1.0; continue; }
            if (e != hipSuccess) break;
            if (__atomic_load_n(done, __ATOMIC_ACQUIRE)) break;
            // the stream is idle and the word is still down: the store is on its way -- or the kernel never ran
            const double now = wall_seconds();
            if (idle_since < 0.0) idle_since = now;
            else if (now - idle_since > S5_DONE_TIMEOUT_S) { e = hipErrorLaunchTimeOut; break; }
        }
    }
    return e;
}
inline int* take_done_word()
{
    static const bool off = getenv("SIM5GPU_NO_DONE_FLAG") != nullptr;       // (debugging: the stream is polled instead)
    if (off) return nullptr;
    int* done = (int*)arena().take_pinned(64);
    if (done) __atomic_store_n(done, 0, __ATOMIC_RELEASE);
    return done;
}

template <typename F>
int run_batch(size_t n, F body, const char* what)
{
    if (n == 0) return SIM5GPU_OK;
    hipError_t e;
    hipStream_t stream = thread_stream();
    // a handful of rays (the n = 1 calls of the SIM5 scalar API): the caller waits for ONE short kernel, and the blocking
    // wait's wake-up costs more than the kernel (measured through tests/tools/shim_rate.sh, shim_latency.py)
    int* done = (n <= 256) ? take_done_word() : nullptr;
    if (done) {
        hipLaunchKernelGGL(map_rays_flag<F>, dim3(1), dim3(256), 0, stream, n, body, done);
        e = hipGetLastError();
        if (e == hipSuccess) e = wait_done_word(done, stream);
        arena().give_pinned();
    } else {
        const unsigned blocks = (unsigned)((n + 255) / 256);
        hipLaunchKernelGGL(map_rays<F>, dim3(blocks), dim3(256), 0, stream, n, body);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
    }
    if (e != hipSuccess) { set_error(what, e); return SIM5GPU_E_HIP; }
    return SIM5GPU_OK;
}

} // namespace s5
