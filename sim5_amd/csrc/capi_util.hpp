// capi_util.hpp -- small host-side helpers shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../../include/sim5gpu.h"
#include "kernels.hpp"
#include "s5_config.hpp"

namespace s5 {

using namespace s5abi;

extern thread_local char g_err[512];
extern DiskConsts g_disk;

void set_error(const char* what, hipError_t e);
int  have_device();
DiskConsts make_disk_consts(double M, double a, double mdot);
int  fill_image_params(const sim5gpu_image_desc* desc, ImageParams& p);

#define S5_HIP(call)                                                          \
    do {                                                                      \
        hipError_t e_ = (call);                                               \
        if (e_ != hipSuccess) { s5::set_error(#call, e_); return SIM5GPU_E_HIP; } \
    } while (0)

// RAII device buffer for the host-array (batch) entry points
template <typename T>
struct DevBuf {
    T* ptr = nullptr;
    size_t n = 0;
    bool failed = false;
    explicit DevBuf(size_t count) : n(count)
    {
        if (n && hipMalloc((void**)&ptr, n * sizeof(T)) != hipSuccess) { ptr = nullptr; failed = true; }
    }
    DevBuf(const T* host, size_t count) : n(count)
    {
        if (n && host) {
            if (hipMalloc((void**)&ptr, n * sizeof(T)) != hipSuccess) { ptr = nullptr; failed = true; return; }
            if (hipMemcpy(ptr, host, n * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) failed = true;
        }
    }
    ~DevBuf() { if (ptr) (void)hipFree(ptr); }
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    bool ok() const { return !failed; }
    hipError_t to_host(T* host) const
    {
        if (!n || !host || !ptr) return hipSuccess;
        return hipMemcpy(host, ptr, n * sizeof(T), hipMemcpyDeviceToHost);
    }
};

} // namespace s5
