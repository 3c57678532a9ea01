// Every entry point of the library runs on the GPU its handle was created on and leaves the caller's
// current device as it found it: one host process can own the 8 GPUs of a node (one settings object /
// MSM handle / NTT handle per device) and call into any of them from any thread.
#pragma once
#include <hip/hip_runtime.h>

namespace kzgamd {

struct DeviceGuard {
    int prev = -1;
    bool changed = false;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int dev) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != dev) {
            err = hipSetDevice(dev);
            changed = err == hipSuccess;
            // a refused device is reported through `err` only: the runtime's sticky last-error must not fail the
            // next, unrelated call's hipGetLastError() check
            if (!changed) (void)hipGetLastError();
        }
    }
    ~DeviceGuard() {
        if (changed) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

}  // namespace kzgamd
