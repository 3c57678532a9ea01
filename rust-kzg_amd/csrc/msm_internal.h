// Internal C++ interface of the MSM engine (shared by msm.hip and the c-kzg layer).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace kzgamd {
struct MsmContext;
// points: blst_p1_affine[n] (host or device); prepare = build fixed-base rows
MsmContext* msm_create(const void* points, size_t n, bool points_on_device, bool prepare);
void msm_destroy(MsmContext* ctx);
// enqueue nbatch MSMs (device pointers, no sync); d_out = blst_p1[nbatch]
void msm_enqueue(MsmContext* ctx, void* d_out, const void* d_scalars, size_t npoints, size_t nbatch, int mont,
                 hipStream_t stream);
void msm_run_host(MsmContext* ctx, void* out, const void* scalars, size_t npoints, size_t nbatch);
}  // namespace kzgamd
