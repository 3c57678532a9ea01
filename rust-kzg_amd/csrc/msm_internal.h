// Internal C++ interface of the MSM engine (shared by msm.hip and the c-kzg layer).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace kzgamd {
struct MsmContext;
// OUT_WINDOWS (unprepared handles): one Jacobian point per (MSM, window); the caller does the Horner steps
// OUT_XYZZ (wide-table handles): g1::Xyzz sums, for consumers on the device (the G1 transforms of FK20)
enum { OUT_JACOBIAN = 0, OUT_COMPRESSED = 1, OUT_WINDOWS = 2, OUT_XYZZ = 3 };
// What the caller knows about the bases' membership in the r-torsion subgroup G1 (the GLV split of the engines is an
// identity of G1 only): G1_TRUSTED = all in G1 (just decoded and checked, or a validated setup), G1_CHECK = test them
// at creation and fall back to the unsplit engine if one fails, G1_NO_SPLIT = never split.
enum { G1_TRUSTED = 0, G1_CHECK = 1, G1_NO_SPLIT = 2 };
// points: blst_p1_affine[n] (host or device) or g1::AffPt[n] (device); prepare = build fixed-base rows
// opt: the handle's configuration (config.h); nullptr = defaults and environment
struct Options;
MsmContext* msm_create(const void* points, size_t n, bool points_on_device, bool prepare, bool points_are_affpt,
                       int g1_policy = G1_TRUSTED, const Options* opt = nullptr);
void msm_destroy(MsmContext* ctx);
// variable-base handle (prepare == false) over new device-resident AffPt bases: same as destroying it and creating
// another, without the stream, the allocations and their synchronisations
void msm_reset_points(MsmContext* ctx, const void* d_affpts, size_t n);
// enqueue nbatch MSMs (device pointers, no sync); d_out = blst_p1[nbatch] or 48-byte compressed points
// nseg != 0 (wide-table handles): MSM b runs over the table bases (b % nseg) * npoints .. + npoints — nseg different
// small base sets in one handle (the 128 columns of 64 points of FK20)
void msm_enqueue(MsmContext* ctx, void* d_out, const void* d_scalars, size_t npoints, size_t nbatch, int mont,
                 hipStream_t stream, int out_mode, bool reserve_only = false, size_t nseg = 0);
bool msm_has_wide_table(MsmContext* ctx);
// true when `stream` has a workspace of its own on this handle (created if there is room): its enqueues use no events
// and may be captured into a graph
bool msm_private_workspace(MsmContext* ctx, hipStream_t stream);
// 48-byte compressed form of `count` g1::Xyzz points (device pointers)
void g1_compress_xyzz(void* d_out48, const void* d_xyzz, size_t count, hipStream_t stream);
int msm_device(MsmContext* ctx);
void msm_run_host(MsmContext* ctx, void* out, const void* scalars, size_t npoints, size_t nbatch, size_t nseg = 0);
void msm_lock(MsmContext* ctx);
void msm_unlock(MsmContext* ctx);
// per-kernel timing of the last enqueue (events on the launch stream); ms < 0 when disabled
void msm_set_profile(MsmContext* ctx, bool on);
int msm_get_profile(MsmContext* ctx, float* accum_ms, float* total_ms);  // number of enqueues averaged
}  // namespace kzgamd
