// Shared helpers for the gfx950 kernels of libsis3d_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include "sis3d.h"

#define SIS3D_WAVE 64

void sis3d_record_hip_error(hipError_t e);

static inline int sis3d_check_launch()
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        sis3d_record_hip_error(e);
        return SIS3D_ELAUNCH;
    }
    return SIS3D_OK;
}

// grant a kernel the largest dynamic LDS size it can ever be launched with (160 KB minus its static LDS).  Called ONCE per kernel
// and device (sis3d_grant_lds below), never per launch and never with a smaller value later.
static inline hipError_t sis3d_allow_max_lds(const void *kern)
{
    hipFuncAttributes fa;
    hipError_t e = hipFuncGetAttributes(&fa, kern);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - (int)fa.sharedSizeBytes);
}

// Fill `bytes` (a multiple of 4, 4-byte aligned) with a 32-bit pattern: a KERNEL of this library, not hipMemsetAsync.  A
// hipMemsetAsync captured in a HIP graph becomes a memset node, and on ROCm 7.2 replaying a graph with a memset node after the host
// has synchronised and launched anything else faults ("Memory access fault by GPU", tools/hipgraph_memset_repro.py: no sis3d code
// involved) -- the root cause of the image-path hazard of rounds 1-2.  Defined in api.hip.
int sis3d_fill32(void *dst, uint32_t pattern, size_t bytes, hipStream_t st);

// The attribute belongs to (kernel, DEVICE): grant it once per device the process launches on, not once per process, and do not
// latch a failed attempt.  `once` is a function-local static at the call site = one per kernel instantiation; bit d =
// "granted on device ordinal d" (ordinals >= 64 are simply re-granted on every launch).  Still never per launch on a device that has
// it, so no attribute write coincides with the enqueue of a captured graph that contains the kernel.  bytes <= 0: the maximum.
struct Sis3dLdsOnce { std::atomic<uint64_t> granted{0}; };
static inline int sis3d_grant_lds(Sis3dLdsOnce &once, const void *kern, int bytes)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return SIS3D_ELAUNCH;
    const uint64_t bit = (dev >= 0 && dev < 64) ? (1ull << dev) : 0;
    if (bit && (once.granted.load(std::memory_order_acquire) & bit)) return SIS3D_OK;
    const hipError_t e = bytes > 0 ? hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) : sis3d_allow_max_lds(kern);
    if (e != hipSuccess) {
        sis3d_record_hip_error(e);
        return SIS3D_ELAUNCH;
    }
    once.granted.fetch_or(bit, std::memory_order_release);
    return SIS3D_OK;
}

static inline hipStream_t as_stream(sis3d_stream_t s) { return (hipStream_t)s; }
static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
