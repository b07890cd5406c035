// Library-level entry points (version, error strings).
#include "common.h"
#include <string.h>

static thread_local char g_hip_err[256] = "";

void sis3d_record_hip_error(hipError_t e)
{
    strncpy(g_hip_err, hipGetErrorString(e), sizeof(g_hip_err) - 1);
}

namespace {
__global__ __launch_bounds__(256) void fill32_kernel(uint4 *__restrict__ dst4, uint32_t *__restrict__ tail, uint32_t pattern, size_t n4, size_t ntail)
{
    const uint4 v = make_uint4(pattern, pattern, pattern, pattern);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) dst4[i] = v;
    if (blockIdx.x == 0 && threadIdx.x < ntail) tail[threadIdx.x] = pattern;
}
} // namespace

int sis3d_fill32(void *dst, uint32_t pattern, size_t bytes, hipStream_t st)
{
    if (bytes == 0) return SIS3D_OK;
    if (!dst || (bytes & 3) || ((uintptr_t)dst & 3)) return SIS3D_EINVAL;
    // head up to 16-byte alignment goes with the tail words
    size_t head = ((16 - ((uintptr_t)dst & 15)) & 15);
    if (head > bytes) head = bytes;
    char *p = (char *)dst;
    if (head) {
        hipLaunchKernelGGL(fill32_kernel, dim3(1), dim3(256), 0, st, (uint4 *)nullptr, (uint32_t *)p, pattern, (size_t)0, head / 4);
        p += head;
        bytes -= head;
    }
    const size_t n4 = bytes / 16, ntail = (bytes - n4 * 16) / 4;
    if (n4 || ntail) {
        const size_t want = (n4 + 255) / 256;
        const unsigned blocks = (unsigned)(want < 1 ? 1 : (want > 4096 ? 4096 : want));
        hipLaunchKernelGGL(fill32_kernel, dim3(blocks), dim3(256), 0, st, (uint4 *)p, (uint32_t *)(p + n4 * 16), pattern, n4, ntail);
    }
    return sis3d_check_launch();
}

extern "C" int sis3d_abi_version(void) { return 1; }

extern "C" const char *sis3d_last_hip_error(void) { return g_hip_err; }

extern "C" const char *sis3d_strerror(int code)
{
    switch (code) {
    case SIS3D_OK: return "ok";
    case SIS3D_EINVAL: return "invalid argument or shape";
    case SIS3D_ELAUNCH: return "HIP kernel launch failed";
    case SIS3D_EWORKSPACE: return "workspace too small";
    case SIS3D_EUNSUPPORTED: return "unsupported configuration";
    default: return "unknown sis3d error";
    }
}
