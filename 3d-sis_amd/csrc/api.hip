// Library-level entry points (version, error strings).
#include "common.h"
#include <string.h>

static thread_local char g_hip_err[256] = "";

void sis3d_record_hip_error(hipError_t e)
{
    strncpy(g_hip_err, hipGetErrorString(e), sizeof(g_hip_err) - 1);
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
