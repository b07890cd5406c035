// Do the fp32 matrix pipe (v_mfma_f32_32x32x2_f32) and the fp32 vector pipe (v_pk_fma_f32) of gfx950 run concurrently?
// Both peak at 64 FLOP/clk/SIMD.  512-thread workgroups: waves 0-3 and 4-7 land on SIMDs 0-3 twice, so in mode 2 every
// SIMD hosts one MFMA wave and one packed-FMA wave of each workgroup.  mode 0: all MFMA, mode 1: all v_pk_fma, mode 2: half/half.
// If the pipes are independent, mode 2 finishes in about half the time of modes 0/1 (same total FLOPs).
// hipcc --offload-arch=gfx950 -O3 tools/dual_pipe.hip -o tools/bin/dual_pipe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(512) void k(float *out, int iters, int mode, float seed)
{
    const int wave = threadIdx.x >> 6;
    const bool do_mfma = mode == 0 || (mode == 2 && wave < 4);
    float s = 0.f;
    if (do_mfma) {
        f32x16 acc[2];
        for (int c = 0; c < 2; ++c)
            for (int r = 0; r < 16; ++r) acc[c][r] = seed * (threadIdx.x + c + r);
        const float a = seed * (threadIdx.x % 7) + 0.5f, b = seed * (threadIdx.x % 5) - 0.25f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int c = 0; c < 2; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
        }
        for (int c = 0; c < 2; ++c)
            for (int r = 0; r < 16; ++r) s += acc[c][r];
    } else {
        f32x2 acc[16];
        for (int c = 0; c < 16; ++c) { acc[c].x = seed * (threadIdx.x + c); acc[c].y = seed * (threadIdx.x - c); }
        f32x2 a, b;
        a.x = 1.0f + seed; a.y = 1.0f - seed; b.x = seed * (threadIdx.x % 5); b.y = -seed * (threadIdx.x % 3);
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int c = 0; c < 16; ++c) asm volatile("v_pk_fma_f32 %0, %1, %0, %2" : "+v"(acc[c]) : "v"(a), "v"(b));
        }
        for (int c = 0; c < 16; ++c) s += acc[c].x + acc[c].y;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main()
{
    float *d; hipMalloc(&d, 512 * 512 * sizeof(float));
    const int iters = 8192, blocks = 512;                      // 2 workgroups per CU: 4 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 3; ++mode) {
            hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, d, iters, mode, 1e-3f);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, d, iters, mode, 1e-3f);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            // per wave and iteration: 4 MFMAs x 4096 FLOP = 64 v_pk_fma x 256 FLOP = 16384 FLOP
            const double flops = (double)blocks * 8 * iters * 16384.0;
            printf("mode %d (%s): %.3f ms  %.1f TFLOP/s total\n", mode,
                   mode == 0 ? "all MFMA" : mode == 1 ? "all v_pk_fma_f32" : "half MFMA + half v_pk_fma_f32", ms, flops / ms / 1e9);
        }
    return 0;
}
