// Calibration micro-benchmark: fp32 MFMA (v_mfma_f32_32x32x2_f32) issue rate on this MI355X as a function of
// independent accumulator chains per wave and waves per SIMD.  hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ void k(float *out, int iters, float seed)
{
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = seed * (threadIdx.x + c + r);
    float a = seed * (threadIdx.x % 7) + 0.5f, b = seed * (threadIdx.x % 5) - 0.25f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0;
    for (int c = 0; c < CHAINS; ++c)
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CHAINS>
void run(int waves_per_simd, float *d)
{
    const int iters = 4096 / CHAINS;
    const int blocks = 256 * waves_per_simd, threads = 256;   // 4 waves per block = 1 per SIMD per block
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<CHAINS>, dim3(blocks), dim3(threads), 0, 0, d, iters, 1e-3f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<CHAINS>, dim3(blocks), dim3(threads), 0, 0, d, iters, 1e-3f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 * iters * 4 * CHAINS * 2.0 * 32 * 32 * 2;
    printf("chains=%d waves/SIMD=%d : %.1f TF (%.3f ms)\n", CHAINS, waves_per_simd, flops / ms / 1e9, ms);
}

int main()
{
    float *d; hipMalloc(&d, 256 * 8 * 256 * 4 * sizeof(float));
    for (int w : {1, 2, 4, 6}) run<1>(w, d);
    for (int w : {1, 2, 4}) run<2>(w, d);
    for (int w : {1, 2}) run<4>(w, d);
    return 0;
}
