// Does the work of a SECOND wave on a SIMD hide behind the first wave's fp32 MFMAs on gfx950?  One workgroup per CU; either 4 waves
// (one per SIMD) that each run NM MFMAs + NX "other" instructions per iteration, or 8 waves (two per SIMD) that each run half of both.
// "Other" = ds_read_b128 (L), s_add_u32 (S), v_pk_add_f32 (V), global_load_lds (G).  Prints ns per iteration per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/issue_overlap.hip -o tools/_bin/issue_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NM, int NL, int NS, int NV, int NG, int NA = 0, int NW = 0, int NLD = 0, int CL = 1>
__global__ __launch_bounds__(512) void k(float *out, const float *src, int iters)
{
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63;
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    float a = lane * 0.5f, b = 1.0f + lane;
    f32x4 l[4] = {};
    f32x2 v0 = {1.f, 2.f}, v1 = {3.f, 4.f};
    f32x2 vv[8] = {};
    unsigned s = 0;
    const float *ldsp = lds + threadIdx.x * 4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            acc[m & 15] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m & 15], 0, 0, 0);
            if (m * NL / NM != (m + 1) * NL / NM) asm volatile("ds_read_b128 %0, %1" : "=v"(l[m & 3]) : "v"((unsigned)(size_t)ldsp & 0xffff));
            if (m * NS / NM != (m + 1) * NS / NM) asm volatile("s_add_u32 %0, %0, 1" : "+s"(s) : : "scc");
            if constexpr (CL == 1) { if (m * NV / NM != (m + 1) * NV / NM) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v0) : "v"(v1)); }
            else if (m * (NV / CL) / NM != (m + 1) * (NV / CL) / NM) {
#pragma unroll
                for (int c = 0; c < CL; ++c) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(vv[c & 7]) : "v"(v1));
            }
            if (m * (NG / CL) / NM != (m + 1) * (NG / CL) / NM)
#pragma unroll
              for (int c = 0; c < (NG ? CL : 0); ++c)
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(src + ((it * 8 + m + c) & 1023) * 256 + lane * 4),
                                                 (void __attribute__((address_space(3))) *)(lds + 8192 + (threadIdx.x >> 6) * 256 + c * 2048), 16, 0, 0);
            if (m * NA / NM != (m + 1) * NA / NM) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v0.x) : "v"(v1.x));
            if (m * NW / NM != (m + 1) * NW / NM) asm volatile("ds_write_b32 %0, %1" : : "v"((unsigned)(size_t)ldsp & 0xffff), "v"(a) : "memory");
            if (m * NLD / NM != (m + 1) * NLD / NM) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(l[m & 3]) : "v"(lane * 16), "s"(src + ((it * 8 + m) & 1023) * 256) : "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float r = s + v0.x + v0.y;
    for (int i = 0; i < 8; ++i) r += vv[i].x;
    for (int i = 0; i < 16; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 4; ++i) r += l[i][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int NM, int NL, int NS, int NV, int NG, int NA = 0, int NW = 0, int NLD = 0, int CL = 1>
void run(const char *what, float *out, const float *src)
{
    const int iters = 2000;
    for (int two = 0; two < 2; ++two) {
        const int thr = two ? 512 : 256, it = two ? iters / 2 : iters;      // two waves per SIMD: each does half of the iterations
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        float best = 1e9;
        for (int r = 0; r < 4; ++r) {
            hipEventRecord(e0);
            hipLaunchKernelGGL((k<NM, NL, NS, NV, NG, NA, NW, NLD, CL>), dim3(256), dim3(thr), 64 * 1024, 0, out, src, it);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipError_t err = hipGetLastError();
            if (err != hipSuccess) { printf("%s: %s\n", what, hipGetErrorString(err)); fflush(stdout); return; }
            if (r && ms < best) best = ms;
        }
        printf("%-44s %s: %7.1f ns per (NM=%d MFMA + others) of one SIMD\n", what, two ? "2 waves/SIMD" : "1 wave /SIMD", best * 1e6 / iters, NM);
        fflush(stdout);
    }
}

int main()
{
    float *out, *src;
    if (hipMalloc(&out, 256 * 512 * 4) != hipSuccess) return 1;
    hipMalloc(&src, 1024 * 256 * 4 + 4096);
    hipMemset(src, 0, 1024 * 256 * 4 + 4096);
    run<64, 0, 0, 0, 0>("64 MFMA", out, src);
    run<64, 0, 0, 48, 0>("64 MFMA + 48 v_pk_add, 1 per gap", out, src);
    run<64, 0, 0, 48, 0, 0, 0, 0, 2>("64 MFMA + 48 v_pk_add, clusters of 2", out, src);
    run<64, 0, 0, 48, 0, 0, 0, 0, 4>("64 MFMA + 48 v_pk_add, clusters of 4", out, src);
    run<64, 0, 0, 48, 0, 0, 0, 0, 8>("64 MFMA + 48 v_pk_add, clusters of 8", out, src);
    run<64, 0, 0, 48, 0, 0, 0, 0, 16>("64 MFMA + 48 v_pk_add, clusters of 16", out, src);
    run<64, 0, 0, 48, 0, 0, 0, 0, 48>("64 MFMA + 48 v_pk_add, one cluster", out, src);
    run<64, 0, 0, 0, 8>("64 MFMA + 8 global_load_lds, 1 per gap", out, src);
    run<64, 0, 0, 0, 8, 0, 0, 0, 4>("64 MFMA + 8 global_load_lds, clusters of 4", out, src);
    run<64, 0, 0, 0, 8, 0, 0, 0, 8>("64 MFMA + 8 global_load_lds, one cluster", out, src);
    return 0;
}
