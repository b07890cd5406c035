// VALU issue rates on gfx950 for ONE wave per SIMD (the epilogue regime of the Winograd kernel): cycles per instruction of
// v_add_f32, v_pk_add_f32 (independent / dependent chains), v_accvgpr_read_b32, v_max_f32.   hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
template <int MODE>
__global__ __launch_bounds__(256, 1) void k(float *out, long long *cyc, int iters)
{
    f32x2 a[8], b = {1.f, 2.f};
    for (int i = 0; i < 8; ++i) a[i] = (f32x2){(float)threadIdx.x + i, 1.f};
    float s[8];
    for (int i = 0; i < 8; ++i) s[i] = threadIdx.x + i;
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {          // 64 independent-ish v_add_f32 (8 chains)
            REP8(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8"
                              : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]), "+v"(s[4]), "+v"(s[5]), "+v"(s[6]), "+v"(s[7]) : "v"(b.x));)
        } else if constexpr (MODE == 1) {   // 64 v_pk_add_f32, 8 chains
            REP8(asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8"
                              : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b));)
        } else if constexpr (MODE == 2) {   // 64 dependent v_add_f32 (one chain)
            REP64(asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[0]) : "v"(b.x));)
        } else if constexpr (MODE == 3) {   // 64 dependent v_pk_add_f32
            REP64(asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[0]) : "v"(b));)
        } else if constexpr (MODE == 4) {   // 64 v_accvgpr_read (8 regs)
            REP8(asm volatile("v_accvgpr_read_b32 %0, a0\n v_accvgpr_read_b32 %1, a1\n v_accvgpr_read_b32 %2, a2\n v_accvgpr_read_b32 %3, a3\n v_accvgpr_read_b32 %4, a4\n v_accvgpr_read_b32 %5, a5\n v_accvgpr_read_b32 %6, a6\n v_accvgpr_read_b32 %7, a7"
                              : "=v"(s[0]), "=v"(s[1]), "=v"(s[2]), "=v"(s[3]), "=v"(s[4]), "=v"(s[5]), "=v"(s[6]), "=v"(s[7]) : : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7");)
        } else if constexpr (MODE == 5) {   // 64 v_max_f32, 8 chains
            REP8(asm volatile("v_max_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n v_max_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_max_f32 %7, %7, %8"
                              : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]), "+v"(s[4]), "+v"(s[5]), "+v"(s[6]), "+v"(s[7]) : "v"(b.x));)
        } else if constexpr (MODE == 6) {   // 64 v_pk_add_f32, 2 chains alternating (dependent distance 2)
            REP8(asm volatile("v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %2\n v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %2\n v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %2\n v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %2"
                              : "+v"(a[0]), "+v"(a[1]) : "v"(b));)
        } else if constexpr (MODE == 7) {   // 64 v_add_f32, 2 chains alternating
            REP8(asm volatile("v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %2\n v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %2\n v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %2\n v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %2"
                              : "+v"(s[0]), "+v"(s[1]) : "v"(b.x));)
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float r = 0;
    for (int i = 0; i < 8; ++i) r += s[i] + a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int MODE> void run(const char *name, float *out, long long *cyc)
{
    const int iters = 200;
    for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, out, cyc, iters); hipDeviceSynchronize(); }
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-44s %.2f cycles per instruction\n", name, (double)c / (iters * 64.0));
}
int main()
{
    float *out; long long *cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
    run<0>("v_add_f32, 8 independent chains", out, cyc);
    run<7>("v_add_f32, 2 alternating chains", out, cyc);
    run<2>("v_add_f32, one dependent chain", out, cyc);
    run<1>("v_pk_add_f32, 8 independent chains", out, cyc);
    run<6>("v_pk_add_f32, 2 alternating chains", out, cyc);
    run<3>("v_pk_add_f32, one dependent chain", out, cyc);
    run<4>("v_accvgpr_read_b32", out, cyc);
    run<5>("v_max_f32, 8 independent chains", out, cyc);
    return 0;
}
