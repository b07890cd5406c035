// Would the Winograd kernel's input transform read its z pairs as cheaply from a [voxel][4 channels] raw brick (what an LDS-DMA staging of
// the channels-last activations would leave in LDS) as it does from the channel-major brick of today?  One workgroup of 4 waves per CU; per
// iteration a wave issues 64 fp32 MFMAs and 24 LDS reads of one kind between them:
//   A  ds_read_b64  at the shipped layout's lane addresses   (kq CHS + 2 tx PS + 2 ty HZS + 2 tz floats)
//   B  ds_read2_b32 (offsets 0 and 4 dwords) on [slot][4] with slot = (x SY + y) 12 + z   -> lane dword address slot 4 + kq
//   C  no LDS reads (the MFMAs alone)
// and the staging side of the same question, per step and lane (reads of kind A / B included):
//   D  today: 3 global_load_dwordx4 into registers, 12 (v_add address + ds_write_b32) one step later      (+ A's reads)
//   E  LDS-DMA: 3 global_load_lds_dwordx4, nothing else                                                     (+ B's reads)
// CAVEAT (profiles/r06_raw_staging_dma_probe.txt): B looks as cheap as A HERE, where nothing else uses the LDS and nothing waits for the rows;
// in the kernel (16 ds_read_b128 of U per step beside them, the transform waiting for its rows) the same ds_read2_b32 reads made every shape
// 18-35 % slower -- they conflict four-way inside the LDS's 16-lane groups.  A microbenchmark of instruction ISSUE, not of LDS throughput.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ds_read2_probe.hip -o tools/_bin/ds_read2_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters, const float *src)
{
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, li = lane & 15, kq = lane >> 4;
    const int tz = li & 3, ty = (li >> 2) & 1, tx = li >> 3;
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = i * 0.001f;
    __syncthreads();
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    float a = lane * 0.5f, b = 1.0f + lane;
    f32x2 d[8] = {};
    f32x4 g[3] = {};
    unsigned waddr = threadIdx.x * 4;
    unsigned addrA = (unsigned)((kq * 1040 + 2 * tx * 100 + 2 * ty * 16 + 2 * tz) * 4);
    unsigned addrB = (unsigned)(((((2 * tx) * 8 + 2 * ty) * 12 + 2 * tz) * 4 + kq) * 4);          // SY = 8 rows of 12 slots
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 64; ++m) {
            acc[m & 15] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m & 15], 0, 0, 0);
            if (KIND != 2 && (m * 24 / 64 != (m + 1) * 24 / 64)) {
                const int r = m * 24 / 64;             // which of the 24 reads: another (dx, dy, h) of the 4x4x4 patch
                if (KIND == 0 || KIND == 3) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(d[r & 7]) : "v"(addrA), "n"(0));
                else asm volatile("ds_read2_b32 %0, %1 offset0:0 offset1:4" : "=v"(d[r & 7]) : "v"(addrB));
                addrA += 64; addrB += 192;             // next row of the patch (kept inside the 64 KB by the wrap below)
            }
            if (KIND == 3) {
                if (m >= 58 && m < 61) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(g[m - 58]) : "v"(lane * 16), "s"(src + ((it * 4 + m) & 1023) * 256) : "memory");
                if (m >= 8 && m < 56 && (m & 3) == 0) {            // 12 stores of last step's rows, one address add each
                    asm volatile("v_add_u32 %0, %0, %1" : "+v"(waddr) : "v"(4u));
                    asm volatile("ds_write_b32 %0, %1" : : "v"(waddr & 0xfffcu), "v"(g[(m >> 2) % 3][m & 3]) : "memory");
                }
            }
            if (KIND == 4 && m >= 58 && m < 61)
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(src + ((it * 4 + m) & 1023) * 256 + lane * 4),
                                                 (void __attribute__((address_space(3))) *)(lds + 8192 + (threadIdx.x >> 6) * 256 + (m - 58) * 1024), 16, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (KIND == 3) asm volatile("s_waitcnt vmcnt(0)" : "+v"(g[0]), "+v"(g[1]), "+v"(g[2]) : : "memory");
        if (KIND == 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        addrA -= 24 * 64; addrB -= 24 * 192;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float r = 0;
    for (int i = 0; i < 8; ++i) r += d[i].x + d[i].y;
    r += g[0][0] + g[1][1] + g[2][2] + waddr;
    for (int i = 0; i < 16; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int KIND>
void run(const char *what, float *out, const float *src)
{
    const int iters = 4000;
    hipFuncSetAttribute((const void *)k<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0, nullptr);
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256), 65536, nullptr, out, iters, src);
        hipEventRecord(e1, nullptr);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("%-70s %7.1f ns per step (64 MFMAs)\n", what, best * 1e6 / iters);
}

int main()
{
    float *out;
    hipMalloc(&out, 256 * 256 * 4);
    float *src;
    hipMalloc(&src, 1024 * 256 * 4 + 4096);
    hipMemset(src, 0, 1024 * 256 * 4 + 4096);
    run<2>("C: 64 MFMAs alone", out, src);
    run<0>("A: + 24 ds_read_b64, channel-major brick (shipped addresses)", out, src);
    run<1>("B: + 24 ds_read2_b32 (0, +4 dwords), [slot][4 channels] brick", out, src);
    run<3>("D: A + staging of today (3 loads, 12 address adds + ds_write_b32)", out, src);
    run<4>("E: B + staging by LDS-DMA (3 global_load_lds_dwordx4)", out, src);
    run<2>("C: 64 MFMAs alone", out, src);
    return 0;
}
