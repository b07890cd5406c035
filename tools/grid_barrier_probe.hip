// VERDICT r5 item 3, route (b): would ONE persistent kernel with a device-scope barrier between the 25 ENet bottlenecks beat 25
// dependent launches?  The part that decides it is the price of the barrier against the price of a kernel boundary at the ENet's
// geometry (103 workgroups of 256 threads, ~70 KB of fresh data per workgroup and phase).  This probe runs the same phase body
// (read 64 KB written by OTHER workgroups in the previous phase, write 64 KB) 25 times (a) as 25 launches of a captured graph and
// (b) inside one launch with a counter barrier (monotonic counter, relaxed agent-scope polls + s_sleep, release fence before the
// arrive, acquire fence after: the form MI355X_MICROARCH.md prices as "barrier-counter").
//   hipcc --offload-arch=gfx950 -O3 tools/grid_barrier_probe.hip -o tools/_bin/grid_barrier_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
constexpr int PHASES = 25, PER_WG = 16384;                      // floats per workgroup and phase: 64 KB
__device__ __forceinline__ void phase_body(const float *__restrict__ in, float *__restrict__ out, int wg, int nwg, int ph)
{
    const int src = (wg + 1 + ph) % nwg;                        // another workgroup's slab of the previous phase
    const float4 *s = reinterpret_cast<const float4 *>(in + (size_t)src * PER_WG);
    float4 *d = reinterpret_cast<float4 *>(out + (size_t)wg * PER_WG);
    for (int i = threadIdx.x; i < PER_WG / 4; i += blockDim.x) {
        float4 v = s[i];
        v.x += 1.f; v.y += 1.f; v.z += 1.f; v.w += 1.f;
        d[i] = v;
    }
}
__global__ __launch_bounds__(256) void one_phase(const float *in, float *out, int ph) { phase_body(in, out, blockIdx.x, gridDim.x, ph); }
__global__ __launch_bounds__(256) void persistent(float *a, float *b, unsigned *counter)
{
    const int nwg = gridDim.x;
    for (int ph = 0; ph < PHASES; ++ph) {
        phase_body((ph & 1) ? b : a, (ph & 1) ? a : b, blockIdx.x, nwg, ph);
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)nwg * (ph + 1);
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(2);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
}
int main()
{
    for (int nwg : {103, 206, 256}) {
        float *a, *b; unsigned *c;
        hipMalloc(&a, (size_t)nwg * PER_WG * 4); hipMalloc(&b, (size_t)nwg * PER_WG * 4); hipMalloc(&c, 4);
        hipMemset(a, 0, (size_t)nwg * PER_WG * 4);
        hipStream_t st; hipStreamCreate(&st);
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
        for (int ph = 0; ph < PHASES; ++ph) hipLaunchKernelGGL(one_phase, dim3(nwg), dim3(256), 0, st, (ph & 1) ? b : a, (ph & 1) ? a : b, ph);
        hipStreamEndCapture(st, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float best_l = 1e9f, best_p = 1e9f;
        for (int r = 0; r < 8; ++r) {
            hipEventRecord(e0, st);
            for (int i = 0; i < 10; ++i) hipGraphLaunch(ge, st);
            hipEventRecord(e1, st); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms / 10 < best_l) best_l = ms / 10;
            hipMemsetAsync(c, 0, 4, st);
            hipEventRecord(e0, st);
            hipLaunchKernelGGL(persistent, dim3(nwg), dim3(256), 0, st, a, b, c);
            hipEventRecord(e1, st); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1); if (ms < best_p) best_p = ms;
        }
        std::vector<float> h(4); hipMemcpy(h.data(), b, 16, hipMemcpyDeviceToHost);
        printf("%3d workgroups x 25 phases of 64 KB in / 64 KB out: 25 launches (graph) %.1f us = %.2f us per phase; one persistent launch with counter barriers %.1f us = %.2f us per phase   (check %g)\n",
               nwg, best_l * 1e3, best_l * 1e3 / PHASES, best_p * 1e3, best_p * 1e3 / PHASES, h[0]);
        hipFree(a); hipFree(b); hipFree(c);
    }
    return 0;
}
