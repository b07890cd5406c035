// Shared helpers of the 16x16x4 fp32 MFMA kernels (pointwise.hip, mlp16.hip): compile-time loops and the transposed
// tile GEMM  D^T[cout][row] = W[cout][k] * X^T[k][row]  whose result is already in the operand layout of the next GEMM.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

__device__ __forceinline__ f32x4 mfma4(const float4 &w, const float4 &x, f32x4 acc)
{
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, x.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, x.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, x.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, x.w, acc, 0, 0, 0);
    return acc;
}

__device__ __forceinline__ float4 relu4(float4 v, bool on)
{
    if (on) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    return v;
}

// D^T tiles of one GEMM: NT output tiles (16 couts each) x KG groups of 16 input channels; two output tiles are
// accumulated in lockstep so that consecutive MFMAs never wait on the 40-cycle accumulator dependency
template <int NT, int KG>
__device__ __forceinline__ void gemm_t(const float4 (&w)[NT][KG], const float4 (&x)[KG], f32x4 (&acc)[NT])
{
    static_for<0, NT>([&](auto N) { acc[decltype(N)::value] = (f32x4){0.f, 0.f, 0.f, 0.f}; });
    static_for<0, (NT + 1) / 2>([&](auto P) {
        constexpr int n0 = 2 * decltype(P)::value, n1 = n0 + 1;
        static_for<0, KG>([&](auto G) {
            constexpr int g = decltype(G)::value;
            if constexpr (n1 < NT) {
                acc[n0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[n0][g].x, x[g].x, acc[n0], 0, 0, 0);
                acc[n1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[n1][g].x, x[g].x, acc[n1], 0, 0, 0);
                acc[n0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[n0][g].y, x[g].y, acc[n0], 0, 0, 0);
                acc[n1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[n1][g].y, x[g].y, acc[n1], 0, 0, 0);
                acc[n0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[n0][g].z, x[g].z, acc[n0], 0, 0, 0);
                acc[n1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[n1][g].z, x[g].z, acc[n1], 0, 0, 0);
                acc[n0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[n0][g].w, x[g].w, acc[n0], 0, 0, 0);
                acc[n1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[n1][g].w, x[g].w, acc[n1], 0, 0, 0);
            } else {
                acc[n0] = mfma4(w[n0][g], x[g], acc[n0]);
            }
        });
    });
}


} // namespace
