// alu_probe: is plain per-lane arithmetic (divisions, square roots, LDS slices indexed at run time - the shape of the
// narrow phase) bit-reproducible while MFMA GEMM waves of another stream share the CUs? Pure function of the thread
// index: any difference between two launches is the platform's, not a data race (no inter-lane communication, no
// global reads besides the seed). Built by tools/alu_probe.py.
#include <hip/hip_runtime.h>
#include <cstdint>

constexpr int kThreads = 128, kSlice = 57;

__global__ __launch_bounds__(kThreads) void k_alu(float* __restrict__ out, const float* __restrict__ seed, int iters) {
    __shared__ float lds[kThreads * kSlice];
    float* my = lds + threadIdx.x * kSlice;
    const uint32_t gid = blockIdx.x * kThreads + threadIdx.x;
    const float x0 = seed[gid];
    for (int k = 0; k < 56; ++k) my[k] = x0 + 0.37f * (float)k;
    float acc = 0.0f;
    for (int it = 0; it < iters; ++it) {
        const int j = (it * 7) % 56;
        const float a = my[j], b = my[(j + 13) % 56];
        const float t = a / (fabsf(a - b) + 1.0f);
        const float s = sqrtf(fabsf(t) + 0.5f);
        const float c = a + (b - a) * (t * 0.001f) + s * 1.0e-3f;
        my[(j + 29) % 56] = c;
        acc = acc * 0.999f + c / (s + 1.0f);
    }
    out[gid] = acc;
}

// the same idea with packed fp32 arithmetic (v_pk_mul_f32 / v_pk_add_f32): 2-vectors of products and sums in long
// dependent chains, eight independent chains per lane to keep ~100 VGPRs live like the narrow phase
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(kThreads) void k_alu_pk(float* __restrict__ out, const float* __restrict__ seed, int iters) {
    const uint32_t gid = blockIdx.x * kThreads + threadIdx.x;
    const float x0 = seed[gid];
    f2 v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = f2{x0 + 0.11f * (float)k, x0 - 0.07f * (float)k};
    const f2 a = f2{0.99991f, 1.00007f}, b = f2{1.0e-3f, -2.0e-3f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const f2 w = v[(k + 5) & 15];
            v[k] = v[k] * a + w * b;   // v_pk_mul_f32 x2, v_pk_add_f32 (no contraction)
        }
    }
    f2 acc = f2{0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k < 16; ++k) acc = acc + v[k];
    out[gid] = acc.x + acc.y;
}

extern "C" int alu_probe_launch_pk(float* out, const float* seed, int blocks, int iters, void* stream) {
    hipLaunchKernelGGL(k_alu_pk, dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream, out, seed, iters);
    return (int)hipGetLastError();
}

extern "C" int alu_probe_launch(float* out, const float* seed, int blocks, int iters, void* stream) {
    hipLaunchKernelGGL(k_alu, dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream, out, seed, iters);
    return (int)hipGetLastError();
}
