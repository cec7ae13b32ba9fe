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

extern "C" int alu_probe_launch(float* out, const float* seed, int blocks, int iters, void* stream) {
    hipLaunchKernelGGL(k_alu, dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream, out, seed, iters);
    return (int)hipGetLastError();
}
