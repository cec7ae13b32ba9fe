// Micro-benchmark (companion of hop_bench.hip): does a tagged hand-off between two workgroups cost less when both sit on
// the SAME XCD and the poll may be served by that XCD's L2?  Workgroups go to the XCDs round robin (blockIdx % 8), so a
// chain over G workgroups "of one XCD" is launched as 8 G workgroups of which only those with blockIdx % 8 == 0 take
// part. Store: sc1 (write-through), as the solvers do. Poll: sc1 (agent scope: may hit this XCD's L2) or sc0|sc1 (system
// scope: past the L2, what the solvers use).
//   hipcc --offload-arch=gfx950 -O3 tools/hop_xcd.hip -o /tmp/hop_xcd && /tmp/hop_xcd
#include <hip/hip_runtime.h>
#include <cstdio>

typedef unsigned v4u __attribute__((ext_vector_type(4)));

template <int AUX>
__device__ __forceinline__ v4u ld16(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, AUX);
}
__device__ __forceinline__ void st16_sc1(__amdgpu_buffer_rsrc_t r, unsigned off, v4u v) {
    __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, 16);
}

// stride: 1 = every workgroup takes part (they spread over all XCDs), 8 = only blockIdx % 8 == 0 (one XCD)
template <int AUX>
__global__ void k_chain(unsigned* rec, unsigned H, unsigned stride, unsigned* fail) {
    if (blockIdx.x % stride != 0) return;
    const unsigned me = blockIdx.x / stride, G = gridDim.x / stride;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(rec, 0, 64 * 32, 0x00020000);
    const unsigned off = (threadIdx.x % 64) * 32;
    for (unsigned h = me; h < H; h += G) {
        v4u a, b;
        unsigned spins = 0;
        for (;;) {
            a = ld16<AUX>(r, off);
            b = ld16<AUX>(r, off + 16);
            if (a.w == h && b.w == h) break;
            if (++spins > (1u << 22)) { *fail = 1; return; }
            __builtin_amdgcn_s_sleep(1);
        }
        a.x += 1; a.w = h + 1; b.w = h + 1;
        st16_sc1(r, off, a);
        st16_sc1(r, off + 16, b);
    }
}

int main() {
    unsigned* rec; unsigned* fail;
    hipMalloc(&rec, 64 * 32); hipMalloc(&fail, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const unsigned H = 4000;
    for (int mode = 0; mode < 4; ++mode) for (unsigned G : {2u, 8u, 24u}) {
        const unsigned stride = (mode & 1) ? 8u : 1u;
        const bool bypass = (mode & 2) != 0;
        float best = 1e9f;
        unsigned f = 0;
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(rec, 0, 64 * 32); hipMemset(fail, 0, 4);
            hipEventRecord(e0);
            if (bypass) k_chain<(int)0x80000010><<<G * stride, 64>>>(rec, H, stride, fail);
            else k_chain<16><<<G * stride, 64>>>(rec, H, stride, fail);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            unsigned ff; hipMemcpy(&ff, fail, 4, hipMemcpyDeviceToHost); f |= ff;
        }
        printf("%-9s poll %-7s G %2u: %.3f us/hop%s\n", stride == 8 ? "one XCD" : "all XCDs", bypass ? "sc0|sc1" : "sc1", G,
               1e3f * best / H, f ? "  (TIMED OUT: stale line)" : "");
        fflush(stdout);
    }
    return 0;
}
