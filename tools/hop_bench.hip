// Micro-benchmark: latency of one data-tagged hand-off hop between workgroups on gfx950.
// A chain of H hops walks round-robin over G workgroups; hop h is taken by workgroup h % G, which polls a
// 32-byte record (two 16-byte granules {x, y, z, seq}) with sc1 loads until both carry seq == h, does
// `work` dependent FMAs, and stores the record with seq = h + 1 through sc1 (write-through) stores.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

__device__ __forceinline__ v4u ld16_sc1(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16);
}
__device__ __forceinline__ void st16_sc1(__amdgpu_buffer_rsrc_t r, unsigned off, v4u v) {
    __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, 16);
}

__global__ void k_chain(unsigned* rec, unsigned nrec, unsigned H, unsigned work, unsigned* fail) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(rec, 0, nrec * 32, 0x00020000);
    const unsigned lane = threadIdx.x;
    const unsigned off = (lane % nrec) * 32;  // every lane walks its own record (same chain structure)
    for (unsigned h = blockIdx.x; h < H; h += gridDim.x) {
        v4u a, b;
        unsigned spins = 0;
        for (;;) {
            a = ld16_sc1(r, off);
            b = ld16_sc1(r, off + 16);
            if (a.w == h && b.w == h) break;
            if (++spins > (1u << 20)) { *fail = 1; return; }
            __builtin_amdgcn_s_sleep(1);
        }
        float x = __uint_as_float(a.x);
        for (unsigned k = 0; k < work; ++k) x = x * 1.0000001f + 1.0f;
        a.x = __float_as_uint(x); a.w = h + 1; b.w = h + 1;
        st16_sc1(r, off, a);
        st16_sc1(r, off + 16, b);
    }
}

int main(int argc, char** argv) {
    unsigned* rec; unsigned* fail;
    hipMalloc(&rec, 64 * 32); hipMalloc(&fail, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const unsigned H = 2000;
    for (unsigned work : {0u, 400u}) for (unsigned lanes : {1u, 64u}) for (unsigned G : {1u, 2u, 8u, 64u, 256u, 1024u}) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(rec, 0, 64 * 32); hipMemset(fail, 0, 4);
            hipEventRecord(e0);
            k_chain<<<G, lanes>>>(rec, lanes, H, work, fail);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        unsigned f; hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost);
        unsigned last[8]; hipMemcpy(last, rec, 32, hipMemcpyDeviceToHost);
        printf("work %4u lanes %2u G %4u: %.3f us/hop  (fail %u, seq %u)\n", work, lanes, G, 1e3f * best / H, f, last[3]);
        fflush(stdout);
    }
    return 0;
}
