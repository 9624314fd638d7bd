// Does one LDS returning atomic hand the lanes of a wave that hit the SAME address their old values in lane
// order?  (If so, atomicAdd(&cnt[digit], 1) is a stable rank among equal digits in one instruction; the
// ballot-match ranking of k_scatter costs ~50.)  Also times both rankings.
//   hipcc -O3 --offload-arch=gfx950 lds_atomic_order.hip -o lds_atomic_order && ./lds_atomic_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

template <int MODE>   // 0: check atomics against ballots; 1: time atomics; 2: time ballots
__global__ __launch_bounds__(256) void k(uint32_t iters, uint32_t n_values, unsigned long long *bad, uint32_t *sink) {
    __shared__ uint32_t cnt[4][256];
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    uint32_t acc = 0;
    unsigned long long mism = 0;
    for (uint32_t it = 0; it < iters; ++it) {
        cnt[wave][lane] = 0; cnt[wave][lane + 64] = 0; cnt[wave][lane + 128] = 0; cnt[wave][lane + 192] = 0;
        __builtin_amdgcn_wave_barrier();
        const uint32_t h = hash32((blockIdx.x * 256u + t) * 2654435761u + it * 40503u);
        // mixture: few values (heavy conflicts) or many
        const uint32_t dig = (it & 1u) ? (h % n_values) : (h & 255u);
        uint32_t r_atomic = 0, r_ballot = 0;
        if (MODE != 2) r_atomic = atomicAdd(&cnt[wave][dig], 1u);
        if (MODE != 1) {
            unsigned long long peers = ~0ull;
#pragma unroll
            for (uint32_t b = 0; b < 8; ++b) {
                const unsigned long long bm = __builtin_amdgcn_ballot_w64((dig >> b) & 1u);
                peers &= ((dig >> b) & 1u) ? bm : ~bm;
            }
            r_ballot = __builtin_amdgcn_mbcnt_hi((uint32_t)(peers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)peers, 0u));
        }
        if (MODE == 0 && r_atomic != r_ballot) ++mism;
        acc += r_atomic + r_ballot;
        __builtin_amdgcn_wave_barrier();
    }
    if (mism) atomicAdd(bad, mism);
    if (acc == 0xdeadbeef) sink[0] = acc;
}

int main() {
    unsigned long long *bad; uint32_t *sink;
    CK(hipMalloc(&bad, 8)); CK(hipMalloc(&sink, 4)); CK(hipMemset(bad, 0, 8));
    for (uint32_t nv : {1u, 2u, 3u, 7u, 16u, 64u}) {
        hipLaunchKernelGGL(k<0>, dim3(2048), dim3(256), 0, 0, 4000u, nv, bad, sink);
        CK(hipDeviceSynchronize());
        unsigned long long h = 0; CK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost));
        printf("values %2u: %llu lane-ops checked, mismatches so far %llu\n", nv, 2048ull * 256 * 4000, h);
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 1; mode <= 2; ++mode) {
        for (uint32_t nv : {2u, 64u}) {
            CK(hipEventRecord(e0));
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(2048), dim3(256), 0, 0, 4000u, nv, bad, sink);
            else hipLaunchKernelGGL(k<2>, dim3(2048), dim3(256), 0, 0, 4000u, nv, bad, sink);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%s, %2u values: %.3f ms for %.1f G lane-ranks -> %.1f G/s\n", mode == 1 ? "atomic" : "ballot", nv, ms,
                   2048.0 * 256 * 4000 / 1e9, 2048.0 * 256 * 4000 / 1e6 / ms);
        }
    }
    return 0;
}
