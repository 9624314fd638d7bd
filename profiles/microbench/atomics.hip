// atomics.hip -- how fast are scattered device-scope atomics on MI355X?  (design input for a
// sort-free junction aggregate: one counter cell per (junction, read offset), one row per junction)
//   hipcc --offload-arch=gfx950 -O3 -o atomics atomics.hip && ./atomics
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// mode 0: cell = (i / per + jitter) * R + hash % R  (stream nearly sorted by junction)
// mode 1: cell = hash % cells                        (no locality)
template <int MODE, typename T>
__global__ void k_add(T *cells, uint64_t n_cells, uint32_t n, uint32_t per, uint32_t R, uint32_t J) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t h = mix(i);
    uint64_t c;
    if (MODE == 0) {
        uint32_t j = i / per + (h >> 28);
        if (j >= J) j = J - 1;
        c = (uint64_t)j * R + (mix(h) % R);
    } else {
        c = (((uint64_t)h << 16) ^ mix(h)) % n_cells;
    }
    atomicAdd(&cells[c], (T)1);
}
template <int MODE>
__global__ void k_min64(unsigned long long *rows, uint32_t n, uint32_t per, uint32_t J) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t h = mix(i);
    uint32_t j = MODE == 0 ? i / per + (h >> 28) : h % J;
    if (j >= J) j = J - 1;
    atomicMin(&rows[j], (unsigned long long)i);
}
// same, but one atomic per run of equal rows inside the wave (leader loop)
__global__ void k_min64_wave(unsigned long long *rows, uint32_t n, uint32_t per, uint32_t J) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool ok = i < n;
    const uint32_t h = mix(i);
    uint32_t j = i / per + (h >> 28);
    if (j >= J) j = J - 1;
    unsigned long long v = i;
    uint64_t todo = __builtin_amdgcn_ballot_w64(ok);
    while (todo) {
        const int lead = __ffsll((unsigned long long)todo) - 1;
        const uint32_t jl = __shfl(j, lead);
        const bool mine = ok && j == jl;
        const uint64_t m = __builtin_amdgcn_ballot_w64(mine);
        unsigned long long x = mine ? v : ~0ull;
        for (int d = 32; d; d >>= 1) { const unsigned long long o = __shfl_xor(x, d); x = o < x ? o : x; }
        if ((int)(threadIdx.x & 63) == lead) atomicMin(&rows[jl], x);
        todo &= ~m;
    }
}
// hash probe: 64-bit keys, open addressing, ~all hits after the first touch
__global__ void k_probe(unsigned long long *table, uint32_t mask, uint32_t n, uint32_t per, uint32_t *out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t h = mix(i);
    const uint32_t j = i / per + (h >> 28);
    const unsigned long long key = 0x9e3779b97f4a7c15ull * (j + 1);
    uint32_t s = (uint32_t)(key >> 40) & mask;
    for (;;) {
        unsigned long long cur = table[s];
        if (cur == key) break;
        if (cur == ~0ull) {
            cur = atomicCAS(&table[s], ~0ull, key);
            if (cur == ~0ull || cur == key) break;
        }
        s = (s + 1) & mask;
    }
    out[i] = s;
}

int main() {
    const uint32_t n = 10000000, per = 32, R = 128, J = n / per + 16;
    const uint64_t n_cells = (uint64_t)J * R;
    uint32_t *cells; unsigned long long *rows, *table; uint32_t *out;
    CK(hipMalloc(&cells, n_cells * 4)); CK(hipMalloc(&rows, (uint64_t)J * 8 * 4));
    const uint32_t tmask = (1u << 22) - 1;
    CK(hipMalloc(&table, (uint64_t)(tmask + 1) * 8)); CK(hipMalloc(&out, (uint64_t)n * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const dim3 g((n + 255) / 256), b(256);
    auto run = [&](const char *name, auto launch) {
        float best = 1e9f;
        for (int r = 0; r < 5; ++r) {
            hipMemsetAsync(cells, 0, n_cells * 4, 0); hipMemsetAsync(rows, 0xff, (uint64_t)J * 32, 0);
            hipMemsetAsync(table, 0xff, (uint64_t)(tmask + 1) * 8, 0);
            hipEventRecord(e0, 0); launch(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("%-44s %8.1f us  %6.2f G/s\n", name, best * 1e3, n / best / 1e6);
    };
    run("u32 add, near-sorted cells (J x 128)", [&] { k_add<0, uint32_t><<<g, b>>>(cells, n_cells, n, per, R, J); });
    run("u32 add, random cells (156 MB)", [&] { k_add<1, uint32_t><<<g, b>>>(cells, n_cells, n, per, R, J); });
    run("u32 add, near-sorted, R=1 (row counters)", [&] { k_add<0, uint32_t><<<g, b>>>(cells, n_cells, n, per, 1, J); });
    run("u64 min, near-sorted rows", [&] { k_min64<0><<<g, b>>>(rows, n, per, J); });
    run("u64 min, random rows", [&] { k_min64<1><<<g, b>>>(rows, n, per, J); });
    run("u64 min, near-sorted, wave-aggregated", [&] { k_min64_wave<<<g, b>>>(rows, n, per, J); });
    run("hash insert+probe 64-bit keys (32 MB table)", [&] { k_probe<<<g, b>>>(table, tmask, n, per, out); });
    run("memset 156 MB", [&] { hipMemsetAsync(cells, 0, n_cells * 4, 0); });
    return 0;
}
