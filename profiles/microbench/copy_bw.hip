// copy_bw.hip -- what a plain streaming kernel reaches on this MI355X: the ceiling k_scatter's
// 16-byte-in / 16-byte-out traffic is measured against (peak HBM3E figure: 8 TB/s).
//   hipcc --offload-arch=gfx950 -O3 -o copy_bw copy_bw.hip && ./copy_bw
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <string>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// grid-stride copy, 16 bytes per lane and iteration
__global__ __launch_bounds__(256) void k_copy(const uint4 *__restrict__ in, uint4 *__restrict__ out, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        out[i] = in[i];
}
// one tile of 2048 items per block (k_scatter's shape), 8 loads in flight per lane
__global__ __launch_bounds__(256) void k_copy_tile(const uint4 *__restrict__ in, uint4 *__restrict__ out, uint64_t n) {
    const uint64_t base = (uint64_t)blockIdx.x * 2048;
    uint4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { const uint64_t i = base + k * 256 + threadIdx.x; if (i < n) v[k] = in[i]; }
#pragma unroll
    for (int k = 0; k < 8; ++k) { const uint64_t i = base + k * 256 + threadIdx.x; if (i < n) out[i] = v[k]; }
}
// the same tile written as 256 runs of 8 items (128 bytes) at permuted places: a scatter's write pattern
__global__ __launch_bounds__(256) void k_copy_runs(const uint4 *__restrict__ in, uint4 *__restrict__ out, uint64_t n,
                                                   uint32_t n_tiles, uint32_t run_items) {
    const uint64_t base = (uint64_t)blockIdx.x * 2048;
    uint4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { const uint64_t i = base + k * 256 + threadIdx.x; v[k] = in[i]; }
    const uint32_t runs_per_tile = 2048 / run_items;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t j = k * 256 + threadIdx.x;          // item in the tile
        const uint32_t run = j / run_items, in_run = j % run_items;
        // run r of tile t lands in the region of "digit" r: region r holds n_tiles runs
        const uint64_t dst = ((uint64_t)run * n_tiles + blockIdx.x) * run_items + in_run;
        if (dst < n) out[dst] = v[k];
    }
    (void)runs_per_tile;
}
__global__ __launch_bounds__(256) void k_read(const uint4 *__restrict__ in, uint32_t *sink, uint64_t n) {
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint4 v = in[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ __launch_bounds__(256) void k_write(uint4 *__restrict__ out, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        out[i] = make_uint4((uint32_t)i, 1, 2, 3);
}

int main() {
    const uint64_t n = 4883ull * 2048;            // 10 M items of 16 bytes = 160 MB, whole tiles
    uint4 *a, *b; uint32_t *sink;
    CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(a, 1, n * 16)); CK(hipMemset(b, 2, n * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char *name, double bytes, auto launch) {
        float best = 1e9f;
        for (int r = 0; r < 8; ++r) {
            (void)hipEventRecord(e0, 0); launch(); (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("%-46s %7.1f us  %6.2f TB/s\n", name, best * 1e3, bytes / best / 1e9);
    };
    const uint32_t tiles = (uint32_t)(n / 2048);
    for (uint32_t blocks : {2048u, 4096u, 8192u, 16384u})
        run(("copy, grid-stride, blocks=" + std::to_string(blocks)).c_str(), 2.0 * n * 16,
            [&] { k_copy<<<blocks, 256>>>(a, b, n); });
    run("copy, one 2048-item tile per block", 2.0 * n * 16, [&] { k_copy_tile<<<tiles, 256>>>(a, b, n); });
    for (uint32_t ri : {64u, 16u, 8u, 4u, 2u, 1u})
        run(("copy, tile -> runs of " + std::to_string(ri * 16) + " B").c_str(), 2.0 * n * 16,
            [&] { k_copy_runs<<<tiles, 256>>>(a, b, n, tiles, ri); });
    run("read only, blocks=8192", 1.0 * n * 16, [&] { k_read<<<8192, 256>>>(a, sink, n); });
    run("write only, blocks=8192", 1.0 * n * 16, [&] { k_write<<<8192, 256>>>(b, n); });
    run("hipMemcpyAsync D2D", 2.0 * n * 16, [&] { (void)hipMemcpyAsync(b, a, n * 16, hipMemcpyDeviceToDevice, 0); });
    return 0;
}
