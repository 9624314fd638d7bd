// ipx_sort.hip -- LSD radix sort of 16-byte junction-record items, and the
// exclusive prefix sum it is built on.
//
// No reference counterpart: the reference keeps a std::map<JunctionKey,...>
// (junctions.hh:225-246); here the map's ordering (junctions.hh:77-100) is
// produced by sorting the 96-bit keys built in ipx_scan.hip.
//
// Per 8-bit digit: k_hist (per-tile digit counts, bin-major) -> exclusive scan
// -> k_skip (a digit on which every key agrees moves nothing: the pass is
// dropped on the device, no host round trip) -> k_scatter (stable multi-split,
// tile reordered in LDS so that each digit's run leaves as one contiguous write).
#include "ipx_common.h"

namespace {

constexpr uint32_t SORT_THREADS = 256;
constexpr uint32_t SORT_IPT = 8;                        // items per thread
constexpr uint32_t SORT_TILE = SORT_THREADS * SORT_IPT; // 2048 items = 32 KiB
constexpr uint32_t SCAN_TILE = 4096;

__device__ __forceinline__ uint32_t digit_of(const uint4 &it, uint32_t d) {
    const uint32_t w = (d >> 2) == 0 ? it.x : ((d >> 2) == 1 ? it.y : it.z);
    return (w >> ((d & 3u) * 8u)) & 0xffu;
}

// which ping-pong buffer holds the data before pass d: parity of performed passes
__device__ __forceinline__ uint32_t parity_before(const uint32_t *pass_skip, uint32_t d) {
    uint32_t par = 0;
    for (uint32_t k = 0; k < d; ++k) par ^= (pass_skip[k] ? 0u : 1u);
    return par;
}

__global__ __launch_bounds__(SORT_THREADS) void k_hist(const uint4 *__restrict__ a,
                                                       const uint4 *__restrict__ b, uint64_t n,
                                                       uint32_t d, const uint32_t *pass_skip,
                                                       uint32_t *__restrict__ hist, uint32_t n_tiles) {
    __shared__ uint32_t h[256];
    const uint4 *src = parity_before(pass_skip, d) ? b : a;
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t base = (uint64_t)blockIdx.x * SORT_TILE;
#pragma unroll
    for (uint32_t r = 0; r < SORT_IPT; ++r) {
        const uint64_t i = base + r * SORT_THREADS + threadIdx.x;
        if (i < n) atomicAdd(&h[digit_of(src[i], d)], 1u);
    }
    __syncthreads();
    hist[(uint64_t)threadIdx.x * n_tiles + blockIdx.x] = h[threadIdx.x];
}

// after the exclusive scan of hist: bin b holds all n items  <=>  pass moves nothing
__global__ void k_skip(const uint32_t *__restrict__ scanned, uint32_t n_tiles, uint64_t n,
                       uint32_t d, uint32_t *pass_skip) {
    __shared__ uint32_t any;
    if (threadIdx.x == 0) any = 0;
    __syncthreads();
    const uint32_t bin = threadIdx.x;
    const uint64_t lo = scanned[(uint64_t)bin * n_tiles];
    const uint64_t hi = (bin == 255) ? n : scanned[(uint64_t)(bin + 1) * n_tiles];
    if (hi - lo == n) any = 1;
    __syncthreads();
    if (threadIdx.x == 0) pass_skip[d] = any;
}

__global__ __launch_bounds__(SORT_THREADS) void k_scatter(const uint4 *a, const uint4 *b, uint64_t n,
                                                          uint32_t d, const uint32_t *pass_skip,
                                                          const uint32_t *__restrict__ scanned,
                                                          uint32_t n_tiles) {
    if (pass_skip[d]) return;
    const uint32_t par = parity_before(pass_skip, d);
    const uint4 *__restrict__ src = par ? b : a;
    uint4 *__restrict__ dst = par ? const_cast<uint4 *>(a) : const_cast<uint4 *>(b);

    __shared__ uint4 tile[SORT_TILE];          // 32 KiB
    __shared__ uint32_t cnt[4][256];           // per-wave digit counts
    __shared__ uint32_t tile_excl[256];        // exclusive digit prefix inside the tile
    __shared__ uint32_t gbase[256];            // global offset of (digit, this tile)

    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const uint64_t lane_lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const uint64_t base = (uint64_t)blockIdx.x * SORT_TILE;
    const uint32_t valid = (uint32_t)((n - base) < SORT_TILE ? (n - base) : SORT_TILE);

    for (uint32_t k = t; k < 4 * 256; k += SORT_THREADS) (&cnt[0][0])[k] = 0;
    gbase[t] = scanned[(uint64_t)t * n_tiles + blockIdx.x];
    __syncthreads();

    // wave w owns tile items [w*512, w*512+512) in 8 rounds of 64: item order is
    // (wave, round, lane), which is memory order -> the split is stable.
    uint4 it[SORT_IPT];
    uint32_t rk[SORT_IPT];                     // rank among equal digits inside the wave's chunk
    uint32_t dg[SORT_IPT];
#pragma unroll
    for (uint32_t r = 0; r < SORT_IPT; ++r) {
        const uint32_t li = wave * 512 + r * 64 + lane;
        const bool ok = li < valid;
        if (ok) it[r] = src[base + li];
        const uint32_t dig = ok ? digit_of(it[r], d) : 0xffffffffu;
        dg[r] = dig;
        // lanes with the same digit (invalid lanes match nobody valid)
        uint64_t peers = __ballot(ok);
#pragma unroll
        for (uint32_t bit = 0; bit < 8; ++bit) {
            const uint64_t bm = __ballot((dig >> bit) & 1u);
            peers &= ((dig >> bit) & 1u) ? bm : ~bm;
        }
        uint32_t before = 0;
        if (ok) {
            before = cnt[wave][dig];
            rk[r] = before + (uint32_t)__popcll(peers & lane_lt);
        }
        __builtin_amdgcn_wave_barrier();
        if (ok && (peers & lane_lt) == 0) cnt[wave][dig] = before + (uint32_t)__popcll(peers);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();

    // digit totals of the tile and their exclusive prefix (one thread per digit)
    {
        const uint32_t c0 = cnt[0][t], c1 = cnt[1][t], c2 = cnt[2][t], c3 = cnt[3][t];
        const uint32_t tot = c0 + c1 + c2 + c3;
        // turn per-wave counts into per-wave exclusive offsets
        cnt[0][t] = 0; cnt[1][t] = c0; cnt[2][t] = c0 + c1; cnt[3][t] = c0 + c1 + c2;
        // block exclusive scan of tot over 256 digits
        uint32_t incl = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t v = __shfl_up(incl, o);
            if ((int)lane >= o) incl += v;
        }
        __shared__ uint32_t wsum[4];
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t woff = 0;
        for (uint32_t w = 0; w < wave; ++w) woff += wsum[w];
        tile_excl[t] = woff + incl - tot;
    }
    __syncthreads();

    // place every item at its sorted position inside the tile
#pragma unroll
    for (uint32_t r = 0; r < SORT_IPT; ++r) {
        const uint32_t dig = dg[r];
        if (dig != 0xffffffffu) tile[tile_excl[dig] + cnt[wave][dig] + rk[r]] = it[r];
    }
    __syncthreads();

    // contiguous runs per digit leave together
#pragma unroll
    for (uint32_t r = 0; r < SORT_IPT; ++r) {
        const uint32_t slot = r * SORT_THREADS + t;
        if (slot < valid) {
            const uint4 v = tile[slot];
            const uint32_t dig = digit_of(v, d);
            dst[(uint64_t)gbase[dig] + (slot - tile_excl[dig])] = v;
        }
    }
}

// ---- exclusive scan of uint32 ------------------------------------------------
// reduce-then-scan: k_scan_sums (per 4096-element tile), k_scan_top (one block
// walks the tile sums), k_scan_apply.
__device__ __forceinline__ uint32_t block_excl_scan_256(uint32_t v, uint32_t *total) {
    __shared__ uint32_t ws[4];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t x = __shfl_up(incl, o);
        if ((int)lane >= o) incl += x;
    }
    __syncthreads();               // ws may still be read by a previous call
    if (lane == 63) ws[wave] = incl;
    __syncthreads();
    uint32_t off = 0;
    for (uint32_t w = 0; w < wave; ++w) off += ws[w];
    *total = ws[0] + ws[1] + ws[2] + ws[3];
    return off + incl - v;
}

__global__ __launch_bounds__(256) void k_scan_sums(const uint32_t *__restrict__ in, uint64_t n,
                                                   uint32_t *__restrict__ sums) {
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * 16;
    uint32_t s = 0;
#pragma unroll
    for (uint32_t k = 0; k < 16; ++k)
        if (base + k < n) s += in[base + k];
    uint32_t tot;
    (void)block_excl_scan_256(s, &tot);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void k_scan_top(uint32_t *sums, uint64_t m, uint32_t *total) {
    uint32_t carry = 0;
    for (uint64_t base = 0; base < m; base += 256) {
        const uint64_t i = base + threadIdx.x;
        const uint32_t v = (i < m) ? sums[i] : 0;
        uint32_t tot;
        const uint32_t ex = block_excl_scan_256(v, &tot);
        if (i < m) sums[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0 && total) *total = carry;
}

// in == out is allowed (each thread reads its 16 values before it writes them)
__global__ __launch_bounds__(256) void k_scan_apply(const uint32_t *in, uint32_t *out, uint64_t n,
                                                    const uint32_t *__restrict__ sums) {
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * 16;
    uint32_t v[16];
    uint32_t s = 0;
#pragma unroll
    for (uint32_t k = 0; k < 16; ++k) {
        v[k] = (base + k < n) ? in[base + k] : 0;
        s += v[k];
    }
    uint32_t tot;
    uint32_t ex = block_excl_scan_256(s, &tot) + sums[blockIdx.x];
#pragma unroll
    for (uint32_t k = 0; k < 16; ++k) {
        if (base + k < n) out[base + k] = ex;
        ex += v[k];
    }
}

}  // namespace

static hipError_t scan_u32(hipStream_t s, const uint32_t *in, uint32_t *out, uint64_t n,
                           uint32_t *block_sums, uint32_t *total, int prof_id) {
    if (n == 0) {
        if (total) return hipMemsetAsync(total, 0, sizeof(uint32_t), s);
        return hipSuccess;
    }
    const uint64_t m = ipx_div_up(n, SCAN_TILE);
    hipLaunchKernelGGL(k_scan_sums, dim3((uint32_t)m), dim3(256), 0, s, in, n, block_sums);
    hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(256), 0, s, block_sums, m, total);
    hipLaunchKernelGGL(k_scan_apply, dim3((uint32_t)m), dim3(256), 0, s, in, out, n, block_sums);
    ipx_prof_mark(s, prof_id);
    return hipGetLastError();
}

hipError_t ipx_launch_scan_u32(hipStream_t s, const uint32_t *in, uint32_t *out, uint64_t n,
                               uint32_t *block_sums, uint32_t *total) {
    return scan_u32(s, in, out, n, block_sums, total, IPXK_SCAN_U32);
}

// workspace: hist[256 * n_tiles] + scan block sums
size_t ipx_sort_workspace_bytes(uint64_t n) {
    const uint64_t n_tiles = ipx_div_up(n ? n : 1, SORT_TILE);
    const uint64_t hist = 256 * n_tiles;
    const uint64_t sums = ipx_div_up(hist, SCAN_TILE) + 1;
    return (size_t)((hist + sums) * sizeof(uint32_t) + 256);
}

hipError_t ipx_launch_sort(hipStream_t s, uint4 *buf_a, uint4 *buf_b, uint64_t n, uint32_t n_digits,
                           void *workspace, size_t workspace_bytes, uint32_t *pass_skip) {
    hipError_t e = hipMemsetAsync(pass_skip, 0, sizeof(uint32_t) * 16, s);
    if (e != hipSuccess) return e;
    if (n == 0) return hipSuccess;
    const uint32_t n_tiles = (uint32_t)ipx_div_up(n, SORT_TILE);
    uint32_t *hist = (uint32_t *)workspace;
    uint32_t *sums = hist + (uint64_t)256 * n_tiles;
    if (ipx_sort_workspace_bytes(n) > workspace_bytes) return hipErrorInvalidValue;
    for (uint32_t d = 0; d < n_digits; ++d) {
        hipLaunchKernelGGL(k_hist, dim3(n_tiles), dim3(SORT_THREADS), 0, s, buf_a, buf_b, n, d,
                           pass_skip, hist, n_tiles);
        ipx_prof_mark(s, IPXK_SORT_HIST);
        e = scan_u32(s, hist, hist, (uint64_t)256 * n_tiles, sums, nullptr, IPXK_SORT_SCAN);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_skip, dim3(1), dim3(256), 0, s, hist, n_tiles, n, d, pass_skip);
        ipx_prof_mark(s, IPXK_SORT_SKIP);
        hipLaunchKernelGGL(k_scatter, dim3(n_tiles), dim3(SORT_THREADS), 0, s, buf_a, buf_b, n, d,
                           pass_skip, hist, n_tiles);
        ipx_prof_mark(s, IPXK_SORT_SCATTER);
    }
    return hipGetLastError();
}
