// ipx_sort.hip -- LSD radix sort of 16-byte junction-record items, and the
// exclusive prefix sum used around it.
//
// No reference counterpart: the reference keeps a std::map<JunctionKey,...>
// (junctions.hh:225-246); here the map's ordering (junctions.hh:77-100) is
// produced by sorting the 96-bit keys built in ipx_scan.hip.
//
// Per 8-bit digit: k_hist (per-tile digit counts, bin-major) -> k_binscan (one block
// per bin: exclusive prefix over the tiles + bin total) -> k_scatter (stable multi-split;
// the tile is reordered in LDS so that each digit's run leaves as one contiguous write).
// Which digits are worth a pass is decided by the caller from the OR/AND of all keys
// (digits on which every key agrees move nothing).
//
// Digit bytes: a histogram needs one byte of each 16-byte item.  k_scatter therefore
// also writes, next to every item it moves, the byte the NEXT pass will look at (1 B per
// item), and k_hist reads that byte stream instead of the items: per pass the sort moves
// 1 + 16 + 16 + 1 bytes per item instead of 16 + 16 + 16.
//
// Tried and dropped (round 1): a one-pass-per-digit "onesweep" (fused histogram +
// decoupled look-back on sc1 status words, in ticketed and in persistent form).  It was
// correct but ~2x slower per pass than k_scatter here: look-back chains over tiles that
// run in lockstep cost more than the separate histogram read saves.
#include "ipx_common.h"
#include <cstdlib>

namespace {

constexpr uint32_t SORT_THREADS = 256;
constexpr uint32_t SORT_IPT = 8;                        // items per thread
constexpr uint32_t SORT_TILE = SORT_THREADS * SORT_IPT; // 2048 items = 32 KiB
constexpr uint32_t SCAN_TILE = 4096;

__device__ __forceinline__ uint32_t digit_of(const uint4 &it, const DigitSpec &d) { return ipx_digit_of(it, d); }

// per-tile digit histogram; BYTES: from the digit-byte stream, else from the items.
// A block walks several tiles (the per-tile work is tiny: launch cost would dominate).
// A block counts GROUP consecutive tiles and then writes, per bin, the group's counts side by side
// (hist is bin-major: with single tiles a block writes 256 lone 4-byte words).  The group's loads are
// issued together; small inputs use GROUP = 1 to have enough blocks.
template <bool BYTES, uint32_t GROUP, uint32_t TILE>
__global__ __launch_bounds__(SORT_THREADS) void k_hist(const uint4 *__restrict__ items,
                                                       const uint8_t *__restrict__ digs, uint64_t n,
                                                       DigitSpec d, uint32_t *__restrict__ hist,
                                                       uint32_t n_tiles) {
    constexpr uint32_t BPT = TILE / SORT_THREADS, WPT = BPT / 4;      // digit bytes (words) per thread and tile
    __shared__ uint32_t hs[GROUP][256];
    const uint32_t n_groups = (n_tiles + GROUP - 1) / GROUP;
    for (uint32_t it = blockIdx.x; it < n_groups; it += gridDim.x) {
        // a grid of exactly n_groups blocks gives every XCD (blockIdx mod 8) one contiguous range of groups, as
        // k_scatter does: the 4-byte counts of neighbouring tiles share the lines of a bin's row
        uint32_t group = it;
        if (gridDim.x == n_groups && n_groups >= 256) {
            const uint32_t x = it & 7u, k = it >> 3, q = n_groups >> 3, r = n_groups & 7u;
            group = x * q + (x < r ? x : r) + k;
        }
        const uint32_t tile0 = group * GROUP;
        // (the digit bytes are asked for BEFORE the counters are cleared and the block meets at the barrier: the kernel is
        // a chain of latencies -- 5 us a launch on S, eight launches a sort -- and this link overlaps for free)
        uint32_t v[BYTES ? GROUP : 1][BYTES ? WPT : 1];
        bool whole[BYTES ? GROUP : 1];
        if (BYTES) {
#pragma unroll
            for (uint32_t g = 0; g < GROUP; ++g) {
                const uint64_t i0 = (uint64_t)(tile0 + g) * TILE + (uint64_t)threadIdx.x * BPT;
                whole[g] = tile0 + g < n_tiles && i0 + BPT <= n;
                // tile bases are TILE-aligned: one 8- or 16-byte load
                if (WPT == 2) {
                    const uint2 x = whole[g] ? *reinterpret_cast<const uint2 *>(digs + i0) : make_uint2(0, 0);
                    v[g][0] = x.x; v[g][1] = x.y;
                } else {
                    const uint4 x = whole[g] ? *reinterpret_cast<const uint4 *>(digs + i0) : make_uint4(0, 0, 0, 0);
                    v[g][0] = x.x; v[g][1] = x.y; v[g][WPT - 2] = x.z; v[g][WPT - 1] = x.w;
                }
            }
        }
#pragma unroll
        for (uint32_t g = 0; g < GROUP; ++g) hs[g][threadIdx.x] = 0;
        __syncthreads();
        if (BYTES) {
#pragma unroll
            for (uint32_t g = 0; g < GROUP; ++g) {
                uint32_t *h = hs[g];
                if (whole[g]) {
                    // high digits of (nearly) sorted keys: the thread's bytes, often the whole wave's,
                    // are one value -> one LDS add instead of BPT (64 * BPT) on a single, contended bin
                    const uint32_t b0 = v[g][0] & 0xffu;
                    bool uni = true;
#pragma unroll
                    for (uint32_t w = 0; w < WPT; ++w) uni = uni && (v[g][w] == b0 * 0x01010101u);
                    const uint64_t act = ipx_ballot(1);
                    const uint32_t first = __builtin_amdgcn_readfirstlane(b0);
                    if (ipx_ballot(uni && b0 == first) == act) {
                        if ((threadIdx.x & 63u) == (uint32_t)(__ffsll((unsigned long long)act) - 1))
                            atomicAdd(&h[first], BPT * (uint32_t)__popcll(act));
                    } else if (uni) {
                        atomicAdd(&h[b0], BPT);
                    } else {
#pragma unroll
                        for (uint32_t w = 0; w < WPT; ++w)
#pragma unroll
                            for (uint32_t k = 0; k < 4; ++k) atomicAdd(&h[(v[g][w] >> (8 * k)) & 0xffu], 1u);
                    }
                } else if (tile0 + g < n_tiles) {
                    const uint64_t i0 = (uint64_t)(tile0 + g) * TILE + (uint64_t)threadIdx.x * BPT;
                    for (uint64_t i = i0; i < n && i < i0 + BPT; ++i) atomicAdd(&h[digs[i]], 1u);
                }
            }
        } else {
            for (uint32_t g = 0; g < GROUP && tile0 + g < n_tiles; ++g) {
                const uint64_t base = (uint64_t)(tile0 + g) * TILE;
#pragma unroll
                for (uint32_t r = 0; r < BPT; ++r) {
                    const uint64_t i = base + r * SORT_THREADS + threadIdx.x;
                    if (i < n) atomicAdd(&hs[g][digit_of(items[i], d)], 1u);
                }
            }
        }
        __syncthreads();
        uint32_t *row = hist + (uint64_t)threadIdx.x * n_tiles + tile0;
#pragma unroll
        for (uint32_t g = 0; g < GROUP; ++g)
            if (tile0 + g < n_tiles) row[g] = hs[g][threadIdx.x];
        __syncthreads();
    }
}

// inclusive prefix sum over the 64 lanes: DPP row shifts and row broadcasts, six v_add_u32_dpp (the __shfl_up form is six
// dependent ds_bpermute round trips, ~100 cycles each)
__device__ __forceinline__ uint32_t sort_wave_incl_scan(uint32_t x) {
    uint32_t v = x;
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);    // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);    // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);    // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);    // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
    return v;
}

// block b: hist[b][0..n_tiles) -> exclusive prefix in place, bin_total[b] = sum.  A thread owns `e` consecutive entries
// of a round (e = 4, 8, 12 or 16: a round covers up to 4096 tiles, i.e. 16 M records), all of them loaded before the
// first is used, and the wave's prefix is a DPP scan: ONE round for every table up to 16 M records where rounds of 1024
// entries with six ds_bpermute trips each took 4.5 us at 1131 tiles and 9 us at 2405 (round 5; eight launches a sort)
__global__ __launch_bounds__(256) void k_binscan(uint32_t *__restrict__ hist, uint32_t n_tiles,
                                                 uint32_t *__restrict__ bin_total) {
    constexpr uint32_t E = 16;
    __shared__ uint32_t ws[2][4];
    uint32_t *row = hist + (uint64_t)blockIdx.x * n_tiles;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t carry = 0, par = 0;
    for (uint32_t base = 0; base < n_tiles; base += 256 * E, par ^= 1u) {
        const uint32_t left = n_tiles - base;
        const uint32_t e = left >= 256 * E ? E : (((left + 255u) / 256u + 3u) & ~3u);      // block-uniform
        const uint32_t i = base + threadIdx.x * e;
        uint32_t v[E];
#pragma unroll
        for (uint32_t k = 0; k < E; ++k) v[k] = (k < e && i + k < n_tiles) ? row[i + k] : 0u;
        uint32_t mine = 0;
#pragma unroll
        for (uint32_t k = 0; k < E; ++k) mine += v[k];
        const uint32_t incl = sort_wave_incl_scan(mine);
        if (lane == 63) ws[par][wave] = incl;
        __syncthreads();                       // (one barrier a round: the rounds alternate between two sets of wave sums)
        uint32_t off = carry + incl - mine;
        for (uint32_t w = 0; w < wave; ++w) off += ws[par][w];
#pragma unroll
        for (uint32_t k = 0; k < E; ++k) {
            if (k < e && i + k < n_tiles) row[i + k] = off;
            off += v[k];
        }
        carry += ws[par][0] + ws[par][1] + ws[par][2] + ws[par][3];
    }
    if (threadIdx.x == 0) bin_total[blockIdx.x] = carry;
}

// NT threads (NT/64 waves) move one TILE-item tile; a CU holds 128 KiB of tiles either way.  More waves per
// tile mean more waves per CU at the same LDS footprint and fewer ranking rounds per wave (256 threads on
// 2048 items: 16 waves per CU, 81 us per pass on S; 512 threads: 32 waves, 69 us).  Larger tiles mean longer
// runs per digit (TILE/256 items on uniform digits) and fewer rows in the histogram.
template <bool WRITE_NEXT, uint32_t NT, uint32_t TILE>
__global__ __launch_bounds__(NT) void k_scatter(const uint4 *__restrict__ src, uint4 *__restrict__ dst,
                                                uint8_t *__restrict__ dig_dst, uint64_t n, DigitSpec d,
                                                DigitSpec next_d, const uint32_t *__restrict__ scanned,
                                                const uint32_t *__restrict__ bin_total,
                                                uint32_t n_tiles, uint32_t xcd_ranges) {
    constexpr uint32_t NW = NT / 64, IPT = TILE / NT, PER_WAVE = TILE / NW;
    __shared__ uint4 tile[TILE];               // 32 or 64 KiB
    __shared__ uint16_t cnt[NW][256];          // per-wave digit counts (<= PER_WAVE each)
    __shared__ uint32_t tile_excl[256];        // exclusive digit prefix inside the tile
    __shared__ uint32_t gbase[256];            // global offset of (digit, this tile)
    __shared__ uint32_t wsum[4], wsum2[4];

    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    // Workgroups go to the 8 XCDs round-robin by blockIdx.  Give every XCD one contiguous range of tiles:
    // inside a digit's region the runs of neighbouring tiles are neighbours in memory, and a 64-byte sector
    // shared by two runs is merged in L2 only if both were written through the SAME XCD's L2.
    uint32_t tile_id = blockIdx.x;
    if (xcd_ranges) {
        const uint32_t x = blockIdx.x & 7u, k = blockIdx.x >> 3, q = n_tiles >> 3, r = n_tiles & 7u;
        tile_id = x * q + (x < r ? x : r) + k;
    }
    const uint64_t base = (uint64_t)tile_id * TILE;
    const uint32_t valid = (uint32_t)((n - base) < TILE ? (n - base) : TILE);

    // wave w owns tile items [w*PER_WAVE, (w+1)*PER_WAVE) in IPT rounds of 64: item order is
    // (wave, round, lane), which is memory order -> the split is stable.  The loads go out first: the
    // bin bases below (two dependent loads, a scan, a barrier) are computed while they are under way.
    uint4 it[IPT];
#pragma unroll
    for (uint32_t r = 0; r < IPT; ++r) {
        const uint32_t li = wave * PER_WAVE + r * 64 + lane;
        if (li < valid) it[r] = src[base + li];
    }

    for (uint32_t k = t; k < NW * 256 / 2; k += NT) reinterpret_cast<uint32_t *>(&cnt[0][0])[k] = 0;
    // global base of bin t = (items in smaller bins) + (bin-t items in earlier tiles); threads 0..255
    uint32_t my_total = 0, my_before = 0, incl2 = 0;
    if (t < 256) {
        my_total = bin_total[t];
        my_before = scanned[(uint64_t)t * n_tiles + tile_id];
        incl2 = sort_wave_incl_scan(my_total);
        if (lane == 63) wsum2[wave] = incl2;
    }
    __syncthreads();
    if (t < 256) {
        uint32_t off = 0;
        for (uint32_t w = 0; w < wave; ++w) off += wsum2[w];
        gbase[t] = off + incl2 - my_total + my_before;
    }

    uint32_t rk[IPT];                          // rank among equal digits inside the wave's chunk
    uint32_t dg[IPT];
#pragma unroll
    for (uint32_t r = 0; r < IPT; ++r) {
        const uint32_t li = wave * PER_WAVE + r * 64 + lane;
        const bool ok = li < valid;
        const uint32_t dig = ok ? digit_of(it[r], d) : 0xffffffffu;
        dg[r] = dig;
        // lanes with the same digit (invalid lanes match nobody valid)
        uint64_t peers = ipx_ballot(ok);
#pragma unroll
        for (uint32_t bit = 0; bit < 8; ++bit) {
            const uint64_t bm = ipx_ballot((dig >> bit) & 1u);
            peers &= ((dig >> bit) & 1u) ? bm : ~bm;
        }
        uint32_t before = 0;
        // equal-digit lanes below this one: v_mbcnt counts the mask bits below the lane in two instructions
        const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(peers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)peers, 0u));
        if (ok) {
            before = cnt[wave][dig];
            rk[r] = before + below;
        }
        __builtin_amdgcn_wave_barrier();
        if (ok && below == 0) cnt[wave][dig] = (uint16_t)(before + (uint32_t)__popcll(peers));
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();

    // digit totals of the tile and their exclusive prefix (one thread per digit)
    uint32_t tot = 0, incl = 0;
    if (t < 256) {
        uint32_t run = 0;
#pragma unroll
        for (uint32_t w = 0; w < NW; ++w) {
            const uint32_t c = cnt[w][t];
            cnt[w][t] = (uint16_t)run;
            run += c;
        }
        tot = run;
        incl = sort_wave_incl_scan(tot);
        if (lane == 63) wsum[wave] = incl;
    }
    __syncthreads();
    if (t < 256) {
        uint32_t woff = 0;
        for (uint32_t w = 0; w < wave; ++w) woff += wsum[w];
        tile_excl[t] = woff + incl - tot;
    }
    __syncthreads();

    // place every item at its sorted position inside the tile
#pragma unroll
    for (uint32_t r = 0; r < IPT; ++r) {
        const uint32_t dig = dg[r];
        if (dig != 0xffffffffu) tile[tile_excl[dig] + cnt[wave][dig] + rk[r]] = it[r];
    }
    __syncthreads();

    // contiguous runs per digit leave together; the next pass's digit byte goes along
#pragma unroll
    for (uint32_t r = 0; r < IPT; ++r) {
        const uint32_t slot = r * NT + t;
        if (slot < valid) {
            const uint4 v = tile[slot];
            const uint32_t dig = digit_of(v, d);
            const uint64_t pos = (uint64_t)gbase[dig] + (slot - tile_excl[dig]);
            dst[pos] = v;
            if (WRITE_NEXT) dig_dst[pos] = (uint8_t)digit_of(v, next_d);
        }
    }
}

// ---- exclusive scan of uint32 ------------------------------------------------
// reduce-then-scan: k_scan_sums (per 4096-element tile), k_scan_top (one block
// walks the tile sums), k_scan_apply.
__device__ __forceinline__ uint32_t block_excl_scan_256(uint32_t v, uint32_t *total) {
    __shared__ uint32_t ws[4];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t incl = sort_wave_incl_scan(v);
    __syncthreads();               // ws may still be read by a previous call
    if (lane == 63) ws[wave] = incl;
    __syncthreads();
    uint32_t off = 0;
    for (uint32_t w = 0; w < wave; ++w) off += ws[w];
    *total = ws[0] + ws[1] + ws[2] + ws[3];
    return off + incl - v;
}

// sixteen consecutive elements of in[0, n) from `base` on (zeros beyond the end).  All sixteen loads go out before the
// first is waited for: written as `(base + k < n) ? in[base + k] : 0` per element the compiler keeps every load behind
// its own branch and the sixteen round trips to memory follow one another (k_scan_small took 12.9 us for 2 x 8 k
// elements that way, 9.1 us now)
__device__ __forceinline__ void load16(const uint32_t *__restrict__ in, uint64_t base, uint64_t n, uint32_t (&v)[16]) {
    if (base + 16 <= n && (reinterpret_cast<uintptr_t>(in) & 15u) == 0) {
        // (base is a multiple of 16 elements: four 16-byte loads instead of sixteen 4-byte ones; round 5)
        const uint4 *q = reinterpret_cast<const uint4 *>(in + base);
        const uint4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
        v[0] = q0.x; v[1] = q0.y; v[2] = q0.z; v[3] = q0.w; v[4] = q1.x; v[5] = q1.y; v[6] = q1.z; v[7] = q1.w;
        v[8] = q2.x; v[9] = q2.y; v[10] = q2.z; v[11] = q2.w; v[12] = q3.x; v[13] = q3.y; v[14] = q3.z; v[15] = q3.w;
    } else if (base + 16 <= n) {
#pragma unroll
        for (uint32_t k = 0; k < 16; ++k) v[k] = in[base + k];
    } else {
        const uint64_t last = n ? n - 1 : 0;
#pragma unroll
        for (uint32_t k = 0; k < 16; ++k) {
            const uint32_t x = (base < n) ? in[base + k < n ? base + k : last] : 0u;      // (a lane wholly beyond the end loads nothing)
            v[k] = (base + k < n) ? x : 0u;
        }
    }
}

// One launch serves one or two independent arrays (blocks [0, a.m) work on job a, the rest on job b): the
// scans of a step are small and latency-bound, so two arrays in three launches instead of six.
struct ScanJob {
    const uint32_t *in;
    uint32_t *out;
    uint64_t n;
    uint32_t *sums;      // one word per SCAN_TILE elements
    uint32_t *total;     // may be null
    uint32_t m;          // tiles = blocks
};

__global__ __launch_bounds__(256) void k_scan_sums(ScanJob a, ScanJob b) {
    const bool first = blockIdx.x < a.m;
    const ScanJob &j = first ? a : b;
    const uint32_t blk = first ? blockIdx.x : blockIdx.x - a.m;
    const uint64_t base = (uint64_t)blk * SCAN_TILE + (uint64_t)threadIdx.x * 16;
    uint32_t v[16];
    load16(j.in, base, j.n, v);
    uint32_t s = 0;
#pragma unroll
    for (uint32_t k = 0; k < 16; ++k) s += v[k];
    uint32_t tot;
    (void)block_excl_scan_256(s, &tot);
    if (threadIdx.x == 0) j.sums[blk] = tot;
}

// block 0: job a, block 1: job b
__global__ __launch_bounds__(256) void k_scan_top(ScanJob a, ScanJob b) {
    const ScanJob &j = blockIdx.x == 0 ? a : b;
    uint32_t carry = 0;
    for (uint64_t base = 0; base < j.m; base += 256) {
        const uint64_t i = base + threadIdx.x;
        const uint32_t v = (i < j.m) ? j.sums[i] : 0;
        uint32_t tot;
        const uint32_t ex = block_excl_scan_256(v, &tot);
        if (i < j.m) j.sums[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0 && j.total) *j.total = carry;
}

// in == out is allowed (each thread reads its 16 values before it writes them)
__global__ __launch_bounds__(256) void k_scan_apply(ScanJob a, ScanJob b) {
    const bool first = blockIdx.x < a.m;
    const ScanJob &j = first ? a : b;
    const uint32_t blk = first ? blockIdx.x : blockIdx.x - a.m;
    const uint64_t base = (uint64_t)blk * SCAN_TILE + (uint64_t)threadIdx.x * 16;
    uint32_t v[16];
    load16(j.in, base, j.n, v);
    uint32_t s = 0;
#pragma unroll
    for (uint32_t k = 0; k < 16; ++k) s += v[k];
    uint32_t tot;
    uint32_t ex = block_excl_scan_256(s, &tot) + j.sums[blk];
#pragma unroll
    for (uint32_t k = 0; k < 16; ++k) {
        if (base + k < j.n) j.out[base + k] = ex;
        ex += v[k];
    }
}

// Short arrays (the per-chunk head counts of a few million records: a few thousand words) in ONE launch: one block
// of 1024 threads, 16 consecutive elements per thread and array.  Three launches of 3-8 us each for 30 KB of data
// were 7 % of the S step.
constexpr uint32_t SMALL_SCAN_MAX = 16384;
__global__ __launch_bounds__(1024) void k_scan_small(ScanJob a, ScanJob b) {
    __shared__ uint32_t ws[2][16];
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    uint32_t v[2][16], s[2] = {0, 0};
#pragma unroll
    for (uint32_t j = 0; j < 2; ++j) {
        const ScanJob &job = j ? b : a;
        const uint64_t base = (uint64_t)t * 16;
        load16(job.in, base, job.n, v[j]);
#pragma unroll
        for (uint32_t k = 0; k < 16; ++k) s[j] += v[j][k];
    }
    const uint32_t incl[2] = {sort_wave_incl_scan(s[0]), sort_wave_incl_scan(s[1])};
    if (lane == 63) { ws[0][wave] = incl[0]; ws[1][wave] = incl[1]; }
    __syncthreads();
#pragma unroll
    for (uint32_t j = 0; j < 2; ++j) {
        const ScanJob &job = j ? b : a;
        uint32_t off = 0, all = 0;
        for (uint32_t w = 0; w < 16; ++w) {
            const uint32_t x = ws[j][w];
            if (w < wave) off += x;
            all += x;
        }
        uint32_t ex = off + incl[j] - s[j];
        const uint64_t base = (uint64_t)t * 16;
        if (base + 16 <= job.n && (reinterpret_cast<uintptr_t>(job.out) & 15u) == 0) {
            uint32_t e[16];
#pragma unroll
            for (uint32_t k = 0; k < 16; ++k) { e[k] = ex; ex += v[j][k]; }
            uint4 *q = reinterpret_cast<uint4 *>(job.out + base);
            q[0] = make_uint4(e[0], e[1], e[2], e[3]); q[1] = make_uint4(e[4], e[5], e[6], e[7]);
            q[2] = make_uint4(e[8], e[9], e[10], e[11]); q[3] = make_uint4(e[12], e[13], e[14], e[15]);
        } else {
#pragma unroll
            for (uint32_t k = 0; k < 16; ++k) {
                if (base + k < job.n) job.out[base + k] = ex;
                ex += v[j][k];
            }
        }
        if (t == 0 && job.total) *job.total = all;
    }
}

}  // namespace

// Exclusive scans of one or two arrays (n2 == 0: one).  block_sums: ipx_scan_sums_words(n1, n2) words.
size_t ipx_scan_sums_words(uint64_t n1, uint64_t n2) { return (size_t)(ipx_div_up(n1, SCAN_TILE) + ipx_div_up(n2, SCAN_TILE) + 4); }

hipError_t ipx_launch_scan_u32_pair(hipStream_t s, const uint32_t *in1, uint32_t *out1, uint64_t n1, uint32_t *total1,
                                    const uint32_t *in2, uint32_t *out2, uint64_t n2, uint32_t *total2,
                                    uint32_t *block_sums) {
    // an empty array only has its total to report
    if (n1 == 0 && total1) { hipError_t e = hipMemsetAsync(total1, 0, sizeof(uint32_t), s); if (e != hipSuccess) return e; }
    if (n2 == 0 && total2) { hipError_t e = hipMemsetAsync(total2, 0, sizeof(uint32_t), s); if (e != hipSuccess) return e; }
    ScanJob a{in1, out1, n1, block_sums, total1, (uint32_t)ipx_div_up(n1, SCAN_TILE)};
    ScanJob b{in2, out2, n2, block_sums + a.m + 2, total2, (uint32_t)ipx_div_up(n2, SCAN_TILE)};
    if (a.m == 0) { a = b; b.m = 0; b.n = 0; }
    if (a.m == 0) return hipSuccess;
    if (a.n <= SMALL_SCAN_MAX && b.n <= SMALL_SCAN_MAX) {
        if (b.n == 0) { b.in = nullptr; b.out = nullptr; b.total = nullptr; }
        hipLaunchKernelGGL(k_scan_small, dim3(1), dim3(1024), 0, s, a, b);
        ipx_prof_mark(s, IPXK_SCAN_U32);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_scan_sums, dim3(a.m + b.m), dim3(256), 0, s, a, b);
    hipLaunchKernelGGL(k_scan_top, dim3(b.m ? 2 : 1), dim3(256), 0, s, a, b);
    hipLaunchKernelGGL(k_scan_apply, dim3(a.m + b.m), dim3(256), 0, s, a, b);
    ipx_prof_mark(s, IPXK_SCAN_U32);
    return hipGetLastError();
}

hipError_t ipx_launch_scan_u32(hipStream_t s, const uint32_t *in, uint32_t *out, uint64_t n,
                               uint32_t *block_sums, uint32_t *total) {
    return ipx_launch_scan_u32_pair(s, in, out, n, total, nullptr, nullptr, 0, nullptr, block_sums);
}

// workspace: hist[256 * n_tiles] + bin_total[256]
size_t ipx_sort_workspace_bytes(uint64_t n) {
    const uint64_t n_tiles = ipx_div_up(n ? n : 1, SORT_TILE);
    return (size_t)((256 * n_tiles + 256) * sizeof(uint32_t) + 256);
}

// Sorts by the listed digits, least significant first.  The result is in buf_b when
// plan.n is odd, in buf_a otherwise.  dig_a (may be NULL) holds the bytes of digit
// plan.digits[0] for the items in buf_a; dig_b is its ping-pong partner.
// first_bytes_valid: dig_a really holds digit plan.digits[0] of buf_a (else the first
// histogram reads the items).
hipError_t ipx_launch_sort(hipStream_t s, uint4 *buf_a, uint4 *buf_b, uint8_t *dig_a, uint8_t *dig_b,
                           bool first_bytes_valid, uint64_t n, const SortPlan &plan, void *workspace,
                           size_t workspace_bytes, int prof_all, const DigitSpec *after_last) {
    if (n == 0 || plan.n == 0) return hipSuccess;
    if (ipx_sort_workspace_bytes(n) > workspace_bytes) return hipErrorInvalidValue;
    // large inputs (3 M records on: L with 4.05 M gains 1.5 % of its step, hist and bin scan get cheaper): 4096-item
    // tiles moved by 1024 threads (IPX_SORT_TILE=2048 keeps the small tile)
    const char *tile_e = getenv("IPX_SORT_TILE");               // tests force either tile size
    const uint32_t tile_env = tile_e ? (uint32_t)atoi(tile_e) : 0u;
    const bool big = tile_env ? tile_env == 4096 : n >= (3u << 20);
    const uint32_t tile_items = big ? 4096u : SORT_TILE;
    const uint32_t n_tiles = (uint32_t)ipx_div_up(n, tile_items);
    uint32_t *hist = (uint32_t *)workspace;
    uint32_t *bin_total = hist + (uint64_t)256 * n_tiles;
    const bool bytes = dig_a != nullptr && dig_b != nullptr;
    // small tiles: 8 per histogram block (16 KiB of digit bytes) once that leaves >= 256 blocks; large
    // tiles: one (measured on S, per step of 8 passes: 1 tile 0.106 ms, 2 0.114, 4 0.118, 8 0.134)
    const uint32_t group = big ? 1u : 8u;
    const bool grouped = (uint64_t)n_tiles * tile_items >= (4u << 20);
    const uint32_t n_groups = grouped ? (n_tiles + group - 1) / group : n_tiles;
    const uint32_t hist_grid = n_groups <= 8192 ? n_groups : 2048;      // one block per group while that is cheap
    uint4 *src = buf_a, *dst = buf_b;
    uint8_t *dsrc = dig_a, *ddst = dig_b;
    // tiles dealt to the XCDs in contiguous ranges, 512 threads on the 2048-item tile: the measured winners of rounds
    // 1-3 (blockIdx order and 256 threads were IPX_SCATTER_XCD=0 / IPX_SCATTER_NT=256 until round 4; NOTEBOOK.md)
    const uint32_t xcd_ranges = n_tiles >= 64 ? 1u : 0u;
    for (uint32_t k = 0; k < plan.n; ++k) {
        const DigitSpec d = plan.digits[k];
        const bool has_next = k + 1 < plan.n || after_last != nullptr;
        const DigitSpec next_d = k + 1 < plan.n ? plan.digits[k + 1] : (after_last ? *after_last : DigitSpec{0, 8, 0, 0});
        const bool from_bytes = bytes && (k > 0 || first_bytes_valid);
#define IPX_HIST(B, G, T) hipLaunchKernelGGL((k_hist<B, G, T>), dim3(hist_grid), dim3(SORT_THREADS), 0, s, src, dsrc, n, d, hist, n_tiles)
        if (big) {
            if (from_bytes) IPX_HIST(true, 1, 4096);
            else if (grouped) IPX_HIST(false, 4, 4096);
            else IPX_HIST(false, 1, 4096);
        } else {
            if (from_bytes && grouped) IPX_HIST(true, 8, 2048);
            else if (from_bytes) IPX_HIST(true, 1, 2048);
            else if (grouped) IPX_HIST(false, 8, 2048);
            else IPX_HIST(false, 1, 2048);
        }
#undef IPX_HIST
        ipx_prof_mark(s, prof_all >= 0 ? prof_all : IPXK_SORT_HIST);
        hipLaunchKernelGGL(k_binscan, dim3(256), dim3(256), 0, s, hist, n_tiles, bin_total);
        ipx_prof_mark(s, prof_all >= 0 ? prof_all : IPXK_SORT_SCAN);
        ipx_prof_pre(s, prof_all >= 0 ? prof_all : IPXK_SORT_SCATTER);
#define IPX_SCATTER(W, NT, T) hipLaunchKernelGGL((k_scatter<W, NT, T>), dim3(n_tiles), dim3(NT), 0, s, src, dst, ddst, n, d, next_d, \
                                                hist, bin_total, n_tiles, xcd_ranges)
        const bool w = bytes && has_next;
        if (big) {
            if (w) IPX_SCATTER(true, 1024, 4096); else IPX_SCATTER(false, 1024, 4096);
        } else {
            if (w) IPX_SCATTER(true, 512, 2048); else IPX_SCATTER(false, 512, 2048);
        }
#undef IPX_SCATTER
        ipx_prof_mark(s, prof_all >= 0 ? prof_all : IPXK_SORT_SCATTER);
        uint4 *ti = src; src = dst; dst = ti;
        uint8_t *tb = dsrc; dsrc = ddst; ddst = tb;
    }
    return hipGetLastError();
}
