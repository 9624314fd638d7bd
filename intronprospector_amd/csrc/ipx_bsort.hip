// ipx_bsort.hip -- bucket sort of the 16-byte junction-record items: the sort of ipx_finalize from a few
// ten thousand records on (ipx_sort.hip's LSD radix sort stays for smaller sets, for keys without 16 spare
// bits, for the `ijunc` ranking and as the fallback).
//
// No reference counterpart: the reference keeps a std::map<JunctionKey,...> (junctions.hh:225-246); the order
// produced here is that map's order (junctions.hh:77-100) on the 96-bit keys built in ipx_scan.hip.
//
// Why: an LSD radix sort moves every record once per 8 key bits -- eight passes of 33 bytes per record for
// the 63-65 bits on which S's and L's keys differ: 10.8 GB per 40 M long-read records (35 % of L's step),
// and 24 dependent launches that are all latency on S's 2 M records.  Here a record moves through HBM three
// times whatever the key width:
//   1. SPLITTERS.  k_bs_sample1 (one block) sorts 4096 sampled keys in LDS and keeps k1 - 1 <= 255 coarse
//      splitters; k_bs_sample2 deals a second, larger sample (16 keys per leaf) into the k1 coarse ranges;
//      k_bs_sort2 (one block per coarse range) sorts its share in LDS and keeps up to 255 fine splitters.
//      Keys are compared in a 64-bit "squeezed" form: the top 64 of the bits on which the handle's keys
//      differ at all (OR ^ AND from the scan) -- order-preserving, ties go to the same leaf.
//   2. LEAF IDS.  k_bs_assign finds every record's leaf (two binary searches: coarse table in LDS, fine
//      table in L2) and writes the leaf's number into the 16 spare bits above the key (bits 80..95; keys of
//      more than 80 bits take the LSD sort).  Two ordinary radix passes (k_hist / k_binscan / k_scatter of
//      ipx_sort.hip) on those 16 bits bring every leaf's records together, leaves in key order.
//   3. LEAVES.  k_bs_bounds reads the leaf boundaries off the id bytes the last pass wrote; k_bs_leaf sorts
//      each leaf (1 365 records on average, 4 096 at most) in LDS -- a radix sort over the bits on which THAT
//      leaf's keys differ, items held in registers between passes -- and writes it back with the id removed.
// A leaf that does not fit LDS (sampling luck, or one key with thousands of copies: unsorted input with the
// combiner finding nothing) raises *overflow; the caller then sorts the same buffer with the LSD sort.  A
// leaf of any size whose keys are all equal needs no sorting and is accepted.
#include "ipx_common.h"
#include <algorithm>
#include <cstdlib>

namespace {

constexpr uint32_t BS_NT = 1024, BS_CAP = 4096, BS_NW = BS_NT / 64, BS_IPT = BS_CAP / BS_NT;
constexpr uint32_t BS_SENT = 0xffffffffu;
constexpr uint32_t BS_LISTCAP = 4096;          // second-level sample keys per coarse range (one block sorts them)
constexpr uint32_t BS_OVERSAMPLE = 16;         // second-level sample keys per leaf
constexpr uint32_t BS_LEAF_AVG = BS_CAP / 3;   // records per leaf aimed at
constexpr uint32_t BS_MAX_LEAVES = 43000;      // 256 coarse ranges x <= 256 leaves, with room for uneven ranges
constexpr uint32_t BS_ID_LO = 80;              // the leaf id sits in key bits 80..95 (item.z >> 16)

// the top 64 varying key bits, as up to 6 runs of bits
struct SqueezeSpec {
    uint32_t n;
    uint32_t lo[6], w[6], sh[6];
};

// bits [lo, lo + w) of the 96-bit key, w <= 64
__device__ __forceinline__ uint64_t ext96(const uint4 &it, uint32_t lo, uint32_t w) {
    const uint64_t a = (uint64_t)it.x | ((uint64_t)it.y << 32), b = (uint64_t)it.y | ((uint64_t)it.z << 32);
    uint64_t v;
    if (lo < 32) {
        v = a >> lo;
        if (lo) v |= (uint64_t)it.z << (64 - lo);
    } else {
        v = b >> (lo - 32);
    }
    return w >= 64 ? v : (v & ((1ull << w) - 1ull));
}
__device__ __forceinline__ uint64_t squeeze(const uint4 &it, const SqueezeSpec &sq) {
    uint64_t k = 0;
#pragma unroll
    for (uint32_t r = 0; r < 6; ++r)
        if (r < sq.n) k |= ext96(it, sq.lo[r], sq.w[r]) << sq.sh[r];
    return k;
}

struct BsShared {
    uint4 tile[BS_CAP];                  // 64 KiB
    uint16_t cnt[BS_NW][256];            // per-wave digit counts of a pass
    uint32_t tile_excl[256];
    uint32_t wsum[4];
    uint32_t vary[6];                    // OR / AND of the block's keys
    uint32_t rng[2];
    uint32_t n_passes;
    SortPlan plan;
};

// position of a thread's r-th item in the block's item order (wave, round, lane); rounds = rounds in use
__device__ __forceinline__ uint32_t bs_pos(uint32_t wave, uint32_t lane, uint32_t r, uint32_t rounds) {
    return (wave * rounds + r) * 64u + lane;
}
__device__ __forceinline__ uint32_t bs_rounds(uint32_t n) { return (n + BS_NT - 1) / BS_NT; }

// Sorts the block's n <= BS_CAP items (it[r] = the item at bs_pos(wave, lane, r, rounds) where that is < n) by
// the key bits below max_bits on which they differ; the sorted items are left in sh.tile[0, n) (visible to all
// threads on return).  Ranking as in k_scatter (ipx_sort.hip): per-wave ballots, per-wave digit counts, one scan
// over the 256 digits, items placed through LDS and taken back into registers for the next pass.
__device__ __forceinline__ void bs_block_sort(BsShared &sh, uint4 (&it)[BS_IPT], uint32_t n, uint32_t max_bits) {
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const uint32_t rounds = bs_rounds(n);
    if (t < 6) sh.vary[t] = t < 3 ? 0u : ~0u;
    __syncthreads();
    {
        uint32_t v[6] = {0u, 0u, 0u, ~0u, ~0u, ~0u};
#pragma unroll
        for (uint32_t r = 0; r < BS_IPT; ++r)
            if (r < rounds && bs_pos(wave, lane, r, rounds) < n) {
                v[0] |= it[r].x; v[1] |= it[r].y; v[2] |= it[r].z;
                v[3] &= it[r].x; v[4] &= it[r].y; v[5] &= it[r].z;
            }
#pragma unroll
        for (int k = 0; k < 6; ++k) {
#pragma unroll
            for (int dlt = 32; dlt >= 1; dlt >>= 1) {
                const uint32_t o = __shfl_xor(v[k], dlt);
                v[k] = (k < 3) ? (v[k] | o) : (v[k] & o);
            }
        }
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (v[k]) atomicOr(&sh.vary[k], v[k]);
                if (v[3 + k] != ~0u) atomicAnd(&sh.vary[3 + k], v[3 + k]);
            }
        }
    }
    __syncthreads();
    if (t == 0) {
        const uint32_t vary[3] = {sh.vary[0] ^ sh.vary[3], sh.vary[1] ^ sh.vary[4], sh.vary[2] ^ sh.vary[5]};
        if (n <= 1 || (vary[0] | vary[1] | vary[2]) == 0u) {
            sh.n_passes = 0;
        } else {
            const SortPlan pl = ipx_plan_sort(vary, max_bits, nullptr);
            sh.plan = pl;
            sh.n_passes = pl.n;
        }
    }
    __syncthreads();
    const uint32_t n_passes = sh.n_passes;
    if (n_passes == 0) {                                   // all keys equal (or none): any order is sorted
#pragma unroll
        for (uint32_t r = 0; r < BS_IPT; ++r) {
            const uint32_t p = bs_pos(wave, lane, r, rounds);
            if (r < rounds && p < n) sh.tile[p] = it[r];
        }
        __syncthreads();
        return;
    }
#pragma unroll 1
    for (uint32_t k = 0; k < n_passes; ++k) {
        DigitSpec d;
        d.lo1 = __builtin_amdgcn_readfirstlane(sh.plan.digits[k].lo1);
        d.w1 = __builtin_amdgcn_readfirstlane(sh.plan.digits[k].w1);
        d.lo2 = __builtin_amdgcn_readfirstlane(sh.plan.digits[k].lo2);
        d.w2 = __builtin_amdgcn_readfirstlane(sh.plan.digits[k].w2);
        {
            uint32_t *c32 = reinterpret_cast<uint32_t *>(&sh.cnt[wave][0]);
            c32[lane] = 0;
            c32[lane + 64] = 0;
        }
        __builtin_amdgcn_wave_barrier();
        uint32_t rk[BS_IPT], dg[BS_IPT];
#pragma unroll
        for (uint32_t r = 0; r < BS_IPT; ++r) {
            dg[r] = 0xffffffffu;
            rk[r] = 0;
            if (r < rounds) {
                const bool ok = bs_pos(wave, lane, r, rounds) < n;
                const uint32_t dig = ok ? ipx_digit_of(it[r], d) : 0xffffffffu;
                dg[r] = dig;
                uint64_t peers = ipx_ballot(ok);
#pragma unroll
                for (uint32_t bit = 0; bit < 8; ++bit) {
                    const uint64_t bm = ipx_ballot((dig >> bit) & 1u);
                    peers &= ((dig >> bit) & 1u) ? bm : ~bm;
                }
                const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(peers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)peers, 0u));
                uint32_t before = 0;
                if (ok) {
                    before = sh.cnt[wave][dig];
                    rk[r] = before + below;
                }
                __builtin_amdgcn_wave_barrier();
                if (ok && below == 0) sh.cnt[wave][dig] = (uint16_t)(before + (uint32_t)__popcll(peers));
                __builtin_amdgcn_wave_barrier();
            }
        }
        __syncthreads();
        uint32_t tot = 0, incl = 0;
        if (t < 256) {
            uint32_t run = 0;
#pragma unroll
            for (uint32_t w = 0; w < BS_NW; ++w) {
                const uint32_t c = sh.cnt[w][t];
                sh.cnt[w][t] = (uint16_t)run;
                run += c;
            }
            tot = run;
            incl = tot;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t v = __shfl_up(incl, o);
                if ((int)lane >= o) incl += v;
            }
            if (lane == 63) sh.wsum[wave] = incl;
        }
        __syncthreads();
        if (t < 256) {
            uint32_t woff = 0;
            for (uint32_t w = 0; w < wave; ++w) woff += sh.wsum[w];
            sh.tile_excl[t] = woff + incl - tot;
        }
        __syncthreads();
#pragma unroll
        for (uint32_t r = 0; r < BS_IPT; ++r) {
            const uint32_t dig = dg[r];
            if (dig != 0xffffffffu) sh.tile[sh.tile_excl[dig] + sh.cnt[wave][dig] + rk[r]] = it[r];
        }
        __syncthreads();
        if (k + 1 < n_passes) {
#pragma unroll
            for (uint32_t r = 0; r < BS_IPT; ++r) {
                const uint32_t p = bs_pos(wave, lane, r, rounds);
                if (r < rounds && p < n) it[r] = sh.tile[p];
            }
        }
    }
}

__device__ __forceinline__ uint4 key64_item(uint64_t k) { return make_uint4((uint32_t)k, (uint32_t)(k >> 32), 0u, 0u); }
__device__ __forceinline__ uint64_t item_key64(const uint4 &it) { return (uint64_t)it.x | ((uint64_t)it.y << 32); }

// number of splitters sp[0 .. n_sp) (ascending, n_sp <= 255) that are <= key
__device__ __forceinline__ uint32_t bs_upper(const uint64_t *sp, uint32_t n_sp, uint64_t key) {
    uint32_t c = 0;
#pragma unroll
    for (uint32_t step = 128; step; step >>= 1) {
        const uint32_t j = c + step;
        if (j <= n_sp && sp[j - 1] <= key) c = j;
    }
    return c;
}

// ---- 1a: first-level sample -> k1 - 1 coarse splitters (one block).  Also resets the sort's device state.
__global__ __launch_bounds__(BS_NT) void k_bs_sample1(const uint4 *__restrict__ items, uint64_t n, SqueezeSpec sq,
                                                      uint32_t k1, uint64_t *__restrict__ coarse,
                                                      uint32_t *__restrict__ list_n, uint32_t *__restrict__ overflow) {
    __shared__ BsShared sh;
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    if (t < 256) list_n[t] = 0;
    if (t == 0) *overflow = 0;
    const uint32_t m = (uint32_t)(n < BS_CAP ? n : BS_CAP);
    const uint32_t rounds = bs_rounds(m);
    uint4 it[BS_IPT];
#pragma unroll
    for (uint32_t r = 0; r < BS_IPT; ++r) {
        const uint32_t p = bs_pos(wave, lane, r, rounds);
        it[r] = make_uint4(0, 0, 0, 0);
        if (r < rounds && p < m) it[r] = key64_item(squeeze(items[((uint64_t)p * n) / m], sq));
    }
    bs_block_sort(sh, it, m, 64);
    // splitter j (1 <= j < k1) = the sample at rank j * m / k1
    if (t + 1 < k1) coarse[t] = item_key64(sh.tile[(uint32_t)(((uint64_t)(t + 1) * m) / k1)]);
}

// ---- 1b: second-level sample, dealt into the coarse ranges
__global__ __launch_bounds__(256) void k_bs_sample2(const uint4 *__restrict__ items, uint64_t n, uint32_t m2, SqueezeSpec sq,
                                                    uint32_t k1, const uint64_t *__restrict__ coarse,
                                                    uint32_t *__restrict__ list_n, uint64_t *__restrict__ lists) {
    __shared__ uint64_t cs[256];
    __shared__ uint32_t lc[256], lbase[256];
    const uint32_t t = threadIdx.x;
    cs[t] = (t + 1 < k1) ? coarse[t] : ~0ull;
    lc[t] = 0;
    __syncthreads();
    const uint32_t g = blockIdx.x * 256u + t;
    const bool ok = g < m2;
    uint64_t key = 0;
    uint32_t c = 0, rank = 0;
    if (ok) {
        key = squeeze(items[((uint64_t)g * n + n / 2) / m2], sq);
        c = bs_upper(cs, k1 - 1, key);
        rank = atomicAdd(&lc[c], 1u);
    }
    __syncthreads();
    if (t < k1 && lc[t]) lbase[t] = atomicAdd(&list_n[t], lc[t]);
    __syncthreads();
    if (ok) {
        const uint32_t pos = lbase[c] + rank;
        if (pos < BS_LISTCAP) lists[(uint64_t)c * BS_LISTCAP + pos] = key;
    }
}

// ---- 1c: per coarse range, its sample sorted -> up to 255 fine splitters (table padded with ~0)
__global__ __launch_bounds__(BS_NT) void k_bs_sort2(const uint32_t *__restrict__ list_n, const uint64_t *__restrict__ lists,
                                                    uint64_t *__restrict__ fine, uint32_t *__restrict__ fine_n) {
    __shared__ BsShared sh;
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6, c = blockIdx.x;
    const uint32_t have = list_n[c];
    const uint32_t m = have < BS_LISTCAP ? have : BS_LISTCAP;
    const uint32_t rounds = bs_rounds(m);
    uint4 it[BS_IPT];
#pragma unroll
    for (uint32_t r = 0; r < BS_IPT; ++r) {
        const uint32_t p = bs_pos(wave, lane, r, rounds);
        it[r] = make_uint4(0, 0, 0, 0);
        if (r < rounds && p < m) it[r] = key64_item(lists[(uint64_t)c * BS_LISTCAP + p]);
    }
    bs_block_sort(sh, it, m, 64);
    // leaves of this range: one per BS_OVERSAMPLE sample keys (at least one, at most 256)
    uint32_t f = (m + BS_OVERSAMPLE / 2) / BS_OVERSAMPLE;
    f = f < 1 ? 1 : (f > 256 ? 256 : f);
    if (t < 256) {
        uint64_t v = ~0ull;
        if (t + 1 < f) v = item_key64(sh.tile[(uint32_t)(((uint64_t)(t + 1) * m) / f)]);
        fine[(uint64_t)c * 256 + t] = v;
    }
    if (t == 0) fine_n[c] = f - 1;
}

// ---- 2: leaf id of every record into key bits 80..95, its low byte as the first pass's digit byte
__global__ __launch_bounds__(256) void k_bs_assign(uint4 *__restrict__ items, uint8_t *__restrict__ digs, uint64_t n,
                                                   SqueezeSpec sq, uint32_t k1, const uint64_t *__restrict__ coarse,
                                                   const uint64_t *__restrict__ fine, const uint32_t *__restrict__ fine_n,
                                                   uint32_t *__restrict__ n_leaves_out) {
    __shared__ uint64_t cs[256];
    __shared__ uint32_t fn[256], base[256], ws[4];
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    cs[t] = (t + 1 < k1) ? coarse[t] : ~0ull;
    const uint32_t my_fn = t < k1 ? fine_n[t] : 0u;
    fn[t] = my_fn;
    // base[c] = leaves of the ranges before c
    const uint32_t v = t < k1 ? my_fn + 1u : 0u;
    uint32_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t x = __shfl_up(incl, o);
        if ((int)lane >= o) incl += x;
    }
    if (lane == 63) ws[wave] = incl;
    __syncthreads();
    uint32_t off = 0;
    for (uint32_t w = 0; w < wave; ++w) off += ws[w];
    base[t] = off + incl - v;
    if (blockIdx.x == 0 && t == 0) *n_leaves_out = ws[0] + ws[1] + ws[2] + ws[3];
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * 256u;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + t; i < n; i += stride) {
        uint4 it = items[i];
        const uint64_t key = squeeze(it, sq);
        const uint32_t c = bs_upper(cs, k1 - 1, key);
        const uint32_t q = bs_upper(fine + (uint64_t)c * 256, fn[c], key);
        const uint32_t id = base[c] + q;
        it.z |= id << 16;
        items[i] = it;
        digs[i] = (uint8_t)id;
    }
}

// ---- 3a: first record of every leaf.  digs: the low id byte of every record, in sorted-by-id order; where it
// changes, the ids on both sides are read from the items.  start[] was set to BS_SENT; start[b] stays unset only
// for b whose boundary went unseen because the id jumped by a multiple of 256 over empty leaves (their records
// then count to the leaf before -- still one contiguous range of keys).
__global__ __launch_bounds__(256) void k_bs_bounds(const uint4 *__restrict__ items, const uint8_t *__restrict__ digs,
                                                   uint64_t n, const uint32_t *__restrict__ n_leaves_p,
                                                   uint32_t *__restrict__ start) {
    const uint64_t g = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    const uint64_t i0 = g * 16;
    if (g == 0) {
        const uint32_t first = items[0].z >> 16, last = items[n - 1].z >> 16, nl = *n_leaves_p;
        for (uint32_t b = 0; b <= first; ++b) start[b] = 0;
        for (uint32_t b = last + 1; b <= nl; ++b) start[b] = (uint32_t)n;
    }
    if (i0 >= n) return;
    uint8_t by[17];
    by[0] = i0 ? digs[i0 - 1] : digs[0];
    if (i0 + 16 <= n) {
        const uint4 v = *reinterpret_cast<const uint4 *>(digs + i0);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (uint32_t k = 0; k < 16; ++k) by[1 + k] = (uint8_t)(w[k >> 2] >> (8 * (k & 3u)));
    } else {
#pragma unroll
        for (uint32_t k = 0; k < 16; ++k) by[1 + k] = (i0 + k < n) ? digs[i0 + k] : by[k];
    }
#pragma unroll
    for (uint32_t k = 0; k < 16; ++k) {
        if (by[1 + k] != by[k]) {
            const uint64_t i = i0 + k;
            const uint32_t p = items[i - 1].z >> 16, c = items[i].z >> 16;
            for (uint32_t b = p + 1; b <= c; ++b) start[b] = (uint32_t)i;
        }
    }
}

// ---- 3b: every leaf sorted in LDS, written back without its id.  direct_n != 0: the whole array is one leaf.
__global__ __launch_bounds__(BS_NT) void k_bs_leaf(uint4 *__restrict__ items, const uint32_t *__restrict__ start,
                                                   const uint32_t *__restrict__ n_leaves_p, uint32_t direct_n,
                                                   uint32_t max_bits, uint32_t *__restrict__ overflow) {
    __shared__ BsShared sh;
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const uint32_t n_leaves = direct_n ? 1u : *n_leaves_p;
#pragma unroll 1
    for (uint32_t b = blockIdx.x; b < n_leaves; b += gridDim.x) {
        if (t == 0) {
            uint32_t s = 0, e = direct_n;
            if (!direct_n) {
                s = start[b];
                e = s;
                if (s != BS_SENT) {
                    uint32_t k = b + 1;
                    while ((e = start[k]) == BS_SENT) ++k;          // start[n_leaves] is always set
                }
            }
            sh.rng[0] = s;
            sh.rng[1] = e;
        }
        __syncthreads();
        const uint32_t s = sh.rng[0], n = sh.rng[1] - s;
        if (s == BS_SENT || n == 0) {
            __syncthreads();
            continue;
        }
        if (n <= BS_CAP) {
            const uint32_t rounds = bs_rounds(n);
            uint4 it[BS_IPT];
#pragma unroll
            for (uint32_t r = 0; r < BS_IPT; ++r) {
                const uint32_t p = bs_pos(wave, lane, r, rounds);
                it[r] = make_uint4(0, 0, 0, 0);
                if (r < rounds && p < n) {
                    it[r] = items[(uint64_t)s + p];
                    it[r].z &= 0xffffu;
                }
            }
            bs_block_sort(sh, it, n, max_bits);
            for (uint32_t slot = t; slot < n; slot += BS_NT) items[(uint64_t)s + slot] = sh.tile[slot];
        } else {
            // too large for LDS: acceptable only if its keys are all equal.  The ids come off either way.
            uint32_t v[6] = {0u, 0u, 0u, ~0u, ~0u, ~0u};
            for (uint32_t i = t; i < n; i += BS_NT) {
                uint4 x = items[(uint64_t)s + i];
                x.z &= 0xffffu;
                items[(uint64_t)s + i] = x;
                v[0] |= x.x; v[1] |= x.y; v[2] |= x.z;
                v[3] &= x.x; v[4] &= x.y; v[5] &= x.z;
            }
            const bool differ = ((v[0] ^ v[3]) | (v[1] ^ v[4]) | (v[2] ^ v[5])) != 0u;
            // every thread saw at least one record (n > BS_CAP >= BS_NT): any difference shows in some thread's
            // OR ^ AND or between two threads' values; compare all with thread 0's first record through LDS
            if (t == 0) { sh.vary[0] = v[0]; sh.vary[1] = v[1]; sh.vary[2] = v[2]; }
            __syncthreads();
            const bool other = v[0] != sh.vary[0] || v[1] != sh.vary[1] || v[2] != sh.vary[2];
            if (ipx_ballot(differ || other) && lane == 0) atomicOr(overflow, 1u);
        }
        __syncthreads();
    }
}

SqueezeSpec make_squeeze(const uint32_t vary[3], uint32_t total_bits) {
    SqueezeSpec sq{};
    auto varies = [&](int b) { return ((vary[b >> 5] >> (b & 31)) & 1u) != 0; };
    uint32_t have = 0;
    int b = (int)total_bits - 1;
    while (b >= 0 && have < 64 && sq.n < 6) {
        if (!varies(b)) { --b; continue; }
        const int top = b;
        while (b > 0 && varies(b - 1) && (uint32_t)(top - b + 1) + have < 64) --b;
        const uint32_t w = (uint32_t)(top - b + 1);
        sq.lo[sq.n] = (uint32_t)b;
        sq.w[sq.n] = w;
        ++sq.n;
        have += w;
        --b;
    }
    uint32_t used = 0;
    for (uint32_t r = 0; r < sq.n; ++r) {
        used += sq.w[r];
        sq.sh[r] = have - used;
    }
    return sq;
}

}  // namespace

bool ipx_bsort_applicable(uint64_t n, const KeyLayout &kl) {
    return n >= 1 && kl.total_bits <= BS_ID_LO && n <= (uint64_t)BS_MAX_LEAVES * BS_LEAF_AVG && n < (1ull << 32);
}

size_t ipx_bsort_workspace_bytes(void) {
    // coarse[256] u64 | list_n[256] | fine_n[256] | n_leaves (+ pad to 256 B) | fine[65536] u64 | start[65538] | lists[256 * LISTCAP] u64
    return (size_t)256 * 8 + 256 * 4 + 256 * 4 + 256 + (size_t)65536 * 8 + (size_t)65540 * 4 + (size_t)256 * BS_LISTCAP * 8;
}

hipError_t ipx_launch_bsort(hipStream_t s, uint4 *buf_a, uint4 *buf_b, uint8_t *dig_a, uint8_t *dig_b, uint64_t n,
                            const KeyLayout &kl, const uint32_t vary[3], void *sort_ws, size_t sort_ws_bytes,
                            void *bs_ws, uint32_t *overflow) {
    if (!ipx_bsort_applicable(n, kl)) return hipErrorInvalidValue;
    char *w = (char *)bs_ws;
    uint64_t *coarse = (uint64_t *)w;            w += 256 * 8;
    uint32_t *list_n = (uint32_t *)w;            w += 256 * 4;
    uint32_t *fine_n = (uint32_t *)w;            w += 256 * 4;
    uint32_t *n_leaves = (uint32_t *)w;          w += 256;
    uint64_t *fine = (uint64_t *)w;              w += (size_t)65536 * 8;
    uint32_t *start = (uint32_t *)w;             w += (size_t)65540 * 4;
    uint64_t *lists = (uint64_t *)w;
    static const uint32_t n_cus = [] {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        return (uint32_t)cus;
    }();
    if (n <= BS_CAP) {                            // one leaf: one launch
        hipError_t e = hipMemsetAsync(overflow, 0, sizeof(uint32_t), s);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_bs_leaf, dim3(1), dim3(BS_NT), 0, s, buf_a, start, n_leaves, (uint32_t)n, kl.total_bits, overflow);
        ipx_prof_mark(s, IPXK_BS_LEAF);
        return hipGetLastError();
    }
    const SqueezeSpec sq = make_squeeze(vary, kl.total_bits);
    const uint64_t k_target = std::max<uint64_t>(1, ipx_div_up(n, BS_LEAF_AVG));
    const uint32_t m2 = (uint32_t)std::min<uint64_t>(n, k_target * BS_OVERSAMPLE);
    // coarse ranges: ~1800 second-level sample keys each (a range that draws 1.6 x its share still fits its list),
    // and few enough leaves per range that an uneven range stays below 256
    uint32_t k1 = (uint32_t)std::max<uint64_t>(ipx_div_up(m2, 1800), ipx_div_up(k_target, 170));
    k1 = std::min<uint32_t>(std::max<uint32_t>(k1, 1u), 256u);
    hipLaunchKernelGGL(k_bs_sample1, dim3(1), dim3(BS_NT), 0, s, buf_a, n, sq, k1, coarse, list_n, overflow);
    hipLaunchKernelGGL(k_bs_sample2, dim3((uint32_t)ipx_div_up(m2, 256)), dim3(256), 0, s, buf_a, n, m2, sq, k1, coarse, list_n, lists);
    hipLaunchKernelGGL(k_bs_sort2, dim3(k1), dim3(BS_NT), 0, s, list_n, lists, fine, fine_n);
    const uint32_t assign_blocks = (uint32_t)std::min<uint64_t>(ipx_div_up(n, 256 * 4), (uint64_t)n_cus * 16);
    hipLaunchKernelGGL(k_bs_assign, dim3(assign_blocks), dim3(256), 0, s, buf_a, dig_a, n, sq, k1, coarse, fine, fine_n, n_leaves);
    hipError_t e = hipMemsetAsync(start, 0xff, (size_t)65540 * 4, s);
    if (e != hipSuccess) return e;
    ipx_prof_mark(s, IPXK_BS_PREP);
    SortPlan plan{};
    plan.n = 2;
    plan.digits[0] = DigitSpec{BS_ID_LO, 8, 0, 0};
    plan.digits[1] = DigitSpec{BS_ID_LO + 8, 8, 0, 0};
    const DigitSpec id_low{BS_ID_LO, 8, 0, 0};
    if ((e = ipx_launch_sort(s, buf_a, buf_b, dig_a, dig_b, true, n, plan, sort_ws, sort_ws_bytes, -1, &id_low)) != hipSuccess) return e;
    // two passes: the records are back in buf_a, the low id bytes the second pass wrote in dig_a
    hipLaunchKernelGGL(k_bs_bounds, dim3((uint32_t)ipx_div_up(ipx_div_up(n, 16), 256)), dim3(256), 0, s, buf_a, dig_a, n, n_leaves, start);
    ipx_prof_mark(s, IPXK_BS_PREP);
    const uint32_t leaf_blocks = (uint32_t)std::min<uint64_t>(k_target + k1, (uint64_t)n_cus * 8);
    hipLaunchKernelGGL(k_bs_leaf, dim3(leaf_blocks), dim3(BS_NT), 0, s, buf_a, start, n_leaves, 0u, kl.total_bits, overflow);
    ipx_prof_mark(s, IPXK_BS_LEAF);
    return hipGetLastError();
}
