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
//   1. SPLITTERS.  k_bs_sample1 (one block) sorts up to 4096 sampled keys in LDS and keeps k1 - 1 <= 255 coarse
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
#include <cstdio>
#include <cstdlib>

namespace {

constexpr uint32_t BS_NT = 256, BS_CAP = 4096, BS_NW = BS_NT / 64, BS_IPT = BS_CAP / BS_NT;   // 4 waves, 16 items per thread
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

// -DIPX_BS_STATS: where a leaf's cycles go (s_memtime laps on thread 0 of a few blocks, printed at the end)
#ifdef IPX_BS_STATS
#define BS_LAP(k) do { if (threadIdx.x == 0) { const unsigned long long now_ = __builtin_readcyclecounter(); st_ph[k] += now_ - st_t0; st_t0 = now_; } } while (0)
#define BS_ST_ARGS , unsigned long long (&st_ph)[8], unsigned long long &st_t0
#define BS_ST_PASS , st_ph, st_t0
#else
#define BS_LAP(k)
#define BS_ST_ARGS
#define BS_ST_PASS
#endif

template <uint32_t IPT>
struct BsShared {
    uint4 tile[BS_NT * IPT];             // 32 or 64 KiB
    uint32_t cnt[BS_NW][128];            // per-wave digit counts of a pass, two 16-bit counters per word
    uint16_t base[BS_NW][256];           // where a wave's run of a digit starts in the tile
    uint32_t vary[6];                    // OR / AND of the block's keys
    uint32_t rng[4];
};

// position of a thread's r-th item in the block's item order (wave, round, lane); rounds = rounds in use
__device__ __forceinline__ uint32_t bs_pos(uint32_t wave, uint32_t lane, uint32_t r, uint32_t rounds) {
    return (wave * rounds + r) * 64u + lane;
}
__device__ __forceinline__ uint32_t bs_rounds(uint32_t n) { return (n + BS_NT - 1) / BS_NT; }

// inclusive prefix sum over the 64 lanes with DPP row shifts and row broadcasts (six v_add_u32_dpp; the __shfl_up
// form is six dependent ds_bpermute round trips, ~700 cycles of every pass of bs_block_sort)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t x) {
    uint32_t v = x;
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);    // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);    // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);    // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);    // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
    return v;
}
// OR / AND over the wave the same way; the result is in lane 63
__device__ __forceinline__ uint32_t wave_or_to_last(uint32_t x) {
    uint32_t v = x;
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}
__device__ __forceinline__ uint32_t wave_and_to_last(uint32_t x) { return ~wave_or_to_last(~x); }

// key words shifted right by 32 * WI + sft bits (sft < 32, wave-uniform), and back
template <uint32_t WI>
__device__ __forceinline__ void bs_shr(uint4 &it, uint32_t sft) {
    const uint32_t a0 = WI == 0 ? it.x : (WI == 1 ? it.y : it.z);
    const uint32_t a1 = WI == 0 ? it.y : (WI == 1 ? it.z : 0u);
    const uint32_t a2 = WI == 0 ? it.z : 0u;
    it.x = __builtin_amdgcn_alignbit(a1, a0, sft);
    it.y = __builtin_amdgcn_alignbit(a2, a1, sft);
    it.z = a2 >> sft;
}
__device__ __forceinline__ uint32_t bs_shl_word(uint32_t hi, uint32_t lo, uint32_t sft) {      // (hi:lo << sft) >> 32
    return (uint32_t)(((((uint64_t)hi) << 32) | lo) << sft >> 32);
}
template <uint32_t WI>
__device__ __forceinline__ void bs_shl(uint4 &it, uint32_t sft) {
    const uint32_t b0 = it.x, b1 = it.y, b2 = it.z;
    if (WI == 0) { it.x = b0 << sft; it.y = bs_shl_word(b1, b0, sft); it.z = bs_shl_word(b2, b1, sft); }
    else if (WI == 1) { it.x = 0; it.y = b0 << sft; it.z = bs_shl_word(b1, b0, sft); }
    else { it.x = 0; it.y = 0; it.z = b0 << sft; }
}

// what bs_block_sort leaves in the tile: the keys shifted right by `lo` bits; low = the bits below (equal in all keys)
struct BsSorted {
    uint32_t lo;
    uint32_t low[3];
};
__device__ __forceinline__ uint4 bs_restore(uint4 it, const BsSorted &bs) {
    const uint32_t wi = bs.lo >> 5, sft = bs.lo & 31u;
    if (wi == 0) bs_shl<0>(it, sft); else if (wi == 1) bs_shl<1>(it, sft); else bs_shl<2>(it, sft);
    it.x |= bs.low[0]; it.y |= bs.low[1]; it.z |= bs.low[2];
    return it;
}

// Sorts the block's n <= 256 * IPT items (it[r] = the item at bs_pos(wave, lane, r, rounds) where that is < n) by
// their 96-bit keys; the sorted items are left in sh.tile[0, n) (visible to all threads on return) with the keys
// shifted right by the returned `lo` (bs_restore undoes it).  An LSD radix sort whose items live in registers
// between passes:
//  * the keys' OR and AND tell on which bits they differ; every key is shifted right by the lowest such bit, so
//    that pass k looks at byte k of the shifted key (one v_bfe_u32; a digit spelled as two runs of bits at run-time
//    positions compiled to forty scalar branches per item) and bytes on which no key differs are skipped;
//  * slots up to the next multiple of 256 are filled with all-ones keys, which sort behind everything, so the
//    passes know no invalid items;
//  * ranking: ONE LDS returning atomic per item, on a 16-bit counter of its wave (two per word).  Lanes of one
//    instruction that hit the same word get their old values in lane order on this hardware
//    (profiles/microbench/lds_atomic_order.hip: 1.3 x 10^10 lane-ops, no exception; bsort_selftest checks it again
//    in every process before the bucket sort is used), and a wave's LDS instructions execute in order -- so the
//    value returned IS the item's rank among the equal digits of its wave in (round, lane) order, which is what a
//    stable split needs.  The ballot-matching of k_scatter costs ~50 instructions per item for the same number;
//  * two barriers per pass: after the counts every wave scans all 256 digit totals ITSELF (four digits per lane,
//    one DPP scan) and writes only its own row of bases, which only it reads; the second barrier stands between
//    placing the items in the tile and taking them back.
// BARRIERS (round-4 audit): `n` is the caller's block-uniform leaf size.  The early return `if (n <= 1 || no varying bit)`
// tests n and vs[], which every thread reads from sh.vary[] after the barrier that follows its last write; the pass loop
// `if (!((active >> k) & 1u)) continue` tests `active`, computed by every thread from the same vs[] (readfirstlane'd); the
// two barriers per pass are unconditional inside the loop body.
template <uint32_t IPT>
__device__ __forceinline__ BsSorted bs_block_sort(BsShared<IPT> &sh, uint4 (&it)[IPT], uint32_t n BS_ST_ARGS) {
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const uint32_t rounds = bs_rounds(n);
    BsSorted out{0u, {0u, 0u, 0u}};
    if (t < 6) sh.vary[t] = t < 3 ? 0u : ~0u;
    __syncthreads();
    {
        uint32_t v[6] = {0u, 0u, 0u, ~0u, ~0u, ~0u};
#pragma unroll
        for (uint32_t r = 0; r < IPT; ++r)
            if (r < rounds && bs_pos(wave, lane, r, rounds) < n) {
                v[0] |= it[r].x; v[1] |= it[r].y; v[2] |= it[r].z;
                v[3] &= it[r].x; v[4] &= it[r].y; v[5] &= it[r].z;
            }
#pragma unroll
        for (int k = 0; k < 3; ++k) { v[k] = wave_or_to_last(v[k]); v[3 + k] = wave_and_to_last(v[3 + k]); }
        if (lane == 63) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (v[k]) atomicOr(&sh.vary[k], v[k]);
                if (v[3 + k] != ~0u) atomicAnd(&sh.vary[3 + k], v[3 + k]);
            }
        }
    }
    __syncthreads();
    uint32_t vs[3], an[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        an[k] = __builtin_amdgcn_readfirstlane(sh.vary[3 + k]);
        vs[k] = __builtin_amdgcn_readfirstlane(sh.vary[k]) ^ an[k];
    }
    BS_LAP(1);                                             // OR / AND of the keys
    if (n <= 1 || (vs[0] | vs[1] | vs[2]) == 0u) {         // all keys equal (or none): any order is sorted
#pragma unroll
        for (uint32_t r = 0; r < IPT; ++r) {
            const uint32_t p = bs_pos(wave, lane, r, rounds);
            if (r < rounds && p < n) sh.tile[p] = it[r];
        }
        __syncthreads();
        return out;
    }
    const uint32_t lo = vs[0] ? (uint32_t)__builtin_ctz(vs[0]) : (vs[1] ? 32u + (uint32_t)__builtin_ctz(vs[1]) : 64u + (uint32_t)__builtin_ctz(vs[2]));
    const uint32_t wi = lo >> 5, sft = lo & 31u;
    out.lo = lo;
    {   // bits below lo: the same in every key
        const uint32_t m = sft ? ((1u << sft) - 1u) : 0u;
        out.low[0] = wi == 0 ? (an[0] & m) : an[0];
        out.low[1] = wi == 0 ? 0u : (wi == 1 ? (an[1] & m) : an[1]);
        out.low[2] = wi == 2 ? (an[2] & m) : 0u;
    }
    // shifted keys; slots past n hold all-ones keys
    uint4 vv = make_uint4(vs[0], vs[1], vs[2], 0u);
    if (wi == 0) {
        bs_shr<0>(vv, sft);
#pragma unroll
        for (uint32_t r = 0; r < IPT; ++r) if (r < rounds) bs_shr<0>(it[r], sft);
    } else if (wi == 1) {
        bs_shr<1>(vv, sft);
#pragma unroll
        for (uint32_t r = 0; r < IPT; ++r) if (r < rounds) bs_shr<1>(it[r], sft);
    } else {
        bs_shr<2>(vv, sft);
#pragma unroll
        for (uint32_t r = 0; r < IPT; ++r) if (r < rounds) bs_shr<2>(it[r], sft);
    }
#pragma unroll
    for (uint32_t r = 0; r < IPT; ++r)
        if (r < rounds && bs_pos(wave, lane, r, rounds) >= n) it[r] = make_uint4(~0u, ~0u, ~0u, ~0u);
    // bytes of the shifted key on which the keys differ
    uint32_t active = 0;
#pragma unroll
    for (uint32_t k = 0; k < 12; ++k) {
        const uint32_t w = k < 4 ? vv.x : (k < 8 ? vv.y : vv.z);
        if ((w >> (8 * (k & 3u))) & 0xffu) active |= 1u << k;
    }
    active = __builtin_amdgcn_readfirstlane(active);
    const uint32_t last_k = 31u - (uint32_t)__builtin_clz(active);
    uint2 *cnt2 = reinterpret_cast<uint2 *>(&sh.cnt[0][0]);             // [wave][64]: digits 4 * lane .. 4 * lane + 3
    uint2 *base2 = reinterpret_cast<uint2 *>(&sh.base[0][0]);
#pragma unroll 1
    for (uint32_t k = 0; k <= last_k; ++k) {
        if (!((active >> k) & 1u)) continue;
        const uint32_t sel = k >> 2, bsh = 8u * (k & 3u);
        cnt2[wave * 64 + lane] = make_uint2(0, 0);
        __builtin_amdgcn_wave_barrier();
        uint32_t rd[IPT];                                  // rank among the wave's equal digits | digit << 16
#define BS_RANK(WORD)                                                                       \
        _Pragma("unroll") for (uint32_t r = 0; r < IPT; ++r) {                              \
            rd[r] = 0;                                                                      \
            if (r < rounds) {                                                               \
                const uint32_t dig = (it[r].WORD >> bsh) & 0xffu;                           \
                const uint32_t hs = (dig & 1u) * 16u;                                       \
                const uint32_t old = atomicAdd(&sh.cnt[wave][dig >> 1], 1u << hs);          \
                rd[r] = ((old >> hs) & 0xffffu) | (dig << 16);                              \
            }                                                                               \
        }
        if (sel == 0) { BS_RANK(x) } else if (sel == 1) { BS_RANK(y) } else { BS_RANK(z) }
#undef BS_RANK
        __syncthreads();
        BS_LAP(2);                                         // digits, ranks, barrier
        {
            uint2 tot = make_uint2(0, 0), pre = make_uint2(0, 0);           // packed pairs; no field exceeds 4096
#pragma unroll
            for (uint32_t w = 0; w < BS_NW; ++w) {
                const uint2 c = cnt2[w * 64 + lane];
                tot.x += c.x; tot.y += c.y;
                if (w < wave) { pre.x += c.x; pre.y += c.y; }
            }
            const uint32_t t0 = tot.x & 0xffffu, t1 = tot.x >> 16, t2 = tot.y & 0xffffu, t3 = tot.y >> 16;
            const uint32_t mine = t0 + t1 + t2 + t3;
            const uint32_t excl = wave_incl_scan(mine) - mine;          // items of the digits below 4 * lane
            const uint32_t b0 = excl + (pre.x & 0xffffu), b1 = excl + t0 + (pre.x >> 16),
                           b2 = excl + t0 + t1 + (pre.y & 0xffffu), b3 = excl + t0 + t1 + t2 + (pre.y >> 16);
            base2[wave * 64 + lane] = make_uint2(b0 | (b1 << 16), b2 | (b3 << 16));
        }
        __builtin_amdgcn_wave_barrier();
        BS_LAP(3);                                         // digit totals -> bases
#pragma unroll
        for (uint32_t r = 0; r < IPT; ++r)
            if (r < rounds) sh.tile[(uint32_t)sh.base[wave][rd[r] >> 16] + (rd[r] & 0xffffu)] = it[r];
        __syncthreads();
        BS_LAP(4);                                         // placement, barrier
        if (k < last_k) {
#pragma unroll
            for (uint32_t r = 0; r < IPT; ++r)
                if (r < rounds) it[r] = sh.tile[bs_pos(wave, lane, r, rounds)];
        }
        BS_LAP(5);                                         // items back into registers
    }
    return out;
}

// Start-up check of the property bs_block_sort's ranking rests on (see there): random digits, few and many equal
// ones per wave, the atomic's return value against the ballot-matched rank.  *bad counts the differences.
__global__ __launch_bounds__(256) void k_bs_selftest(uint32_t iters, unsigned long long *bad) {
    // (the counters as bs_block_sort has them: two 16-bit counters per word, the digit's parity picks the half, so lanes
    // of one instruction add DIFFERENT increments to one word -- round 3's check had a word per digit)
    __shared__ uint32_t cnt[4][128];
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    unsigned long long mism = 0;
    for (uint32_t it = 0; it < iters; ++it) {
#pragma unroll
        for (uint32_t q = 0; q < 2; ++q) cnt[wave][q * 64 + lane] = 0;
        __builtin_amdgcn_wave_barrier();
        uint32_t h = (blockIdx.x * 256u + t) * 2654435761u + it * 40503u;
        h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
        const uint32_t dig = (it & 1u) ? (h % (1u + (it >> 1) % 9u)) : (h & 255u);
        const uint32_t hs = (dig & 1u) * 16u;
        const uint32_t r_atomic = (atomicAdd(&cnt[wave][dig >> 1], 1u << hs) >> hs) & 0xffffu;
        unsigned long long peers = ~0ull;
#pragma unroll
        for (uint32_t b = 0; b < 8; ++b) {
            const unsigned long long bm = ipx_ballot((dig >> b) & 1u);
            peers &= ((dig >> b) & 1u) ? bm : ~bm;
        }
        const uint32_t r_ballot = __builtin_amdgcn_mbcnt_hi((uint32_t)(peers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)peers, 0u));
        if (r_atomic != r_ballot) ++mism;
        // ... and the DPP prefix sum against the shuffle form
        uint32_t ref = h & 1023u;
        const uint32_t got = wave_incl_scan(ref);
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t x = __shfl_up(ref, o);
            if ((int)lane >= o) ref += x;
        }
        if (got != ref) ++mism;
        __builtin_amdgcn_wave_barrier();
    }
    if (mism) atomicAdd(bad, mism);
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
                                                      uint32_t k1, uint32_t m1, uint64_t *__restrict__ coarse,
                                                      uint32_t *__restrict__ list_n, uint32_t *__restrict__ overflow) {
    __shared__ BsShared<BS_IPT> sh;
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    list_n[t] = 0;                            // BS_NT == 256
    if (t == 0) *overflow = 0;
    const uint32_t m = (uint32_t)(n < m1 ? n : m1);
    const uint32_t rounds = bs_rounds(m);
    uint4 it[BS_IPT];
#pragma unroll
    for (uint32_t r = 0; r < BS_IPT; ++r) {
        const uint32_t p = bs_pos(wave, lane, r, rounds);
        it[r] = make_uint4(0, 0, 0, 0);
        if (r < rounds && p < m) it[r] = key64_item(squeeze(items[((uint64_t)p * n) / m], sq));
    }
#ifdef IPX_BS_STATS
    unsigned long long st_ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_t0 = 0;
#endif
    const BsSorted bs = bs_block_sort(sh, it, m BS_ST_PASS);
    // splitter j (1 <= j < k1) = the sample at rank j * m / k1
    if (t + 1 < k1) coarse[t] = item_key64(bs_restore(sh.tile[(uint32_t)(((uint64_t)(t + 1) * m) / k1)], bs));
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
    __shared__ BsShared<BS_IPT> sh;
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
#ifdef IPX_BS_STATS
    unsigned long long st_ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_t0 = 0;
#endif
    const BsSorted bs = bs_block_sort(sh, it, m BS_ST_PASS);
    // leaves of this range: one per BS_OVERSAMPLE sample keys (at least one, at most 256): candidate t + 1 of f - 1 is
    // the sample key at rank (t + 1) * m / f.  A key with so many copies that it makes up a run of the sorted sample
    // (10 equal sample keys stand for ~0.6 of a leaf) is FENCED: it is emitted as the pair v, v + 1, so that the leaf
    // between them holds the copies of v and nothing else (in squeezed form) -- k_bs_leaf accepts a leaf of any size
    // whose keys are all equal; unfenced, such a key shares its leaf with whole neighbouring keys and overflows it.
    // A candidate equal to its predecessor is dropped.  At most 255 splitters: what does not fit is dropped too.
    uint32_t f = (m + BS_OVERSAMPLE / 2) / BS_OVERSAMPLE;
    f = f < 1 ? 1 : (f > 256 ? 256 : f);
    uint64_t v = ~0ull;
    uint32_t emit = 0;
    if (t + 1 < f) {
        const uint32_t r = (uint32_t)(((uint64_t)(t + 1) * m) / f);
        v = item_key64(bs_restore(sh.tile[r], bs));
        const bool dup = t >= 1 && v == item_key64(bs_restore(sh.tile[(uint32_t)(((uint64_t)t * m) / f)], bs));
        const bool heavy = v != ~0ull && ((r + 10 < m && item_key64(bs_restore(sh.tile[r + 10], bs)) == v) ||
                                          (r >= 10 && item_key64(bs_restore(sh.tile[r - 10], bs)) == v));
        emit = dup ? 0u : (heavy ? 2u : 1u);
    }
    const uint32_t incl = wave_incl_scan(emit);
    __syncthreads();                                        // (everybody is done with sh.vary)
    if (lane == 63) sh.rng[wave] = incl;
    fine[(uint64_t)c * 256 + t] = ~0ull;                    // BS_NT == 256: the table's padding
    __syncthreads();
    uint32_t pos = incl - emit, total = 0;
    for (uint32_t w = 0; w < BS_NW; ++w) { if (w < wave) pos += sh.rng[w]; total += sh.rng[w]; }
    if (emit >= 1 && pos < 255) fine[(uint64_t)c * 256 + pos] = v;
    if (emit == 2 && pos + 1 < 255) fine[(uint64_t)c * 256 + pos + 1] = v + 1;
    if (t == 0) fine_n[c] = total < 255 ? total : 255;
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
    if (blockIdx.x == 0 && t == 0) { n_leaves_out[0] = ws[0] + ws[1] + ws[2] + ws[3]; n_leaves_out[1] = 0; }   // [1]: k_bs_big_leaves' count
    __syncthreads();
    // Four records per thread at a time: a record's search is a chain of sixteen dependent look-ups (eight in the coarse
    // table in LDS, eight in its range's fine table in L2), and one chain per thread left the kernel waiting -- 0.47 ms
    // on L for 1.3 GB of traffic.  The four chains' look-ups of a step go out together; every look-up is made (from an
    // index inside the padded table) and the range check selects, so that no load sits behind a branch.
    constexpr uint32_t U = 4;
    const uint64_t stride = (uint64_t)gridDim.x * 256u * U;
    for (uint64_t i0 = (uint64_t)blockIdx.x * 256u * U + t; i0 < n; i0 += stride) {
        uint4 it[U];
        uint64_t key[U];
        uint32_t c[U], q[U], fnu[U];
        const uint64_t *f[U];
#pragma unroll
        for (uint32_t u = 0; u < U; ++u) {
            const uint64_t i = i0 + (uint64_t)u * 256u;
            it[u] = items[i < n ? i : n - 1];
        }
#pragma unroll
        for (uint32_t u = 0; u < U; ++u) { key[u] = squeeze(it[u], sq); c[u] = 0; q[u] = 0; }
#pragma unroll
        for (uint32_t step = 128; step; step >>= 1)
#pragma unroll
            for (uint32_t u = 0; u < U; ++u) {
                const uint32_t j = c[u] + step;                      // (j - 1 <= 254: inside cs[256])
                const uint64_t v = cs[j - 1];
                if (j <= k1 - 1 && v <= key[u]) c[u] = j;
            }
#pragma unroll
        for (uint32_t u = 0; u < U; ++u) { f[u] = fine + (uint64_t)c[u] * 256; fnu[u] = fn[c[u]]; }
#pragma unroll
        for (uint32_t step = 128; step; step >>= 1)
#pragma unroll
            for (uint32_t u = 0; u < U; ++u) {
                const uint32_t j = q[u] + step;                      // (inside the range's 256-entry row, padded with ~0)
                const uint64_t v = f[u][j - 1];
                if (j <= fnu[u] && v <= key[u]) q[u] = j;
            }
#pragma unroll
        for (uint32_t u = 0; u < U; ++u) {
            const uint64_t i = i0 + (uint64_t)u * 256u;
            if (i < n) {
                const uint32_t id = base[c[u]] + q[u];
                it[u].z |= id << 16;
                items[i] = it[u];
                digs[i] = (uint8_t)id;
            }
        }
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

// ---- 3a': the leaves of more than n_min records, listed (in any order) for the launch that sorts them.  That launch
// used to walk ALL leaves, block b taking b, b + grid, ...: one leaf in twenty is large, so a block met none or three of
// them and the kernel took as long as its unluckiest block -- 80 us for the 600 large leaves of 9.9 M unfolded short-read
// records, 140 us for L's.  A leaf's bounds are read the way k_bs_leaf reads them.
__global__ __launch_bounds__(256) void k_bs_big_leaves(const uint32_t *__restrict__ start, uint32_t *__restrict__ n_leaves_p,
                                                       uint32_t n_min, uint32_t *__restrict__ list) {
    const uint32_t b = blockIdx.x * 256u + threadIdx.x, lane = threadIdx.x & 63u, n_leaves = n_leaves_p[0];
    bool big = false;
    if (b < n_leaves) {
        const uint32_t s = start[b];
        if (s != BS_SENT) {
            uint32_t e = start[b + 1];
            if (e == BS_SENT) {
                uint32_t k = b + 2;
                while ((e = start[k]) == BS_SENT) ++k;              // start[n_leaves] is always set
            }
            big = e - s > n_min;
        }
    }
    const uint64_t m = ipx_ballot(big);                              // one atomic per wave
    if (m == 0) return;
    uint32_t base = 0;
    if (lane == (uint32_t)__builtin_ctzll(m)) base = atomicAdd(n_leaves_p + 1, (uint32_t)__builtin_popcountll(m));
    base = __shfl(base, __builtin_ctzll(m));
    if (big) list[base + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull))] = b;
}

// ---- 3b: every leaf sorted in LDS, written back without its id.  direct_n != 0: the whole array is one leaf.
// Two launches share the leaves: IPT = 8 takes those of up to 2048 records (32 KiB of tile: four blocks per CU;
// 19 of 20 leaves), IPT = 16 the larger ones (n > n_min) and what does not fit LDS at all.
template <uint32_t IPT>
__global__ __launch_bounds__(BS_NT) void k_bs_leaf(uint4 *__restrict__ items, const uint32_t *__restrict__ start,
                                                   const uint32_t *__restrict__ n_leaves_p, uint32_t direct_n,
                                                   uint32_t n_min, uint32_t max_bits, uint32_t *__restrict__ overflow,
                                                   const uint32_t *__restrict__ list) {
    constexpr uint32_t CAP = BS_NT * IPT;
    __shared__ BsShared<IPT> sh;
#ifdef IPX_BS_STATS
    unsigned long long st_ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_t0 = __builtin_readcyclecounter();
    uint32_t st_leaves = 0;
#endif
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    // list: this launch takes the leaves k_bs_big_leaves listed (n_leaves_p[1] of them) instead of all; uniform per launch
    const uint32_t n_leaves = direct_n ? 1u : (list ? n_leaves_p[1] : n_leaves_p[0]);
    // thread 0 fetches a leaf's bounds one leaf ahead: two dependent loads (three with a list) that would otherwise open
    // every leaf
    uint32_t nx_b = blockIdx.x, nx_s = 0, nx_e = direct_n;
    if (t == 0 && !direct_n && blockIdx.x < n_leaves) {
        if (list) nx_b = list[blockIdx.x];
        nx_s = start[nx_b];
        nx_e = start[nx_b + 1];
    }
#pragma unroll 1
    for (uint32_t li = blockIdx.x; li < n_leaves; li += gridDim.x) {
        if (t == 0) {
            uint32_t s = nx_s, e = nx_e;
            if (!direct_n) {
                if (s != BS_SENT && e == BS_SENT) {
                    uint32_t k = nx_b + 2;
                    while ((e = start[k]) == BS_SENT) ++k;          // start[n_leaves] is always set
                }
                const uint32_t ni = li + gridDim.x;
                if (ni < n_leaves) {
                    nx_b = list ? list[ni] : ni;
                    nx_s = start[nx_b];
                    nx_e = start[nx_b + 1];
                }
            }
            sh.rng[0] = s;
            sh.rng[1] = e;
        }
        __syncthreads();
        // (uniform: sh.rng[] was written by thread 0 before the barrier above and is written again only after the barrier
        // that closes the leaf -- the one below, or the one at the loop's end; n_min, CAP: arguments / constants.  Every
        // branch of the leaf -- skip, sort, too-large -- ends with exactly one barrier before the next leaf's bounds)
        const uint32_t s = sh.rng[0], n = sh.rng[1] - s;
        if (s == BS_SENT || n <= n_min || (n > CAP && CAP < BS_CAP)) {       // nothing here, or the other launch's leaf
            __syncthreads();
            continue;
        }
        if (n <= CAP) {
            const uint32_t rounds = bs_rounds(n);
            uint4 it[IPT];
#pragma unroll
            // (every lane loads, from a clamped address, and selects: behind `if (p < n)` the compiler waited for each of
            // the IPT loads before it issued the next -- eight round trips to memory per leaf, the 5.5-9 k cycles of "load")
            for (uint32_t r = 0; r < IPT; ++r) {
                const uint32_t p = bs_pos(wave, lane, r, rounds);
                const bool ok = r < rounds && p < n;
                it[r] = items[(uint64_t)s + (ok ? p : 0u)];
            }
#pragma unroll
            for (uint32_t r = 0; r < IPT; ++r) {
                const uint32_t p = bs_pos(wave, lane, r, rounds);
                const bool ok = r < rounds && p < n;
                it[r].z &= 0xffffu;
                if (!ok) it[r] = make_uint4(0, 0, 0, 0);
            }
            BS_LAP(0);                                     // bounds, loads issued (their wait falls into the next lap)
            const BsSorted bs = bs_block_sort(sh, it, n BS_ST_PASS);
            for (uint32_t slot = t; slot < n; slot += BS_NT) items[(uint64_t)s + slot] = bs_restore(sh.tile[slot], bs);
#ifdef IPX_BS_STATS
            ++st_leaves;
#endif
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
        BS_LAP(6);                                         // store + end-of-leaf barrier
    }
#ifdef IPX_BS_STATS
    if (t == 0 && st_leaves && (blockIdx.x % 97) == 0)
        printf("leaf<%u> block %u: %u leaves; cycles per leaf: load %llu, or/and %llu, rank %llu, scan %llu, place %llu, back %llu, store %llu\n",
               IPT, blockIdx.x, st_leaves, st_ph[0] / st_leaves, st_ph[1] / st_leaves, st_ph[2] / st_leaves,
               st_ph[3] / st_leaves, st_ph[4] / st_leaves, st_ph[5] / st_leaves, st_ph[6] / st_leaves);
#endif
}

// The top 64 varying key bits as at most six runs.  A mask in more pieces (few distinct values in a field leave
// scattered bits) has its closest runs joined across the constant bits between them -- those cost width, not order --
// and what then exceeds 64 bits loses its low end: ties, which go to the same leaf.
SqueezeSpec make_squeeze(const uint32_t vary[3], uint32_t total_bits) {
    auto varies = [&](int b) { return ((vary[b >> 5] >> (b & 31)) & 1u) != 0; };
    struct Run { int lo, hi; };                  // bits [lo, hi], top run first
    Run runs[96];
    int n = 0;
    for (int b = (int)total_bits - 1; b >= 0;) {
        if (!varies(b)) { --b; continue; }
        const int hi = b;
        while (b > 0 && varies(b - 1)) --b;
        runs[n++] = Run{b, hi};
        --b;
    }
    while (n > 6) {                              // join the two neighbours with the fewest constant bits between them
        int best = 0, gap = 1 << 30;
        for (int i = 0; i + 1 < n; ++i) {
            const int g = runs[i].lo - runs[i + 1].hi - 1;
            if (g < gap) { gap = g; best = i; }
        }
        runs[best].lo = runs[best + 1].lo;
        for (int i = best + 1; i + 1 < n; ++i) runs[i] = runs[i + 1];
        --n;
    }
    SqueezeSpec sq{};
    uint32_t have = 0;
    for (int i = 0; i < n && have < 64; ++i) {
        uint32_t w = (uint32_t)(runs[i].hi - runs[i].lo + 1);
        uint32_t lo = (uint32_t)runs[i].lo;
        if (have + w > 64) { lo += have + w - 64; w = 64 - have; }     // keep the run's top bits
        sq.lo[sq.n] = lo;
        sq.w[sq.n] = w;
        ++sq.n;
        have += w;
    }
    uint32_t used = 0;
    for (uint32_t r = 0; r < sq.n; ++r) {
        used += sq.w[r];
        sq.sh[r] = have - used;
    }
    return sq;
}

}  // namespace

// Once per process: does an LDS returning atomic rank the lanes that hit one address in lane order (bs_block_sort)?
// 64 blocks x 256 threads x 400 rounds against the ballot-matched rank, ~0.1 ms.  A device that fails it, or any HIP
// error on the way, leaves the bucket sort switched off: ipx_finalize then sorts by radix passes as before.
static bool bsort_selftest() {
    static const bool ok = [] {
        unsigned long long *bad = nullptr, h = 1;
        if (hipMalloc(&bad, sizeof h) != hipSuccess) return false;
        bool good = hipMemset(bad, 0, sizeof h) == hipSuccess;
        if (good) {
            hipLaunchKernelGGL(k_bs_selftest, dim3(64), dim3(256), 0, 0, 400u, bad);
            good = hipGetLastError() == hipSuccess && hipMemcpy(&h, bad, sizeof h, hipMemcpyDeviceToHost) == hipSuccess && h == 0;
        }
        (void)hipFree(bad);
        if (!good) fprintf(stderr, "ipx: LDS atomics do not rank equal lanes in lane order on this device: bucket sort off\n");
        return good;
    }();
    return ok;
}

bool ipx_bsort_applicable(uint64_t n, const KeyLayout &kl) {
    return n >= 1 && kl.total_bits <= BS_ID_LO && n <= (uint64_t)BS_MAX_LEAVES * BS_LEAF_AVG && n < (1ull << 32) && bsort_selftest();
}

size_t ipx_bsort_workspace_bytes(void) {
    // coarse[256] u64 | list_n[256] | fine_n[256] | n_leaves, large leaves (+ pad to 256 B) | fine[65536] u64 | start[65538] |
    // big_list[65536] | lists[256 * LISTCAP] u64
    return (size_t)256 * 8 + 256 * 4 + 256 * 4 + 256 + (size_t)65536 * 8 + (size_t)65540 * 4 + (size_t)65540 * 4 + (size_t)256 * BS_LISTCAP * 8;
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
    uint32_t *big_list = (uint32_t *)w;          w += (size_t)65540 * 4;
    uint64_t *lists = (uint64_t *)w;
    static const uint32_t n_cus = [] {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        return (uint32_t)cus;
    }();
    if (n <= BS_CAP) {                            // one leaf: one launch
        hipError_t e = hipMemsetAsync(overflow, 0, sizeof(uint32_t), s);
        if (e != hipSuccess) return e;
        if (n <= BS_CAP / 2)
            hipLaunchKernelGGL(k_bs_leaf<BS_IPT / 2>, dim3(1), dim3(BS_NT), 0, s, buf_a, start, n_leaves, (uint32_t)n, 0u, kl.total_bits, overflow, nullptr);
        else
            hipLaunchKernelGGL(k_bs_leaf<BS_IPT>, dim3(1), dim3(BS_NT), 0, s, buf_a, start, n_leaves, (uint32_t)n, 0u, kl.total_bits, overflow, nullptr);
        ipx_prof_mark(s, IPXK_BS_LEAF);
        return hipGetLastError();
    }
    const SqueezeSpec sq = make_squeeze(vary, kl.total_bits);
    // IPX_BSORT_LEAF_AVG: records per leaf aimed at (tests: a huge value makes one leaf of everything -> overflow)
    const char *avg_e = getenv("IPX_BSORT_LEAF_AVG");
    const uint64_t leaf_avg = avg_e && atoll(avg_e) > 0 ? (uint64_t)atoll(avg_e) : BS_LEAF_AVG;
    const uint64_t k_target = std::min<uint64_t>(std::max<uint64_t>(1, ipx_div_up(n, leaf_avg)), BS_MAX_LEAVES);
    const uint32_t m2 = (uint32_t)std::min<uint64_t>(n, k_target * BS_OVERSAMPLE);
    // coarse ranges: ~600 second-level sample keys each, i.e. ~37 leaves -- a range's table holds 255 splitters, a fenced
    // key takes two of them, and what does not fit is DROPPED (the range's last leaf then takes the rest and overflows).
    // With ~1800 keys a range (rounds 3-5) the ranges over the hot junctions of 10 M unfolded short reads -- 1 900 keys
    // with 500 to 16 000 copies each, neighbours in key order -- needed 180-212 of the 255 in every sort and more than
    // 255 in one sort of eight (a model of the sampling on S's keys; on the device: every few steps a fallback to the
    // radix passes and fifteen sorts without the bucket sort).  At ~600 the same ranges need 55-90.  From 13 M records
    // on there are 256 ranges either way (L, config 5's share: unchanged)
    uint32_t k1 = (uint32_t)std::max<uint64_t>(ipx_div_up(m2, 600), ipx_div_up(k_target, 170));
    k1 = std::min<uint32_t>(std::max<uint32_t>(k1, 1u), 256u);
    const uint32_t m1 = std::min<uint32_t>(std::max<uint32_t>(16u * k1, 256u), BS_CAP);     // 16 first-level sample keys per coarse range
    hipLaunchKernelGGL(k_bs_sample1, dim3(1), dim3(BS_NT), 0, s, buf_a, n, sq, k1, m1, coarse, list_n, overflow);
    hipLaunchKernelGGL(k_bs_sample2, dim3((uint32_t)ipx_div_up(m2, 256)), dim3(256), 0, s, buf_a, n, m2, sq, k1, coarse, list_n, lists);
    hipLaunchKernelGGL(k_bs_sort2, dim3(k1), dim3(BS_NT), 0, s, list_n, lists, fine, fine_n);
    ipx_prof_mark(s, IPXK_BS_SAMPLE);
    const uint32_t assign_blocks = (uint32_t)std::min<uint64_t>(ipx_div_up(n, 256 * 4), (uint64_t)n_cus * 16);
    hipLaunchKernelGGL(k_bs_assign, dim3(assign_blocks), dim3(256), 0, s, buf_a, dig_a, n, sq, k1, coarse, fine, fine_n, n_leaves);
    hipError_t e = hipMemsetAsync(start, 0xff, (size_t)65540 * 4, s);
    if (e != hipSuccess) return e;
    ipx_prof_mark(s, IPXK_BS_ASSIGN);
    SortPlan plan{};
    plan.n = 2;
    plan.digits[0] = DigitSpec{BS_ID_LO, 8, 0, 0};
    plan.digits[1] = DigitSpec{BS_ID_LO + 8, 8, 0, 0};
    const DigitSpec id_low{BS_ID_LO, 8, 0, 0};
    if ((e = ipx_launch_sort(s, buf_a, buf_b, dig_a, dig_b, true, n, plan, sort_ws, sort_ws_bytes, -1, &id_low)) != hipSuccess) return e;
    // two passes: the records are back in buf_a, the low id bytes the second pass wrote in dig_a
    hipLaunchKernelGGL(k_bs_bounds, dim3((uint32_t)ipx_div_up(ipx_div_up(n, 16), 256)), dim3(256), 0, s, buf_a, dig_a, n, n_leaves, start);
    ipx_prof_mark(s, IPXK_BS_BOUNDS);
    const uint32_t leaf_blocks = (uint32_t)std::min<uint64_t>(k_target + k1, (uint64_t)n_cus * 16);
    // (n_leaves <= 65536: 256 ranges of <= 256 leaves)
    hipLaunchKernelGGL(k_bs_big_leaves, dim3(256), dim3(256), 0, s, start, n_leaves, BS_CAP / 2, big_list);
    hipLaunchKernelGGL(k_bs_leaf<BS_IPT / 2>, dim3(leaf_blocks), dim3(BS_NT), 0, s, buf_a, start, n_leaves, 0u, 0u, kl.total_bits, overflow, nullptr);
    const uint32_t big_blocks = (uint32_t)std::min<uint64_t>(k_target / 8 + 1, (uint64_t)n_cus * 2);
    hipLaunchKernelGGL(k_bs_leaf<BS_IPT>, dim3(big_blocks), dim3(BS_NT), 0, s, buf_a, start, n_leaves, 0u, BS_CAP / 2, kl.total_bits, overflow, big_list);
    ipx_prof_mark(s, IPXK_BS_LEAF);
    return hipGetLastError();
}
