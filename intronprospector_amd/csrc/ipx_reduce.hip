// ipx_reduce.hip -- segmented reduce of the sorted junction records into the
// junction table, with the order-dependent semantics of the reference's
// sequential map insert/update restated as order-independent reductions:
//
//   add_junction      junctions_extractor.cc:488-506   new key: filter decides;
//                     a record that fails leaves no trace and the next record of
//                     the key is tried with ITS strand
//   get_splice_sites  junctions_extractor.cc:405-428   (device 2-bit genome)
//   correct_strand    junctions_extractor.cc:431-444
//   junction_filter_check  splice_juncs.cc:52-61
//   create_junction   junctions_extractor.cc:446-464   creator fixes strand/motif
//   update_junction   junctions_extractor.cc:466-485   min/max anchors, count_read
//   Junction::count_read junctions.hh:163-167          counts by category, offsets
//
// For one key, with records in input order r0 < r1 < ...: the creator is the
// first record whose strand passes the filter against the key's motif; records
// before it are dropped, every record from it on is counted.  So
//   phase A: creator[key] = min order over passing records (segmented min)
//   phase B: tallies over records with order >= creator[key] (segmented sums)
// Records arrive sorted by (key, off16): a key's records are one contiguous
// segment and, inside it, equal offsets are contiguous runs (histogram bins).
//
// Work split: one wavefront per chunk of 512 consecutive sorted records, 8
// coalesced rounds of 64; segment ids come from per-chunk head counts that were
// prefix-summed once (k_count_heads + scan), so no per-record id array is stored.
#include "ipx_common.h"
#include "ipx_reduce.h"
#include <algorithm>
#include <cstdlib>

namespace {

constexpr uint32_t CHUNK = 512;
constexpr uint32_t ROUNDS = CHUNK / 64;
constexpr unsigned long long NONE64 = 0xffffffffffffffffull;

struct Heads {
    bool ok, head_j, head_b;
    bool desc;          // the item's key is SMALLER than its predecessor's: the items are not sorted
    uint4 it;
};

// flags of item i (lane-parallel inside one round)
__device__ __forceinline__ Heads load_heads(const uint4 *__restrict__ items, uint64_t i, uint64_t n,
                                            uint32_t lane) {
    Heads h;
    h.ok = i < n;
    // the item and its predecessor as two independent loads (the second hits the lines of the first): taking the
    // predecessor from the lane below and letting lane 0 fetch its own afterwards put two load latencies in a row
    // (both from clamped addresses, for every lane, then selected: a load behind a branch is waited for at the
    // branch's end, and two of them in a row are two round trips)
    const bool has_prev = h.ok && i > 0;
    const uint4 it_raw = items[h.ok ? i : (n ? n - 1 : 0)];      // (n == 0: the launcher does not launch; the index stays in range anyway)
    const uint4 p_raw = items[has_prev ? i - 1 : 0];
    h.it = h.ok ? it_raw : make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
    const uint4 p = has_prev ? p_raw : make_uint4(0, 0, 0, 0);
    const uint32_t px = p.x, py = p.y, pz = p.z;
    (void)lane;
    const bool first = (i == 0);
    const bool dj = ((px >> 16) != (h.it.x >> 16)) || (py != h.it.y) || (pz != h.it.z);
    const bool db = dj || ((px & 0xffffu) != (h.it.x & 0xffffu));
    h.head_j = h.ok && (first || dj);
    h.head_b = h.ok && (first || db);
    h.desc = has_prev && (pz > h.it.z || (pz == h.it.z && (py > h.it.y || (py == h.it.y && px > h.it.x))));
    return h;
}

// unsorted: |= 1 when two neighbouring items are out of order.  Every sort of ipx_finalize ends here, so whatever one of
// them rests on -- the bucket sort's leaf ranks come from the order in which LDS atomics of one instruction return
// (ipx_bsort.hip) -- a table is never reduced from items that are not sorted: ipx_finalize sorts them again by radix
// passes, and fails if those leave them unsorted too.
__global__ __launch_bounds__(256) void k_count_heads(const uint4 *__restrict__ items,
                                                     uint64_t n, uint32_t *__restrict__ chunk_j,
                                                     uint32_t *__restrict__ chunk_b, uint32_t *__restrict__ unsorted) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t chunk = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t base = chunk * CHUNK;
    if (base >= n) return;
    uint32_t cj = 0, cb = 0;
    bool desc = false;
#pragma unroll
    for (uint32_t r = 0; r < ROUNDS; ++r) {
        const Heads h = load_heads(items, base + r * 64 + lane, n, lane);
        cj += (uint32_t)__popcll(ipx_ballot(h.head_j));
        cb += (uint32_t)__popcll(ipx_ballot(h.head_b));
        desc = desc || h.desc;
    }
    if (lane == 0) { chunk_j[chunk] = cj; chunk_b[chunk] = cb; }
    if (ipx_ballot(desc) && lane == 0) atomicOr(unsorted, 1u);
}

// junction / bin descriptors at the segment heads
__global__ __launch_bounds__(256) void k_write_heads(const uint4 *__restrict__ items,
                                                     uint64_t n, KeyLayout kl,
                                                     const uint32_t *__restrict__ chunk_j,
                                                     const uint32_t *__restrict__ chunk_b,
                                                     uint32_t *__restrict__ jrank,
                                                     uint32_t *__restrict__ jstart,
                                                     uint32_t *__restrict__ jend,
                                                     uint32_t *__restrict__ jfirstbin,
                                                     uint32_t *__restrict__ bin_off,
                                                     unsigned long long *__restrict__ first,
                                                     unsigned long long *__restrict__ creator,
                                                     unsigned long long *__restrict__ counts,
                                                     uint32_t *__restrict__ amin, uint32_t *__restrict__ amax,
                                                     uint32_t *__restrict__ bin_count, uint32_t *__restrict__ totals,
                                                     uint32_t *__restrict__ jmotif0, uint32_t n_j, uint32_t n_b,
                                                     uint16_t *__restrict__ hist_value0) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t lane_le = ~0ull >> (63 - lane);
    const uint64_t chunk = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t base = chunk * CHUNK;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        totals[2] = 0; totals[3] = 0; totals[4] = 0; totals[5] = ~0u; totals[6] = ~0u;
        // without a genome nothing is dropped later (jmotif0 != null): the table has all n_j junctions and n_b bins
        if (jmotif0) { totals[0] = n_j; totals[1] = n_b; }
    }
    if (base >= n) return;
    uint32_t nj = chunk_j[chunk], nb = chunk_b[chunk];     // heads before this chunk
#pragma unroll
    for (uint32_t r = 0; r < ROUNDS; ++r) {
        const Heads h = load_heads(items, base + r * 64 + lane, n, lane);
        const uint64_t mj = ipx_ballot(h.head_j), mb = ipx_ballot(h.head_b);
        const uint32_t jid = nj + (uint32_t)__popcll(mj & lane_le) - 1;
        const uint32_t bid = nb + (uint32_t)__popcll(mb & lane_le) - 1;
        if (h.head_b) {
            const KeyParts k = unpack_key(h.it, kl);
            if (hist_value0) hist_value0[bid] = (uint16_t)k.off16;      // straight into the table (no genome: bins keep their places)
            else bin_off[bid] = k.off16;
            bin_count[bid] = 0;                  // the rows k_reduce accumulates into start at the identity
            if (h.head_j) {
                jrank[jid] = k.rank; jstart[jid] = k.start; jend[jid] = k.end; jfirstbin[jid] = bid;
                if (jmotif0) jmotif0[jid] = 0;           // what k_motif would write without a genome
                first[jid] = ~0ull; creator[jid] = ~0ull;
                amin[jid] = ~0u; amax[jid] = 0u;
                counts[3ull * jid] = 0; counts[3ull * jid + 1] = 0; counts[3ull * jid + 2] = 0;
            }
        }
        nj += (uint32_t)__popcll(mj);
        nb += (uint32_t)__popcll(mb);
    }
}

// jmotif: IPX_MOTIF_* bits | bases; error marker in bits 11..8:
//   0x100 = sequence missing and not skippable, 0x200 = fetch ran past the end
__global__ __launch_bounds__(256) void k_motif(uint32_t n_j, GenomeDev g, int32_t skip_missing,
                                               const uint32_t *__restrict__ jrank,
                                               const uint32_t *__restrict__ jstart,
                                               const uint32_t *__restrict__ jend,
                                               uint32_t *__restrict__ jmotif) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_j) return;
    const uint32_t m = ipx_motif_of(g, skip_missing, jrank[j], jstart[j], jend[j]);
    jmotif[j] = m;
}

// Without a genome k_reduce<0> writes the segment descriptors itself (see there); what is left of k_write_heads
// is setting the rows it accumulates into to the identity: one pass over J + B words instead of over the records.
__global__ __launch_bounds__(256) void k_init_rows(uint32_t n_j, uint32_t n_b, unsigned long long *__restrict__ first,
                                                   unsigned long long *__restrict__ creator,
                                                   unsigned long long *__restrict__ counts, uint32_t *__restrict__ amin,
                                                   uint32_t *__restrict__ amax, uint32_t *__restrict__ bin_count,
                                                   uint32_t *__restrict__ totals) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x, t0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t0 == 0) {
        totals[0] = n_j; totals[1] = n_b;          // nothing is dropped without a genome
        totals[2] = 0; totals[3] = 0; totals[4] = 0; totals[5] = ~0u; totals[6] = ~0u;
    }
    for (uint64_t j = t0; j < n_j; j += stride) {
        first[j] = ~0ull; creator[j] = ~0ull;
        amin[j] = ~0u; amax[j] = 0u;
        counts[3 * j] = 0; counts[3 * j + 1] = 0; counts[3 * j + 2] = 0;
    }
    for (uint64_t b = t0; b < n_b; b += stride) bin_count[b] = 0;
}

// (Barriers, round-4 audit: k_reduce, k_count_heads, k_write_heads have none -- a wave works its chunk alone, and the
// early `return` of a wave whose chunk lies beyond the items leaves no barrier behind.  The two block barriers of this file,
// in k_finish_junctions and k_rank_* further down, are unconditional: the grid-stride loops before them hold none.)
// ---- wave-segmented reductions -------------------------------------------------
// seg ids are non-decreasing over the lanes; the result is valid at head lanes.
__device__ __forceinline__ uint32_t seg_last_lane(uint64_t heads, uint32_t lane) {
    const uint64_t above = (lane == 63) ? 0ull : (heads & (~0ull << (lane + 1)));
    return above ? (uint32_t)(__ffsll((unsigned long long)above) - 2) : 63u;
}

template <typename T, typename Op>
__device__ __forceinline__ T seg_reduce(T v, uint32_t lane, uint32_t last, Op op) {
#pragma unroll
    for (uint32_t d = 1; d < 64; d <<= 1) {
        const T o = __shfl_down(v, d);
        if (lane + d <= last) v = op(v, o);
    }
    return v;
}

struct MinU64 { __device__ unsigned long long operator()(unsigned long long a, unsigned long long b) const { return a < b ? a : b; } };
struct MinU32 { __device__ uint32_t operator()(uint32_t a, uint32_t b) const { return a < b ? a : b; } };
struct MaxU32 { __device__ uint32_t operator()(uint32_t a, uint32_t b) const { return a > b ? a : b; } };
struct AddU32 { __device__ uint32_t operator()(uint32_t a, uint32_t b) const { return a + b; } };
struct AddU64 { __device__ unsigned long long operator()(unsigned long long a, unsigned long long b) const { return a + b; } };

// ---- the reduce kernel ---------------------------------------------------------------
// MODE 0  no record can fail the filter: creator = first = min order, everything counted
//         (one pass: phases A and B fused)
// MODE 1  phase A with an active filter: first[j], creator[j] (min over passing records)
// MODE 2  phase B with an active filter: tallies over records with order >= creator[j]
//
// A wave walks its 512-record chunk in 8 rounds.  Per round the junction segments (and the
// (junction, offset) bins) are reduced with wave-segmented shuffles.  A segment that runs
// on into the next round is CARRIED in wave-uniform registers instead of being flushed,
// so a junction costs one flush per chunk, not per round (a junction with 10^6 records
// otherwise serialises ~10^5 same-address atomics).  A segment that begins and ends inside
// the chunk is flushed with plain stores: no other wave ever touches its row.  Only
// segments that cross a chunk boundary use atomics (rows are pre-set to the identity).
struct JAcc {
    unsigned long long m1, m2;      // min order over all / over passing records
    uint32_t c0, c1, c2, lo, hi;    // counts by category, min anchor_start, max anchor_end
};

template <int MODE>
__device__ __forceinline__ void flush_junction(bool plain, uint32_t jid, const JAcc &a,
                                               unsigned long long *__restrict__ first,
                                               unsigned long long *__restrict__ creator,
                                               unsigned long long *__restrict__ counts,
                                               uint32_t *__restrict__ amin, uint32_t *__restrict__ amax) {
    if (MODE == 0 || MODE == 1) {
        if (plain) {
            first[jid] = a.m1;
            creator[jid] = (MODE == 0) ? a.m1 : a.m2;
        } else {
            atomicMin(&first[jid], a.m1);
            if (MODE == 0) atomicMin(&creator[jid], a.m1);
            else if (a.m2 != NONE64) atomicMin(&creator[jid], a.m2);
        }
    }
    if (MODE == 0 || MODE == 2) {
        if (plain) {
            counts[(uint64_t)jid * 3 + 0] = a.c0;
            counts[(uint64_t)jid * 3 + 1] = a.c1;
            counts[(uint64_t)jid * 3 + 2] = a.c2;
            amin[jid] = a.lo;
            amax[jid] = a.hi;
        } else if ((a.c0 | a.c1 | a.c2) || a.lo != 0xffffffffu) {
            if (a.c0) atomicAdd(&counts[(uint64_t)jid * 3 + 0], (unsigned long long)a.c0);
            if (a.c1) atomicAdd(&counts[(uint64_t)jid * 3 + 1], (unsigned long long)a.c1);
            if (a.c2) atomicAdd(&counts[(uint64_t)jid * 3 + 2], (unsigned long long)a.c2);
            atomicMin(&amin[jid], a.lo);
            atomicMax(&amax[jid], a.hi);
        }
    }
}

// MODE 0 never looks at m2 (its creator is m1): the four registers it would take carried around are worth a wave
template <int MODE>
__device__ __forceinline__ JAcc merge_acc(const JAcc &x, const JAcc &y) {
    JAcc r;
    r.m1 = x.m1 < y.m1 ? x.m1 : y.m1;
    r.m2 = (MODE == 0) ? NONE64 : (x.m2 < y.m2 ? x.m2 : y.m2);
    r.c0 = x.c0 + y.c0; r.c1 = x.c1 + y.c1; r.c2 = x.c2 + y.c2;
    r.lo = x.lo < y.lo ? x.lo : y.lo;
    r.hi = x.hi > y.hi ? x.hi : y.hi;
    return r;
}

template <int MODE>
__device__ __forceinline__ JAcc bcast_acc(const JAcc &x, int src) {
    JAcc r;
    r.m1 = __shfl(x.m1, src); r.m2 = (MODE == 0) ? NONE64 : __shfl(x.m2, src);
    r.c0 = __shfl(x.c0, src); r.c1 = __shfl(x.c1, src); r.c2 = __shfl(x.c2, src);
    r.lo = __shfl(x.lo, src); r.hi = __shfl(x.hi, src);
    return r;
}

// where k_reduce<0> leaves the segment descriptors when no genome is set (jrank == nullptr: k_write_heads did it)
struct HeadOut {
    uint32_t *jrank, *jstart, *jend, *jfirstbin, *jmotif;
    uint16_t *hist_value;
    KeyLayout kl;
};

// HAS_CNTS false: no record of the table stands for several (nothing combined, nothing adopted with counts), the count
// word is IPX_CNT_ONE of the record's category and cnts is not read: one random 128-byte line per record less.  HAS_CNTS
// true: the same for every record whose item does not carry IPX_ITEM_FOLDED (round 5)
template <int MODE, bool HAS_CNTS>
__global__ __launch_bounds__(256, 8) void k_reduce(const uint4 *__restrict__ items, uint64_t n,
                                                const uint4 *__restrict__ pays,
                                                const uint32_t *__restrict__ cnts,
                                                const uint32_t *__restrict__ chunk_j,
                                                const uint32_t *__restrict__ chunk_b,
                                                const uint32_t *__restrict__ jmotif, int32_t sj_filter,
                                                unsigned long long *__restrict__ first,
                                                unsigned long long *__restrict__ creator,
                                                unsigned long long *__restrict__ counts,
                                                uint32_t *__restrict__ amin, uint32_t *__restrict__ amax,
                                                uint32_t *__restrict__ bin_count, Counters *counters, HeadOut ho) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t lane_le = ~0ull >> (63 - lane);
    const uint64_t chunk = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t base = chunk * CHUNK;
    if (base >= n) return;
    uint32_t nj = chunk_j[chunk], nb = (MODE != 1) ? chunk_b[chunk] : 0;

    // carried (unfinished) junction segment and bin; wave-uniform
    bool cj_valid = false, cj_here = false;
    uint32_t cj_id = 0;
    JAcc cj{NONE64, NONE64, 0, 0, 0, 0xffffffffu, 0};
    bool cb_valid = false, cb_here = false;
    uint32_t cb_id = 0, cb_cnt = 0;
    uint4 last_item = make_uint4(0, 0, 0, 0);
    uint64_t last_index = base;

#pragma unroll 1
    for (uint32_t r = 0; r < ROUNDS; ++r) {
        const uint64_t i = base + r * 64 + lane;
        const Heads h = load_heads(items, i, n, lane);
        const uint64_t ok_mask = ipx_ballot(h.ok);
        if (!ok_mask) break;
        const uint64_t mj = ipx_ballot(h.head_j), mb = ipx_ballot(h.head_b);
        const uint32_t jid = nj + (uint32_t)__popcll(mj & lane_le) - 1;
        const uint32_t bid = nb + (uint32_t)__popcll(mb & lane_le) - 1;
        nj += (uint32_t)__popcll(mj);
        nb += (uint32_t)__popcll(mb);
        if (MODE == 0 && ho.jrank && h.head_b) {
            // segment descriptors (k_write_heads' job when a genome needs them before this kernel)
            const KeyParts k = unpack_key(h.it, ho.kl);
            ho.hist_value[bid] = (uint16_t)k.off16;
            if (h.head_j) {
                ho.jrank[jid] = k.rank; ho.jstart[jid] = k.start; ho.jend[jid] = k.end; ho.jfirstbin[jid] = bid;
                ho.jmotif[jid] = 0;
            }
        }
        const int n_ok = __popcll(ok_mask);                 // valid lanes are 0..n_ok-1
        last_item = make_uint4(__shfl(h.it.x, n_ok - 1), __shfl(h.it.y, n_ok - 1), __shfl(h.it.z, n_ok - 1), 0);
        last_index = base + r * 64 + (uint64_t)(n_ok - 1);

        JAcc v{NONE64, NONE64, 0, 0, 0, 0xffffffffu, 0};
        uint32_t counted = 0;
        unsigned long long poison = NONE64;
        if (h.ok) {
            const uint32_t ri = IPX_ITEM_INDEX(h.it.w);
            const uint4 p = pays[ri];
            // reads per category this record stands for: one, unless the scan folded copies of it into it -- the item says
            // so (IPX_ITEM_FOLDED), and only then is the count word worth its random line
            const uint32_t cw = (MODE != 1) ? ((HAS_CNTS && (h.it.w & IPX_ITEM_FOLDED)) ? cnts[ri] : IPX_CNT_ONE(p.x & 3u)) : 0u;
            const unsigned long long ord = (unsigned long long)p.x | ((unsigned long long)p.y << 32);
            bool cnt_it = true;
            if (MODE == 1) {
                v.m1 = ord;
                v.m2 = ipx_record_passes(IPX_ORD_STRAND(ord), jmotif[jid], sj_filter) ? ord : NONE64;
                cnt_it = false;
            } else if (MODE == 2) {
                const unsigned long long cr = creator[jid];
                cnt_it = (cr != NONE64) && IPX_ORD_SEQ(ord) >= IPX_ORD_SEQ(cr);
            } else {
                v.m1 = ord;
                v.m2 = NONE64;            // MODE 0: the creator is m1
            }
            if (cnt_it) {
                const uint32_t cat = IPX_ORD_CAT(ord);
                if (cat == 3) poison = (ord & ~0x3ffull) | (IPX_ERRK_CATEGORY << 8);   // junctions_extractor.cc:73-74
                v.c0 = cw & IPX_CNT_MASK; v.c1 = (cw >> IPX_CNT_BITS) & IPX_CNT_MASK; v.c2 = cw >> (2 * IPX_CNT_BITS);
                counted = v.c0 + v.c1 + v.c2;
                v.lo = p.z;
                v.hi = p.w;
            }
        }
        // lanes past the end form their own segment; lane 0 always opens one
        const uint64_t tail = ~ok_mask & (ok_mask << 1);
        const uint64_t heads_j = mj | 1ull | tail;
        const uint32_t last_j = seg_last_lane(heads_j, lane);
        if (MODE != 2) {
            v.m1 = seg_reduce(v.m1, lane, last_j, MinU64());
            if (MODE != 0) v.m2 = seg_reduce(v.m2, lane, last_j, MinU64());          // MODE 0: first == creator, m2 is dead
        }
        if (MODE != 1) {
            // <= 64 * 1023 per field and round
            unsigned long long cpk = (unsigned long long)v.c0 | ((unsigned long long)v.c1 << 16) | ((unsigned long long)v.c2 << 32);
            cpk = seg_reduce(cpk, lane, last_j, AddU64());
            v.c0 = (uint32_t)cpk & 0xffffu; v.c1 = (uint32_t)(cpk >> 16) & 0xffffu; v.c2 = (uint32_t)(cpk >> 32);
            v.lo = seg_reduce(v.lo, lane, last_j, MinU32());
            v.hi = seg_reduce(v.hi, lane, last_j, MaxU32());
        }

        // ---- junction segments: carry in, flush complete ones, carry out -------------------
        {
            const bool cont0 = !(mj & 1ull);                 // lane 0 continues a segment begun earlier
            const uint32_t jid0 = __shfl(jid, 0);
            bool here0 = false;                              // did lane 0's segment begin inside this chunk?
            if (cj_valid) {
                if (cont0 && cj_id == jid0) {
                    if (lane == 0) v = merge_acc<MODE>(v, cj);
                    here0 = cj_here;
                } else if (lane == 0) {
                    flush_junction<MODE>(cj_here, cj_id, cj, first, creator, counts, amin, amax);
                }
                cj_valid = false;
            }
            const bool is_head = h.ok && ((heads_j >> lane) & 1ull);
            const bool runs_on = (last_j == 63) && ((ok_mask >> 63) & 1ull);   // may continue in the next round
            const bool began_here = (lane == 0 && cont0) ? here0 : true;
            if (is_head && !runs_on) flush_junction<MODE>(began_here, jid, v, first, creator, counts, amin, amax);
            const uint64_t carry_mask = ipx_ballot(is_head && runs_on);
            if (carry_mask) {
                const int src = __ffsll((unsigned long long)carry_mask) - 1;
                cj = bcast_acc<MODE>(v, src);
                cj_id = __shfl(jid, src);
                cj_here = __shfl((int)began_here, src) != 0;
                cj_valid = true;
            }
        }
        // ---- bins ---------------------------------------------------------------------------
        if (MODE != 1) {
            const uint64_t heads_b = mb | 1ull | tail;
            const uint32_t last_b = seg_last_lane(heads_b, lane);
            uint32_t s_bin = seg_reduce(counted, lane, last_b, AddU32());
            const bool cont0 = !(mb & 1ull);
            const uint32_t bid0 = __shfl(bid, 0);
            bool here0 = false;
            if (cb_valid) {
                if (cont0 && cb_id == bid0) {
                    if (lane == 0) s_bin += cb_cnt;
                    here0 = cb_here;
                } else if (lane == 0) {
                    if (cb_here) bin_count[cb_id] = cb_cnt;
                    else if (cb_cnt) atomicAdd(&bin_count[cb_id], cb_cnt);
                }
                cb_valid = false;
            }
            const bool is_head = h.ok && ((heads_b >> lane) & 1ull);
            const bool runs_on = (last_b == 63) && ((ok_mask >> 63) & 1ull);
            const bool began_here = (lane == 0 && cont0) ? here0 : true;
            if (is_head && !runs_on) {
                if (began_here) bin_count[bid] = s_bin;
                else if (s_bin) atomicAdd(&bin_count[bid], s_bin);
            }
            const uint64_t carry_mask = ipx_ballot(is_head && runs_on);
            if (carry_mask) {
                const int src = __ffsll((unsigned long long)carry_mask) - 1;
                cb_cnt = __shfl(s_bin, src);
                cb_id = __shfl(bid, src);
                cb_here = __shfl((int)began_here, src) != 0;
                cb_valid = true;
            }
            // rare: a counted record whose category lookup throws
            unsigned long long e = poison;
            for (int d = 32; d >= 1; d >>= 1) {
                const unsigned long long o = __shfl_xor(e, d);
                e = o < e ? o : e;
            }
            if (lane == 0 && e != NONE64) atomicMin(&counters->err_word, e);
        }
    }
    // what is still carried ends with the chunk: complete only if the next record has another key
    if (cj_valid || cb_valid) {
        bool same_j = false, same_b = false;
        if (last_index + 1 < n) {
            const uint4 nx = items[last_index + 1];
            same_j = ((nx.x >> 16) == (last_item.x >> 16)) && nx.y == last_item.y && nx.z == last_item.z;
            same_b = same_j && ((nx.x & 0xffffu) == (last_item.x & 0xffffu));
        }
        if (lane == 0) {
            if (cj_valid) flush_junction<MODE>(cj_here && !same_j, cj_id, cj, first, creator, counts, amin, amax);
            if (MODE != 1 && cb_valid) {
                if (cb_here && !same_b) bin_count[cb_id] = cb_cnt;
                else if (cb_cnt) atomicAdd(&bin_count[cb_id], cb_cnt);
            }
        }
    }
}

// per junction: validity, final strand, genome errors.  Grid-stride with few blocks: the OR of the
// creators' ordinals leaves each block as ONE atomic (same-address atomics serialise at ~17 ns).
__global__ __launch_bounds__(1024) void k_finish_junctions(uint32_t n_j,
                                                          const unsigned long long *__restrict__ first,
                                                          const unsigned long long *__restrict__ creator,
                                                          const uint32_t *__restrict__ jmotif,
                                                          uint32_t *__restrict__ jvalid,
                                                          uint8_t *__restrict__ jstrand,
                                                          Counters *counters, uint32_t *__restrict__ ord_or) {
    __shared__ uint32_t or_s[16];
    uint32_t o = 0;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n_j; j += gridDim.x * blockDim.x) {
        const uint32_t m = jmotif[j];
        if (m & 0x300u) {
            // the first record of the key is the one whose fetch throws
            const unsigned long long f = first[j];
            atomicMin(&counters->err_word,
                      (f & ~0x3ffull) | (IPX_ERRK_GENOME << 8) | ((m & 0x200u) ? 1u : 0u));
        }
        const unsigned long long cr = creator[j];
        const bool valid = (cr != NONE64) && !(m & 0x300u);
        jvalid[j] = valid ? 1u : 0u;
        uint32_t s = '.';
        if (valid) {
            o |= (uint32_t)(cr >> IPX_ORD_SHIFT_ORDINAL) & IPX_ORD_MAX_ORDINAL;
            s = IPX_ORD_STRAND(cr);
            if ((m & IPX_MOTIF_PRESENT) && s == '.') {          // correct_strand :431-444
                if (m & IPX_MOTIF_CANON_P) s = '+';
                else if (m & IPX_MOTIF_CANON_RC) s = '-';
            }
        }
        jstrand[j] = (uint8_t)s;
    }
    for (int d = 32; d >= 1; d >>= 1) o |= __shfl_xor(o, d);
    if ((threadIdx.x & 63u) == 0) or_s[threadIdx.x >> 6] = o;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t all = 0;
        for (uint32_t w = 0; w < (blockDim.x >> 6); ++w) all |= or_s[w];
        if (all) atomicOr(ord_or, all);
    }
}

// non-empty bins: flag for compaction
__global__ __launch_bounds__(256) void k_bins(uint32_t n_b, const uint32_t *__restrict__ bin_count,
                                              uint32_t *__restrict__ bin_flag) {
    const uint32_t bidx = blockIdx.x * blockDim.x + threadIdx.x;
    if (bidx >= n_b) return;
    bin_flag[bidx] = bin_count[bidx] ? 1u : 0u;
}

__global__ __launch_bounds__(256) void k_emit_bins(uint32_t n_b, const uint32_t *__restrict__ bin_count,
                                                   const uint32_t *__restrict__ bin_off,
                                                   const uint32_t *__restrict__ bin_pos,
                                                   uint16_t *__restrict__ hist_value,
                                                   uint32_t *__restrict__ hist_count, uint32_t identity) {
    const uint32_t bidx = blockIdx.x * blockDim.x + threadIdx.x;
    if (bidx >= n_b) return;
    const uint32_t c = bin_count[bidx];
    if (c) {
        const uint32_t p = identity ? bidx : bin_pos[bidx];
        hist_value[p] = (uint16_t)bin_off[bidx];
        hist_count[p] = c;
    }
}

struct TableDev {
    uint32_t *target, *intron_start, *intron_end, *anchor_start, *anchor_end;
    uint8_t *strand;
    uint16_t *motif;
    uint64_t *read_counts, *creator_order, *hist_begin;
    uint32_t *ijunc;
};

__global__ __launch_bounds__(1024) void k_emit_junctions(
    uint32_t n_j, const uint32_t *__restrict__ jvalid, const uint32_t *__restrict__ jpos,
    const uint32_t *__restrict__ jrank, const uint32_t *__restrict__ jstart,
    const uint32_t *__restrict__ jend, const uint32_t *__restrict__ amin,
    const uint32_t *__restrict__ amax, const uint8_t *__restrict__ jstrand,
    const uint32_t *__restrict__ jmotif, const unsigned long long *__restrict__ counts,
    const unsigned long long *__restrict__ creator, const uint32_t *__restrict__ jfirstbin,
    const uint32_t *__restrict__ bin_pos,
    const uint32_t *__restrict__ tid_of_rank, const uint32_t *n_valid_total,
    const uint32_t *n_bins_total, TableDev out, uint4 *__restrict__ rank_items, uint32_t *__restrict__ rank_bits,
    uint32_t identity) {
    __shared__ uint32_t bits_s[16][4];
    if (blockIdx.x == 0 && threadIdx.x == 0) out.hist_begin[*n_valid_total] = *n_bins_total;
    // creation order packed as read_index << (bits of the largest ordinal) | ordinal; the OR / AND of those
    // keys (rank_bits[1..4]) tell the host which bytes vary, i.e. which radix passes the ranking needs
    const uint32_t ord_bits = 32u - (uint32_t)__clz((int)rank_bits[0]);
    uint32_t o0 = 0, o1 = 0, a0 = ~0u, a1 = ~0u;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n_j; j += gridDim.x * blockDim.x) {
        if (!identity && !jvalid[j]) continue;
        const uint32_t o = identity ? j : jpos[j];
        out.target[o] = tid_of_rank[jrank[j]];
        out.intron_start[o] = jstart[j];
        out.intron_end[o] = jend[j];
        out.anchor_start[o] = amin[j];
        out.anchor_end[o] = amax[j];
        out.strand[o] = jstrand[j];
        out.motif[o] = (uint16_t)(jmotif[j] & 0xf0ffu);
        out.read_counts[(uint64_t)o * 3 + 0] = counts[(uint64_t)j * 3 + 0];
        out.read_counts[(uint64_t)o * 3 + 1] = counts[(uint64_t)j * 3 + 1];
        out.read_counts[(uint64_t)o * 3 + 2] = counts[(uint64_t)j * 3 + 2];
        const unsigned long long cr = creator[j];
        out.creator_order[o] = IPX_ORD_SEQ(cr);
        out.hist_begin[o] = identity ? jfirstbin[j] : bin_pos[jfirstbin[j]];      // bins are ordered by (junction, offset)
        const unsigned long long rk = (IPX_ORD_READ(cr) << ord_bits) | ((cr >> IPX_ORD_SHIFT_ORDINAL) & IPX_ORD_MAX_ORDINAL);
        rank_items[o] = make_uint4((uint32_t)rk, (uint32_t)(rk >> 32), 0u, o);
        o0 |= (uint32_t)rk; o1 |= (uint32_t)(rk >> 32);
        a0 &= (uint32_t)rk; a1 &= (uint32_t)(rk >> 32);
    }
    for (int d = 32; d >= 1; d >>= 1) {
        o0 |= __shfl_xor(o0, d); o1 |= __shfl_xor(o1, d);
        a0 &= __shfl_xor(a0, d); a1 &= __shfl_xor(a1, d);
    }
    const uint32_t wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63u) == 0) { bits_s[wave][0] = o0; bits_s[wave][1] = o1; bits_s[wave][2] = a0; bits_s[wave][3] = a1; }
    __syncthreads();
    if (threadIdx.x < 4) {
        const uint32_t k = threadIdx.x, nw = blockDim.x >> 6;
        uint32_t v = k < 2 ? 0u : ~0u;
        for (uint32_t w = 0; w < nw; ++w) v = k < 2 ? (v | bits_s[w][k]) : (v & bits_s[w][k]);
        if (k < 2) { if (v) atomicOr(&rank_bits[1 + k], v); }
        else if (v != ~0u) atomicAnd(&rank_bits[1 + k], v);
    }
}

// Without a genome every junction is valid, keeps its creator's strand and its place (ipx_launch_reduce: `identity`):
// k_finish_junctions and k_emit_junctions are one pass over the junctions then.  The creation-order key packs the ordinal
// in its full twenty bits (k_emit_junctions packs it in as many bits as the table's largest ordinal has, which it learns
// from k_finish_junctions -- the one thing that kept the two apart; the rank's bitmap squeezes constant bits out anyway).
__global__ __launch_bounds__(1024) void k_finish_emit_identity(
    uint32_t n_j, const uint32_t *__restrict__ jrank, const uint32_t *__restrict__ jstart, const uint32_t *__restrict__ jend,
    const uint32_t *__restrict__ amin, const uint32_t *__restrict__ amax, const unsigned long long *__restrict__ counts,
    const unsigned long long *__restrict__ creator, const uint32_t *__restrict__ jfirstbin, const uint32_t *__restrict__ tid_of_rank,
    const uint32_t *n_valid_total, const uint32_t *n_bins_total, TableDev out, uint4 *__restrict__ rank_items,
    uint32_t *__restrict__ rank_bits, uint32_t *__restrict__ jvalid, uint8_t *__restrict__ jstrand) {
    __shared__ uint32_t bits_s[16][5];
    if (blockIdx.x == 0 && threadIdx.x == 0) out.hist_begin[*n_valid_total] = *n_bins_total;
    constexpr uint32_t ord_bits = 20;
    uint32_t o0 = 0, o1 = 0, a0 = ~0u, a1 = ~0u, oo = 0;
    // Four junctions a turn, everything they need loaded before the first store (the table's pointers are not `restrict`,
    // so the compiler keeps loads behind earlier stores: one junction a turn was nine dependent round trips a thread,
    // 13.7 us for S's 305 k junctions)
    constexpr uint32_t U = 4;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t j0 = blockIdx.x * blockDim.x + threadIdx.x; j0 < n_j; j0 += U * stride) {
        unsigned long long cr[U], c0[U], c1[U], c2[U];
        uint32_t rk_[U], st[U], en[U], lo[U], hi[U], fb[U], tg[U];
#pragma unroll
        for (uint32_t u = 0; u < U; ++u) {
            const uint32_t j = j0 + u * stride < n_j ? j0 + u * stride : j0;       // (clamped: loaded, not stored)
            cr[u] = creator[j];
            rk_[u] = jrank[j]; st[u] = jstart[j]; en[u] = jend[j]; lo[u] = amin[j]; hi[u] = amax[j]; fb[u] = jfirstbin[j];
            c0[u] = counts[(uint64_t)j * 3 + 0]; c1[u] = counts[(uint64_t)j * 3 + 1]; c2[u] = counts[(uint64_t)j * 3 + 2];
        }
#pragma unroll
        for (uint32_t u = 0; u < U; ++u) tg[u] = tid_of_rank[rk_[u]];
#pragma unroll
        for (uint32_t u = 0; u < U; ++u) {
            const uint32_t j = j0 + u * stride;
            if (j >= n_j) break;
            const uint32_t s = IPX_ORD_STRAND(cr[u]);
            jvalid[j] = 1u;
            jstrand[j] = (uint8_t)s;
            out.target[j] = tg[u];
            out.intron_start[j] = st[u];
            out.intron_end[j] = en[u];
            out.anchor_start[j] = lo[u];
            out.anchor_end[j] = hi[u];
            out.strand[j] = (uint8_t)s;
            out.motif[j] = 0;
            out.read_counts[(uint64_t)j * 3 + 0] = c0[u];
            out.read_counts[(uint64_t)j * 3 + 1] = c1[u];
            out.read_counts[(uint64_t)j * 3 + 2] = c2[u];
            out.creator_order[j] = IPX_ORD_SEQ(cr[u]);
            out.hist_begin[j] = fb[u];
            const uint32_t ordinal = (uint32_t)(cr[u] >> IPX_ORD_SHIFT_ORDINAL) & IPX_ORD_MAX_ORDINAL;
            const unsigned long long rk = (IPX_ORD_READ(cr[u]) << ord_bits) | ordinal;
            rank_items[j] = make_uint4((uint32_t)rk, (uint32_t)(rk >> 32), 0u, j);
            oo |= ordinal;
            o0 |= (uint32_t)rk; o1 |= (uint32_t)(rk >> 32);
            a0 &= (uint32_t)rk; a1 &= (uint32_t)(rk >> 32);
        }
    }
    for (int d = 32; d >= 1; d >>= 1) {
        o0 |= __shfl_xor(o0, d); o1 |= __shfl_xor(o1, d); oo |= __shfl_xor(oo, d);
        a0 &= __shfl_xor(a0, d); a1 &= __shfl_xor(a1, d);
    }
    const uint32_t wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63u) == 0) { bits_s[wave][0] = o0; bits_s[wave][1] = o1; bits_s[wave][2] = a0; bits_s[wave][3] = a1; bits_s[wave][4] = oo; }
    __syncthreads();
    if (threadIdx.x < 5) {
        const uint32_t k = threadIdx.x, nw = blockDim.x >> 6;
        const bool is_and = k == 2 || k == 3;
        uint32_t v = is_and ? ~0u : 0u;
        for (uint32_t w = 0; w < nw; ++w) v = is_and ? (v & bits_s[w][k]) : (v | bits_s[w][k]);
        if (k < 2) { if (v) atomicOr(&rank_bits[1 + k], v); }
        else if (k < 4) { if (v != ~0u) atomicAnd(&rank_bits[1 + k], v); }
        else if (v) atomicOr(&rank_bits[0], v);
    }
}

__global__ __launch_bounds__(256) void k_emit_ijunc(uint32_t n, const uint4 *__restrict__ items,
                                                    uint32_t *__restrict__ ijunc) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) ijunc[items[i].w] = i;
}

// ---- ijunc by bitmap ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t squeeze_rank_key(const uint4 &it, const RankBits &rb) {
    const uint64_t k = (uint64_t)it.x | ((uint64_t)it.y << 32);
    uint32_t out = 0, sh = 0;
#pragma unroll
    for (uint32_t r = 0; r < 4; ++r) {
        if (r < rb.n_runs) {
            out |= (uint32_t)((k >> rb.lo[r]) & ((1ull << rb.width[r]) - 1ull)) << sh;
            sh += rb.width[r];
        }
    }
    return out;
}

__global__ __launch_bounds__(256) void k_rank_mark(uint32_t n, const uint4 *__restrict__ items, RankBits rb,
                                                   uint32_t *__restrict__ bitmap) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const uint32_t k = squeeze_rank_key(items[i], rb);
        atomicOr(&bitmap[k >> 5], 1u << (k & 31u));
    }
}

// set bits below each word in two levels: inside its group of 64 words (one wave: before[w]) and per group
// (group_total[g], prefix-summed afterwards -- 64 times fewer elements than a scan over all words: one launch
// for a bitmap of 2^25 bits instead of three passes over a million counts)
__global__ __launch_bounds__(256) void k_rank_popc(uint32_t n_words, const uint32_t *__restrict__ bitmap,
                                                   uint32_t *__restrict__ before, uint32_t *__restrict__ group_total) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63u;
    const uint32_t c = i < n_words ? (uint32_t)__popc(bitmap[i]) : 0u;
    uint32_t incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t x = __shfl_up(incl, o);
        if ((int)lane >= o) incl += x;
    }
    if (i < n_words) before[i] = incl - c;
    if (lane == 63 && (i >> 6) < ((n_words + 63u) >> 6)) group_total[i >> 6] = incl;
}

__global__ __launch_bounds__(256) void k_rank_lookup(uint32_t n, const uint4 *__restrict__ items, RankBits rb,
                                                     const uint32_t *__restrict__ bitmap,
                                                     const uint32_t *__restrict__ before,
                                                     const uint32_t *__restrict__ group_before, uint32_t *__restrict__ ijunc) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const uint4 it = items[i];
        const uint32_t k = squeeze_rank_key(it, rb);
        ijunc[it.w] = group_before[k >> 11] + before[k >> 5] + (uint32_t)__popc(bitmap[k >> 5] & ((1u << (k & 31u)) - 1u));
    }
}

// after a genome error: which key did the failing record belong to?
__global__ __launch_bounds__(256) void k_find_genome_error(uint32_t n_j,
                                                           const unsigned long long *__restrict__ first,
                                                           const uint32_t *__restrict__ jmotif,
                                                           const uint32_t *__restrict__ jrank,
                                                           const uint32_t *__restrict__ jstart,
                                                           const uint32_t *__restrict__ jend,
                                                           unsigned long long err_seq, uint32_t *detail4) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_j) return;
    if ((jmotif[j] & 0x300u) && IPX_ORD_SEQ(first[j]) == err_seq) {
        detail4[0] = jrank[j]; detail4[1] = jstart[j]; detail4[2] = jend[j]; detail4[3] = 1;
    }
}

}  // namespace

// ---- launch wrappers (host side of this translation unit) ----------------------

static inline dim3 grid1d(uint64_t n, uint32_t per_block) { return dim3((uint32_t)ipx_div_up(n ? n : 1, per_block)); }

uint64_t ipx_reduce_chunks(uint64_t n) { return ipx_div_up(n, CHUNK); }

hipError_t ipx_launch_count_heads(hipStream_t s, const SortedItems &si, uint32_t *chunk_j, uint32_t *chunk_b, uint32_t *unsorted) {
    const uint64_t chunks = ipx_div_up(si.n, CHUNK);
    hipLaunchKernelGGL(k_count_heads, grid1d(chunks, 4), dim3(256), 0, s, si.items, si.n, chunk_j, chunk_b, unsorted);
    ipx_prof_mark(s, IPXK_COUNT_HEADS);
    return hipGetLastError();
}

hipError_t ipx_launch_reduce(hipStream_t s, const SortedItems &si, const uint4 *pays, const uint32_t *cnts, const KeyLayout &kl,
                             const ReduceBuffers &rb, const GenomeDev &genome, int32_t sj_filter,
                             int32_t skip_missing, const uint32_t *tid_of_rank, Counters *counters) {
    const uint64_t chunks = ipx_div_up(si.n, CHUNK);
    const uint32_t J = rb.n_j, B = rb.n_b;
    // Without a genome there are no motifs, no filter and no fetch errors: every junction has a creator (its
    // first record) and every bin a counted record, so nothing is dropped and the table keeps the order and
    // the count of the segments -- no motif kernel, no flags, no compaction scans.
    const uint32_t identity = genome.seq_of_rank == nullptr ? 1u : 0u;
    // ... and the bins' offsets and counts are written where the table has them (no k_emit_bins)
    uint32_t *bin_count = identity ? rb.out_hist_count : rb.bin_count;
    // (k_write_heads is the path of a run with a genome; without one the descriptors come from k_reduce<0>)
    const bool fuse_heads = identity;
    HeadOut ho{};
    if (fuse_heads) {
        const uint64_t cells = (uint64_t)J + B;
        hipLaunchKernelGGL(k_init_rows, dim3((uint32_t)std::min<uint64_t>(ipx_div_up(cells ? cells : 1, 1024), 4096)), dim3(256), 0, s, J, B,
                           rb.first, rb.creator, rb.counts, rb.amin, rb.amax, bin_count, rb.totals);
        ho = HeadOut{rb.jrank, rb.jstart, rb.jend, rb.jfirstbin, rb.jmotif, rb.out_hist_value, kl};
    } else {
        hipLaunchKernelGGL(k_write_heads, grid1d(chunks, 4), dim3(256), 0, s, si.items, si.n, kl, rb.chunk_j, rb.chunk_b, rb.jrank, rb.jstart, rb.jend,
                           rb.jfirstbin, rb.bin_off, rb.first, rb.creator, rb.counts, rb.amin, rb.amax, bin_count, rb.totals,
                           identity ? rb.jmotif : nullptr, J, B, identity ? rb.out_hist_value : nullptr);
    }
    ipx_prof_mark(s, IPXK_WRITE_HEADS);
    if (!identity) {
        hipLaunchKernelGGL(k_motif, grid1d(J, 256), dim3(256), 0, s, J, genome, skip_missing, rb.jrank,
                           rb.jstart, rb.jend, rb.jmotif);
        ipx_prof_mark(s, IPXK_MOTIF);
    }
    hipError_t e;
    // A record can fail the filter only against a motif and under "canon" (splice_juncs.cc:52-61)
    const bool filter_active = genome.seq_of_rank != nullptr && sj_filter == IPX_SJ_CANON;
    if (!filter_active) {
        ipx_prof_pre(s, IPXK_PHASE_B);
        if (cnts)
            hipLaunchKernelGGL((k_reduce<0, true>), grid1d(chunks, 4), dim3(256), 0, s, si.items, si.n, pays, cnts, rb.chunk_j, rb.chunk_b,
                               rb.jmotif, sj_filter, rb.first, rb.creator, rb.counts, rb.amin, rb.amax, bin_count, counters, ho);
        else
            hipLaunchKernelGGL((k_reduce<0, false>), grid1d(chunks, 4), dim3(256), 0, s, si.items, si.n, pays, cnts, rb.chunk_j, rb.chunk_b,
                               rb.jmotif, sj_filter, rb.first, rb.creator, rb.counts, rb.amin, rb.amax, bin_count, counters, ho);
        ipx_prof_mark(s, IPXK_PHASE_B);
    } else {
        hipLaunchKernelGGL((k_reduce<1, true>), grid1d(chunks, 4), dim3(256), 0, s, si.items, si.n, pays, cnts, rb.chunk_j, rb.chunk_b,
                           rb.jmotif, sj_filter, rb.first, rb.creator, rb.counts, rb.amin, rb.amax, rb.bin_count, counters, HeadOut{});
        ipx_prof_mark(s, IPXK_PHASE_A);
        if (cnts)
            hipLaunchKernelGGL((k_reduce<2, true>), grid1d(chunks, 4), dim3(256), 0, s, si.items, si.n, pays, cnts, rb.chunk_j, rb.chunk_b,
                               rb.jmotif, sj_filter, rb.first, rb.creator, rb.counts, rb.amin, rb.amax, rb.bin_count, counters, HeadOut{});
        else
            hipLaunchKernelGGL((k_reduce<2, false>), grid1d(chunks, 4), dim3(256), 0, s, si.items, si.n, pays, cnts, rb.chunk_j, rb.chunk_b,
                               rb.jmotif, sj_filter, rb.first, rb.creator, rb.counts, rb.amin, rb.amax, rb.bin_count, counters, HeadOut{});
        ipx_prof_mark(s, IPXK_PHASE_B);
    }
    TableDev td{rb.out_target, rb.out_intron_start, rb.out_intron_end, rb.out_anchor_start,
                rb.out_anchor_end, rb.out_strand, rb.out_motif, rb.out_read_counts,
                rb.out_creator_order, rb.out_hist_begin, rb.out_ijunc};
    if (identity) {
        // blocks of 1024 threads, four junctions a thread and turn; at most 256 blocks: each ends in five atomics on one line
        // (128 blocks of 256 threads took 150 us for the 3.9 M junctions of config 5's share: 119 junctions a thread, in turn)
        hipLaunchKernelGGL(k_finish_emit_identity, dim3(std::min<uint32_t>((uint32_t)((J + 4095u) / 4096u) + (J ? 0u : 1u), 256u)), dim3(1024), 0, s, J, rb.jrank,
                           rb.jstart, rb.jend, rb.amin, rb.amax, rb.counts, rb.creator, rb.jfirstbin, tid_of_rank, rb.totals + 0,
                           rb.totals + 1, td, rb.rank_a, rb.totals + 2, rb.jvalid, rb.jstrand);
        ipx_prof_mark(s, IPXK_EMIT_J);
        return hipGetLastError();
    }
    // (blocks of 1024 threads, a grid that grows with J up to 256 blocks -- every block ends in atomics on one line; 128 blocks
    // of 256 threads left a thread a hundred junctions, one after the other, in a table of millions)
    const dim3 j_grid(std::min<uint32_t>((uint32_t)((J + 4095u) / 4096u) + (J ? 0u : 1u), 256u));
    hipLaunchKernelGGL(k_finish_junctions, j_grid, dim3(1024), 0, s, J, rb.first, rb.creator,
                       rb.jmotif, rb.jvalid, rb.jstrand, counters, rb.totals + 2);
    ipx_prof_mark(s, IPXK_FINISH_J);
    if (!identity) {
        hipLaunchKernelGGL(k_bins, grid1d(B, 256), dim3(256), 0, s, B, rb.bin_count, rb.bin_flag);
        ipx_prof_mark(s, IPXK_BINS);
        // compaction offsets: junctions, bins, bins per junction
        if ((e = ipx_launch_scan_u32_pair(s, rb.jvalid, rb.jpos, J, rb.totals + 0, rb.bin_flag, rb.bin_pos, B, rb.totals + 1,
                                          rb.scan_sums)) != hipSuccess) return e;
    }
    if (!identity) {
        hipLaunchKernelGGL(k_emit_bins, grid1d(B, 256), dim3(256), 0, s, B, rb.bin_count, rb.bin_off,
                           rb.bin_pos, rb.out_hist_value, rb.out_hist_count, identity);
        ipx_prof_mark(s, IPXK_EMIT_BINS);
    }
    hipLaunchKernelGGL(k_emit_junctions, j_grid, dim3(1024), 0, s, J, rb.jvalid, rb.jpos, rb.jrank,
                       rb.jstart, rb.jend, rb.amin, rb.amax, rb.jstrand, rb.jmotif, rb.counts, rb.creator,
                       rb.jfirstbin, rb.bin_pos, tid_of_rank, rb.totals + 0, rb.totals + 1, td, rb.rank_a, rb.totals + 2, identity);
    ipx_prof_mark(s, IPXK_EMIT_J);
    return hipGetLastError();
}

hipError_t ipx_launch_emit_ijunc(hipStream_t s, uint32_t n_valid, const uint4 *__restrict__ items, uint32_t *ijunc) {
    if (n_valid == 0) return hipSuccess;
    hipLaunchKernelGGL(k_emit_ijunc, grid1d(n_valid, 256), dim3(256), 0, s, n_valid, items, ijunc);
    ipx_prof_mark(s, IPXK_EMIT_IJUNC);
    return hipGetLastError();
}

size_t ipx_rank_bitmap_workspace_bytes(uint32_t bits) {
    const uint64_t words = bits > 5 ? (1ull << (bits - 5)) : 1ull;
    const uint64_t groups = ipx_div_up(words, 64);
    return (size_t)((2 * words + groups + 64 + ipx_scan_sums_words(groups, 0)) * sizeof(uint32_t));
}

hipError_t ipx_launch_rank_by_bitmap(hipStream_t s, uint32_t n_valid, const uint4 *rank_items, const RankBits &rb,
                                     uint32_t *workspace, uint32_t *ijunc) {
    if (n_valid == 0) return hipSuccess;
    const uint32_t words = rb.bits > 5 ? (1u << (rb.bits - 5)) : 1u;
    const uint32_t groups = (words + 63u) / 64u;
    uint32_t *bitmap = workspace, *before = workspace + words, *group_before = before + words, *sums = group_before + groups + 64;
    hipError_t e;
    if ((e = hipMemsetAsync(bitmap, 0, (size_t)words * sizeof(uint32_t), s)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_rank_mark, grid1d(n_valid, 256), dim3(256), 0, s, n_valid, rank_items, rb, bitmap);
    hipLaunchKernelGGL(k_rank_popc, grid1d(words, 256), dim3(256), 0, s, words, bitmap, before, group_before);
    if ((e = ipx_launch_scan_u32(s, group_before, group_before, groups, sums, nullptr)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_rank_lookup, grid1d(n_valid, 256), dim3(256), 0, s, n_valid, rank_items, rb, bitmap, before, group_before, ijunc);
    ipx_prof_mark(s, IPXK_RANK_SORT);
    return hipGetLastError();
}

hipError_t ipx_launch_find_genome_error(hipStream_t s, uint32_t n_j, const unsigned long long *first,
                                        const uint32_t *jmotif, const uint32_t *jrank,
                                        const uint32_t *jstart, const uint32_t *jend,
                                        unsigned long long err_seq, uint32_t *detail4) {
    hipLaunchKernelGGL(k_find_genome_error, grid1d(n_j, 256), dim3(256), 0, s, n_j, first, jmotif, jrank,
                       jstart, jend, err_seq, detail4);
    return hipGetLastError();
}
