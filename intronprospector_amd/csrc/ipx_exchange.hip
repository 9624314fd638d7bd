// ipx_exchange.hip -- multi-GPU record exchange (no reference counterpart).
//
// Alignments shard over the GPUs by contiguous read ranges, but a junction's records
// must meet on ONE GPU to be reduced with the reference's order-dependent semantics.
// Records therefore move once, over xGMI, by key range: every rank splits its pending
// records at the same splitters (partition key = chromosome rank << 32 | intron start,
// so all records of a junction land in one part), the parts are exchanged with one
// all_to_all, and each rank sorts + reduces the key range it owns.  Every record carries
// its global order64, so nothing depends on the arrival order.
#include "ipx_common.h"

namespace {

constexpr uint32_t MAX_PARTS = 64;
// records at the head of a slot of the slotted exchange: one of flags and totals, then the sender's counts for ALL parts
// (eight 32-bit counts per record), so that every receiver knows the whole counts matrix
__host__ __device__ constexpr uint32_t slot_header_records(uint32_t n_parts) { return 1u + (n_parts + 7u) / 8u; }

__device__ __forceinline__ unsigned long long part_key(const uint4 &it, const KeyLayout &kl) {
    const KeyParts k = unpack_key(it, kl);
    return ((unsigned long long)k.rank << 32) | k.start;
}

// number of splitters <= key  (splitters ascending)
__device__ __forceinline__ uint32_t part_of(unsigned long long key, const unsigned long long *spl, uint32_t n_parts) {
    uint32_t p = 0;
    for (uint32_t i = 0; i + 1 < n_parts; ++i) p += (spl[i] <= key) ? 1u : 0u;
    return p;
}

__global__ void k_sample(const uint4 *__restrict__ items, uint64_t n, KeyLayout kl, uint32_t n_samples,
                         unsigned long long *__restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_samples) return;
    const uint64_t idx = n ? (uint64_t)(((unsigned __int128)i * n) / n_samples) : 0;
    out[i] = n ? part_key(items[idx < n ? idx : n - 1], kl) : ~0ull;
}

// Partitioning without global atomics: block b of a fixed grid owns the tiles b, b + G, b + 2G, ... (2048
// records each) in both kernels.  k_part_count leaves every block's per-part counts, k_part_plan turns
// them into every block's write position per part (parts laid out one after the other) and the part
// totals, k_part_scatter replays the same tiles and places the records.
__global__ __launch_bounds__(256) void k_part_count(const uint4 *__restrict__ items, uint64_t n, KeyLayout kl,
                                                    uint32_t n_parts, const unsigned long long *__restrict__ spl_g,
                                                    unsigned long long *__restrict__ blk) {
    __shared__ unsigned long long spl[MAX_PARTS];
    __shared__ uint32_t c[MAX_PARTS];
    if (threadIdx.x < MAX_PARTS) {
        c[threadIdx.x] = 0;
        spl[threadIdx.x] = (threadIdx.x + 1 < n_parts) ? spl_g[threadIdx.x] : ~0ull;
    }
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t n_tiles = (n + 2047) / 2048;
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
#pragma unroll
        for (uint32_t r = 0; r < 8; ++r) {
            const uint64_t i = tile * 2048 + r * 256 + threadIdx.x;
            const bool ok = i < n;
            const uint32_t part = ok ? part_of(part_key(items[i], kl), spl, n_parts) : 0xffffffffu;
            // records arrive in scan order, i.e. nearly in genome order: a wave rarely holds more than two parts
            unsigned long long todo = ipx_ballot(ok);
            while (todo) {
                const uint32_t pv = __builtin_amdgcn_readlane(part, __ffsll(todo) - 1);
                const unsigned long long m = ipx_ballot(part == pv);
                if (lane == (uint32_t)(__ffsll(m) - 1)) atomicAdd(&c[pv], (uint32_t)__popcll(m));
                todo &= ~m;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < MAX_PARTS) blk[(uint64_t)blockIdx.x * MAX_PARTS + threadIdx.x] = c[threadIdx.x];
}

// one block of 1024 threads, thread b for partition block b: blk[b][p] (counts) -> blk[b][p] (where block b
// writes part p: parts one after the other, inside a part the blocks in order), totals[p]
__global__ __launch_bounds__(1024) void k_part_plan(unsigned long long *__restrict__ blk, uint32_t n_blocks,
                                                    uint32_t n_parts, unsigned long long *__restrict__ totals) {
    __shared__ unsigned long long wave_sum[16];
    __shared__ unsigned long long part_total[MAX_PARTS];
    const uint32_t b = threadIdx.x, lane = b & 63u, wave = b >> 6;
    // (barriers, round-4 audit: both sit in the body of a loop over p < n_parts, a kernel argument: every thread makes
    // every turn; `b < n_blocks` guards loads and stores only.  The same in k_part_plan_slots below.)
    for (uint32_t p = 0; p < n_parts; ++p) {
        const unsigned long long v = b < n_blocks ? blk[(uint64_t)b * MAX_PARTS + p] : 0ull;
        unsigned long long incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned long long x = __shfl_up(incl, o);
            if ((int)lane >= o) incl += x;
        }
        if (lane == 63) wave_sum[wave] = incl;
        __syncthreads();
        unsigned long long before = 0, all = 0;
        for (uint32_t w = 0; w < 16; ++w) {
            if (w < wave) before += wave_sum[w];
            all += wave_sum[w];
        }
        if (b < n_blocks) blk[(uint64_t)b * MAX_PARTS + p] = before + incl - v;     // exclusive over the blocks
        if (b == 0) part_total[p] = all;
        __syncthreads();
    }
    if (b < MAX_PARTS) totals[b] = b < n_parts ? part_total[b] : 0ull;
    for (uint32_t p = 1; p < n_parts; ++p) {
        unsigned long long off = 0;
        for (uint32_t q = 0; q < p; ++q) off += part_total[q];
        if (b < n_blocks) blk[(uint64_t)b * MAX_PARTS + p] += off;
    }
}

__global__ __launch_bounds__(256) void k_part_scatter(const uint4 *__restrict__ items, const uint4 *__restrict__ pays,
                                                      const uint32_t *__restrict__ cnts,
                                                      uint64_t n, KeyLayout kl, uint32_t n_parts,
                                                      const unsigned long long *__restrict__ spl_g,
                                                      const unsigned long long *__restrict__ blk,
                                                      uint4 *__restrict__ out, uint64_t slot_records,
                                                      const unsigned long long *__restrict__ totals,
                                                      uint4 *__restrict__ spill, uint64_t spill_records) {
    __shared__ unsigned long long spl[MAX_PARTS];
    __shared__ unsigned long long cursor[MAX_PARTS];        // this block's next position per part
    __shared__ unsigned long long spill_base[MAX_PARTS];    // slots with a spill: where part p's tail begins in it
    if (threadIdx.x < MAX_PARTS) {
        spl[threadIdx.x] = (threadIdx.x + 1 < n_parts) ? spl_g[threadIdx.x] : ~0ull;
        cursor[threadIdx.x] = blk[(uint64_t)blockIdx.x * MAX_PARTS + threadIdx.x];
        unsigned long long before = 0;
        if (spill && slot_records) {
            // (totals: written by k_part_plan_slots, the launch before this one)
            const unsigned long long room = slot_records - slot_header_records(n_parts);
            for (uint32_t q = 0; q < threadIdx.x && q < n_parts; ++q) before += totals[q] > room ? totals[q] - room : 0ull;
        }
        spill_base[threadIdx.x] = before;
    }
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t n_tiles = (n + 2047) / 2048;
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
#pragma unroll
        for (uint32_t r = 0; r < 8; ++r) {
            const uint64_t i = tile * 2048 + r * 256 + threadIdx.x;
            const bool ok = i < n;
            uint4 it = make_uint4(0, 0, 0, 0);
            if (ok) it = items[i];
            const uint32_t part = ok ? part_of(part_key(it, kl), spl, n_parts) : 0xffffffffu;
            unsigned long long pos = 0;
            unsigned long long todo = ipx_ballot(ok);
            while (todo) {
                const uint32_t pv = __builtin_amdgcn_readlane(part, __ffsll(todo) - 1);
                const unsigned long long m = ipx_ballot(part == pv);
                const int leader = __ffsll(m) - 1;
                unsigned long long at = 0;
                if ((int)lane == leader) at = atomicAdd(&cursor[pv], (unsigned long long)__popcll(m));
                at = __shfl(at, leader);
                if (part == pv)
                    pos = at + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                todo &= ~m;
            }
            // (slots: a part's records beyond its slot go to the spill, part after part -- the header says that there are
            // some and a second all-to-all brings them over; without a spill they are not sent and the caller repeats)
            if (ok && (slot_records == 0 || pos < ((unsigned long long)part + 1) * slot_records)) {      // (a slot's header lies below its first position)
                // on the way the item's fourth word (the record index, meaningless elsewhere) carries the count word
                const uint32_t ri = IPX_ITEM_INDEX(it.w);
                const uint4 py = pays[ri];
                it.w = (it.w & IPX_ITEM_FOLDED) ? cnts[ri] : IPX_CNT_ONE(py.x & 3u);
                out[2 * pos] = it;
                out[2 * pos + 1] = py;
            } else if (ok && spill) {
                const unsigned long long at = spill_base[part] + (pos - ((unsigned long long)part + 1) * slot_records);
                if (at < spill_records) {         // (the caller gives room for every record; a smaller spill loses the rest -- its header says overflow all the same)
                    const uint32_t ri = IPX_ITEM_INDEX(it.w);
                    const uint4 py = pays[ri];
                    it.w = (it.w & IPX_ITEM_FOLDED) ? cnts[ri] : IPX_CNT_ONE(py.x & 3u);
                    spill[2 * at] = it;
                    spill[2 * at + 1] = py;
                }
            }
        }
    }
}

// The slotted form of the plan (ipx_records_partition_slots): part p goes to SLOT p, slot_records records wide, whose
// first record is a header -- [count of the part (64 bit), the sender's largest part (64 bit), flags (bit 0: some
// part of this sender does not fit its slot), error status, error order (64 bit)] -- so that an all-to-all with equal
// splits carries the counts, the overflow flag and the ranks' errors along and no host has to look at any of them
// before the exchange.  blk[b][p] becomes block b's first position in slot p (in records, from the start of dst).
__global__ __launch_bounds__(1024) void k_part_plan_slots(unsigned long long *__restrict__ blk, uint32_t n_blocks,
                                                          uint32_t n_parts, uint64_t slot_records,
                                                          unsigned long long error_order, uint32_t error_status,
                                                          unsigned long long *__restrict__ totals, uint4 *__restrict__ out) {
    __shared__ unsigned long long wave_sum[16];
    __shared__ unsigned long long part_total[MAX_PARTS];
    const uint32_t b = threadIdx.x, lane = b & 63u, wave = b >> 6;
    for (uint32_t p = 0; p < n_parts; ++p) {
        const unsigned long long v = b < n_blocks ? blk[(uint64_t)b * MAX_PARTS + p] : 0ull;
        unsigned long long incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned long long x = __shfl_up(incl, o);
            if ((int)lane >= o) incl += x;
        }
        if (lane == 63) wave_sum[wave] = incl;
        __syncthreads();
        unsigned long long before = 0, all = 0;
        for (uint32_t w = 0; w < 16; ++w) {
            if (w < wave) before += wave_sum[w];
            all += wave_sum[w];
        }
        if (b < n_blocks) blk[(uint64_t)b * MAX_PARTS + p] = (unsigned long long)p * slot_records + slot_header_records(n_parts) + before + incl - v;
        if (b == 0) part_total[p] = all;
        __syncthreads();
    }
    if (b < MAX_PARTS) totals[b] = b < n_parts ? part_total[b] : 0ull;
    if (b < n_parts) {
        unsigned long long mx = 0;
        for (uint32_t q = 0; q < n_parts; ++q) mx = part_total[q] > mx ? part_total[q] : mx;
        const unsigned long long c = part_total[b];
        out[2 * (uint64_t)b * slot_records] = make_uint4((uint32_t)c, (uint32_t)(c >> 32), (uint32_t)mx, (uint32_t)(mx >> 32));
        out[2 * (uint64_t)b * slot_records + 1] = make_uint4(mx + slot_header_records(n_parts) > slot_records ? 1u : 0u, error_status,
                                                             (uint32_t)error_order, (uint32_t)(error_order >> 32));
        // the counts row (the same in every slot this rank sends): four counts per uint4
        uint32_t *row = reinterpret_cast<uint32_t *>(out + 2 * (uint64_t)b * slot_records + 2);
        for (uint32_t q = 0; q < ((n_parts + 7u) & ~7u); ++q)
            row[q] = q < n_parts ? (uint32_t)(part_total[q] > 0xffffffffull ? 0xffffffffull : part_total[q]) : 0u;
    }
}

// received slots -> the handle's record arrays: slot s holds min(count_s, slot_records - 1) records behind its header;
// they are laid out slot after slot.  info[0..1] total records, [2..3] the largest part any sender had, [4] overflow
// (some sender's flag), [5] the sender with the earliest error, [6..7] that error's order, [8] its status, [9..10] the
// most any rank receives, [11..12] what all ranks receive together.
__global__ __launch_bounds__(256) void k_adopt_slots(const uint4 *__restrict__ rec, uint32_t n_slots, uint64_t slot_records,
                                                     uint64_t first_index, uint4 *__restrict__ items, uint4 *__restrict__ pays,
                                                     uint32_t *__restrict__ cnts, uint8_t *__restrict__ digs, DigitSpec first,
                                                     Counters *counters, uint32_t *__restrict__ info) {
    __shared__ uint32_t kb_s[4][6];
    __shared__ unsigned long long recv_tot[MAX_PARTS];
    const uint32_t slot = blockIdx.y;
    const uint64_t H = slot_header_records(n_slots), room = slot_records - H;
    uint64_t base = 0, count = 0;
    for (uint32_t q = 0; q <= slot; ++q) {
        const uint4 h = rec[2 * (uint64_t)q * slot_records];
        uint64_t c = (uint64_t)h.x | ((uint64_t)h.y << 32);
        if (c > room) c = room;
        if (q < slot) base += c; else count = c;
    }
    if (blockIdx.x == 0 && slot == 0) {              // (the barrier inside: blockIdx only -- the whole block or none of it)
        // what every rank receives in all (the counts rows of all senders): the same numbers on every rank
        if (threadIdx.x < MAX_PARTS) {
            unsigned long long tot = 0;
            if (threadIdx.x < n_slots)
                for (uint32_t q = 0; q < n_slots; ++q)
                    tot += reinterpret_cast<const uint32_t *>(rec + 2 * (uint64_t)q * slot_records + 2)[threadIdx.x];
            recv_tot[threadIdx.x] = tot;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint64_t total = 0, mx = 0;
            unsigned long long eo = ~0ull, most = 0, all = 0;
            uint32_t ovf = 0, es = 0, st = 0;
            for (uint32_t q = 0; q < n_slots; ++q) {
                const uint4 h = rec[2 * (uint64_t)q * slot_records], g = rec[2 * (uint64_t)q * slot_records + 1];
                const uint64_t c = (uint64_t)h.x | ((uint64_t)h.y << 32), m = (uint64_t)h.z | ((uint64_t)h.w << 32);
                total += c > room ? room : c;
                mx = m > mx ? m : mx;
                ovf |= g.x & 1u;
                const unsigned long long o = (unsigned long long)g.z | ((unsigned long long)g.w << 32);
                if (o < eo) { eo = o; es = q; st = g.y; }
                most = recv_tot[q] > most ? recv_tot[q] : most;
                all += recv_tot[q];
            }
            counters->n_records = first_index + total;
            info[0] = (uint32_t)total; info[1] = (uint32_t)(total >> 32); info[2] = (uint32_t)mx; info[3] = (uint32_t)(mx >> 32);
            info[4] = ovf; info[5] = es; info[6] = (uint32_t)eo; info[7] = (uint32_t)(eo >> 32); info[8] = st;
            info[9] = (uint32_t)most; info[10] = (uint32_t)(most >> 32); info[11] = (uint32_t)all; info[12] = (uint32_t)(all >> 32);
        }
    }
    uint32_t v[6] = {0u, 0u, 0u, ~0u, ~0u, ~0u};
    unsigned long long extra = 0;
    const uint4 *src = rec + 2 * ((uint64_t)slot * slot_records + H);
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (uint64_t)gridDim.x * 256) {
        uint4 it = src[2 * i];
        const uint64_t idx = first_index + base + i;
        cnts[idx] = it.w;
        const uint32_t reads = (it.w & IPX_CNT_MASK) + ((it.w >> IPX_CNT_BITS) & IPX_CNT_MASK) + (it.w >> (2 * IPX_CNT_BITS));
        extra += reads ? reads - 1u : 0u;
        const uint4 py = src[2 * i + 1];
        it.w = (uint32_t)idx | (it.w != IPX_CNT_ONE(py.x & 3u) ? IPX_ITEM_FOLDED : 0u);
        items[idx] = it;
        pays[idx] = py;
        digs[idx] = (uint8_t)ipx_digit_of(it, first);
        v[0] |= it.x; v[1] |= it.y; v[2] |= it.z;
        v[3] &= it.x; v[4] &= it.y; v[5] &= it.z;
    }
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (int d = 32; d >= 1; d >>= 1) extra += __shfl_xor(extra, d);
    if (lane == 0 && extra)
        atomicAdd(reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(counters) + IPX_CNT_MERGED), extra);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        for (int d = 32; d >= 1; d >>= 1) {
            const uint32_t o = __shfl_xor(v[k], d);
            v[k] = (k < 3) ? (v[k] | o) : (v[k] & o);
        }
        if (lane == 0) kb_s[wave][k] = v[k];
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const uint32_t t = threadIdx.x;
        const uint32_t x0 = kb_s[0][t], x1 = kb_s[1][t], x2 = kb_s[2][t], x3 = kb_s[3][t];
        if (t < 3) {
            const uint32_t o = x0 | x1 | x2 | x3;
            if (o) atomicOr(&counters->key_or[t], o);
        } else {
            const uint32_t a = x0 & x1 & x2 & x3;
            if (a != ~0u) atomicAnd(&counters->key_and[t - 3], a);
        }
    }
}

// received records -> the handle's record arrays (record index = position), digit-0 bytes,
// OR/AND of the keys
__global__ __launch_bounds__(256) void k_adopt(const uint4 *__restrict__ rec, uint64_t n, uint64_t first_index,
                                               uint4 *__restrict__ items, uint4 *__restrict__ pays,
                                               uint32_t *__restrict__ cnts,
                                               uint8_t *__restrict__ digs, DigitSpec first, Counters *counters) {
    __shared__ uint32_t kb_s[4][6];
    uint32_t v[6] = {0u, 0u, 0u, ~0u, ~0u, ~0u};
    unsigned long long extra = 0;             // reads the records stand for beyond one each (combined records)
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        uint4 it = rec[2 * i];
        const uint64_t idx = first_index + i;
        cnts[idx] = it.w;
        const uint32_t reads = (it.w & IPX_CNT_MASK) + ((it.w >> IPX_CNT_BITS) & IPX_CNT_MASK) + (it.w >> (2 * IPX_CNT_BITS));
        extra += reads ? reads - 1u : 0u;
        const uint4 py = rec[2 * i + 1];
        it.w = (uint32_t)idx | (it.w != IPX_CNT_ONE(py.x & 3u) ? IPX_ITEM_FOLDED : 0u);
        items[idx] = it;
        pays[idx] = py;
        digs[idx] = (uint8_t)ipx_digit_of(it, first);
        v[0] |= it.x; v[1] |= it.y; v[2] |= it.z;
        v[3] &= it.x; v[4] &= it.y; v[5] &= it.z;
    }
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (int d = 32; d >= 1; d >>= 1) extra += __shfl_xor(extra, d);
    if (lane == 0 && extra)
        atomicAdd(reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(counters) + IPX_CNT_MERGED), extra);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        for (int d = 32; d >= 1; d >>= 1) {
            const uint32_t o = __shfl_xor(v[k], d);
            v[k] = (k < 3) ? (v[k] | o) : (v[k] & o);
        }
        if (lane == 0) kb_s[wave][k] = v[k];
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const uint32_t t = threadIdx.x;
        const uint32_t x0 = kb_s[0][t], x1 = kb_s[1][t], x2 = kb_s[2][t], x3 = kb_s[3][t];
        if (t < 3) {
            const uint32_t o = x0 | x1 | x2 | x3;
            if (o) atomicOr(&counters->key_or[t], o);
        } else {
            const uint32_t a = x0 & x1 & x2 & x3;
            if (a != ~0u) atomicAnd(&counters->key_and[t - 3], a);
        }
    }
}

}  // namespace

hipError_t ipx_launch_sample(hipStream_t s, const uint4 *items, uint64_t n, const KeyLayout &kl,
                             uint32_t n_samples, unsigned long long *out) {
    if (!n_samples) return hipSuccess;
    hipLaunchKernelGGL(k_sample, dim3((n_samples + 255) / 256), dim3(256), 0, s, items, n, kl, n_samples, out);
    return hipGetLastError();
}

// blocks of the partition kernels: a fixed grid, the same for count, plan and scatter
uint32_t ipx_partition_blocks(uint64_t n) {
    const uint64_t tiles = ipx_div_up(n ? n : 1, 2048);
    return (uint32_t)(tiles < 1024 ? tiles : 1024);
}

// blk: MAX_PARTS (64) words per block of ipx_partition_blocks(n); totals: 64 words
hipError_t ipx_launch_partition(hipStream_t s, const uint4 *items, const uint4 *pays, const uint32_t *cnts, uint64_t n,
                                const KeyLayout &kl, uint32_t n_parts, const unsigned long long *splitters,
                                unsigned long long *blk, unsigned long long *totals, uint4 *out_records) {
    if (n_parts > MAX_PARTS) return hipErrorInvalidValue;
    const uint32_t blocks = ipx_partition_blocks(n);
    hipLaunchKernelGGL(k_part_count, dim3(blocks), dim3(256), 0, s, items, n, kl, n_parts, splitters, blk);
    hipLaunchKernelGGL(k_part_plan, dim3(1), dim3(1024), 0, s, blk, blocks, n_parts, totals);
    if (n)
        hipLaunchKernelGGL(k_part_scatter, dim3(blocks), dim3(256), 0, s, items, pays, cnts, n, kl, n_parts, splitters, blk,
                           out_records, (uint64_t)0, (const unsigned long long *)nullptr, (uint4 *)nullptr, (uint64_t)0);
    ipx_prof_mark(s, IPXK_PARTITION);
    return hipGetLastError();
}

hipError_t ipx_launch_partition_slots(hipStream_t s, const uint4 *items, const uint4 *pays, const uint32_t *cnts, uint64_t n,
                                      const KeyLayout &kl, uint32_t n_parts, const unsigned long long *splitters,
                                      unsigned long long *blk, unsigned long long *totals, uint64_t slot_records,
                                      unsigned long long error_order, uint32_t error_status, uint4 *out_slots,
                                      uint4 *spill, uint64_t spill_records) {
    if (n_parts > MAX_PARTS || slot_records <= slot_header_records(n_parts)) return hipErrorInvalidValue;
    const uint32_t blocks = ipx_partition_blocks(n);
    hipLaunchKernelGGL(k_part_count, dim3(blocks), dim3(256), 0, s, items, n, kl, n_parts, splitters, blk);
    hipLaunchKernelGGL(k_part_plan_slots, dim3(1), dim3(1024), 0, s, blk, blocks, n_parts, slot_records, error_order, error_status,
                       totals, out_slots);
    if (n)
        hipLaunchKernelGGL(k_part_scatter, dim3(blocks), dim3(256), 0, s, items, pays, cnts, n, kl, n_parts, splitters, blk,
                           out_slots, slot_records, (const unsigned long long *)totals, spill, spill_records);
    ipx_prof_mark(s, IPXK_PARTITION);
    return hipGetLastError();
}

hipError_t ipx_launch_adopt_slots(hipStream_t s, const uint4 *slots, uint32_t n_slots, uint64_t slot_records, uint64_t first_index,
                                  uint4 *items, uint4 *pays, uint32_t *cnts, uint8_t *digs, DigitSpec first, Counters *counters,
                                  uint32_t *info) {
    uint64_t bx = ipx_div_up(slot_records, 2048);
    if (bx > 512) bx = 512;
    hipLaunchKernelGGL(k_adopt_slots, dim3((uint32_t)bx, n_slots), dim3(256), 0, s, slots, n_slots, slot_records, first_index, items,
                       pays, cnts, digs, first, counters, info);
    ipx_prof_mark(s, IPXK_ADOPT);
    return hipGetLastError();
}

hipError_t ipx_launch_adopt(hipStream_t s, const uint4 *records, uint64_t n, uint64_t first_index, uint4 *items,
                            uint4 *pays, uint32_t *cnts, uint8_t *digs, DigitSpec first, Counters *counters) {
    if (!n) return hipSuccess;
    uint64_t blocks = ipx_div_up(n, 2048);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_adopt, dim3((uint32_t)blocks), dim3(256), 0, s, records, n, first_index, items, pays, cnts, digs,
                       first, counters);
    ipx_prof_mark(s, IPXK_ADOPT);
    return hipGetLastError();
}
