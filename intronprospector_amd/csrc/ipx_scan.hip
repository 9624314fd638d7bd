// ipx_scan.hip -- CIGAR scan: alignments -> QC-passed junction records.
//
// Device restatement of the reference's per-alignment work
//   process_alignment        junctions_extractor.cc:533-549
//   ReadJunctionExtractor    junctions_extractor.cc:114-319 (state machine)
//   get_junction_strand      junctions_extractor.cc:322-371
//   get_read_category        junctions_extractor.cc:49-95
//   junction_qc              junctions_extractor.cc:384-401
// Everything that depends on the junction table (filter, strand correction,
// creation order, counts) happens later in ipx_reduce.hip.
//
// Mapping: one lane per alignment; see k_scan below for the tile structure.
#include "ipx_common.h"

namespace {

enum : uint32_t { ST_LIMBO = 1, ST_LEFT = 2, ST_INTRON = 4, ST_RIGHT = 8 };

// BAM flag bits
enum : uint32_t {
    F_PAIRED = 0x1, F_PROPER = 0x2, F_UNMAP = 0x4, F_MUNMAP = 0x8, F_REVERSE = 0x10,
    F_MREVERSE = 0x20, F_READ1 = 0x40, F_READ2 = 0x80, F_SECONDARY = 0x100, F_QCFAIL = 0x200,
    F_DUP = 0x400, F_SUPPLEMENTARY = 0x800
};

// junctions_extractor.cc:333-359
__device__ __forceinline__ uint32_t strand_from_flag(uint32_t flag, int32_t strandness) {
    uint32_t reversed = (flag & F_REVERSE) != 0, mate_reversed = (flag & F_MREVERSE) != 0;
    uint32_t first = (flag & F_READ1) != 0, second = (flag & F_READ2) != 0;
    uint32_t first_strand = first ^ reversed, second_strand = second ^ mate_reversed;
    uint32_t flip = (strandness - 1) ? 0u : 1u;    // RF (1): xor 1, FR (2): xor 0
    first_strand ^= flip;
    second_strand ^= flip;
    uint32_t s = first_strand ? '+' : '-';
    return (first_strand == second_strand) ? s : '.';
}

// junctions_extractor.cc:71-95 over the nh class; 3 = would throw
__device__ __forceinline__ uint32_t read_category(uint32_t flag, uint32_t nh_class) {
    uint32_t by_tag = (nh_class == 0) ? 2u : (nh_class == 1 ? 0u : 1u);   // UNSURE / SINGLE / MULTI
    if (flag & (F_QCFAIL | F_UNMAP | F_DUP)) return 3u;
    if (flag & (F_SECONDARY | F_SUPPLEMENTARY)) return 1u;
    if (flag & F_PAIRED) return (flag & F_PROPER) ? by_tag : 2u;
    return by_tag;
}

struct Rec {
    int64_t anchor_start, intron_start, intron_end, anchor_end;
    uint32_t lmm, rmm, lind, rind;
};

struct ReadCtx {
    int64_t pos;
    uint32_t flag, nhc, rank, strand, cat;
    uint64_t read_index;
    bool read_rejected;        // junction_qc's per-read part, junctions_extractor.cc:388-391
};

// junction_qc, junctions_extractor.cc:392-397 (int64 arithmetic, uint32 sums wrap)
__device__ __forceinline__ bool qc_pass(const ScanParams &p, const ReadCtx &c, const Rec &r) {
    const int64_t ilen = r.intron_end - r.intron_start;
    const int64_t left = (r.intron_start - r.anchor_start) - (int64_t)(uint32_t)(r.lmm + r.lind);
    const int64_t right = (r.anchor_end - r.intron_end) - (int64_t)(uint32_t)(r.rmm + r.rind);
    return !c.read_rejected && ilen >= (int64_t)p.min_intron_length && ilen <= (int64_t)p.max_intron_length &&
           left >= (int64_t)p.min_anchor_length && right >= (int64_t)p.min_anchor_length;
}

struct KeyBits { uint32_t o0, o1, o2, a0, a1, a2; };

__device__ __forceinline__ void store_record(const ScanParams &p, const ReadCtx &c, const Rec &r,
                                             uint32_t ordinal, uint64_t idx, uint4 *__restrict__ items,
                                             uint4 *__restrict__ pays, uint8_t *__restrict__ digs, KeyBits &kb) {
    const uint32_t start = (uint32_t)r.intron_start;
    const uint32_t len = (uint32_t)(r.intron_end - r.intron_start);
    const uint32_t off16 = (uint32_t)(r.intron_start - c.pos) & 0xffffu;
    const uint32_t s1 = 16 + p.key.len_bits;        // <= 48
    const uint32_t s2 = s1 + 32;                    // <= 80
    uint64_t lo = (uint64_t)off16 | ((uint64_t)len << 16) | ((uint64_t)start << s1);
    uint64_t hi = (s1 > 32) ? ((uint64_t)start >> (64 - s1)) : 0;
    if (s2 < 64) {
        lo |= (uint64_t)c.rank << s2;
        if (s2 > 32) hi |= (uint64_t)c.rank >> (64 - s2);
    } else {
        hi |= (uint64_t)c.rank << (s2 - 64);
    }
    const unsigned long long order = (c.read_index << IPX_ORD_SHIFT_READ) |
                                     ((unsigned long long)ordinal << IPX_ORD_SHIFT_ORDINAL) |
                                     (c.strand << 2) | c.cat;
    items[idx] = make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)idx);
    pays[idx] = make_uint4((uint32_t)order, (uint32_t)(order >> 32), (uint32_t)r.anchor_start,
                           (uint32_t)r.anchor_end);
    digs[idx] = (uint8_t)lo;                       // digit 0 for the first sort pass
    kb.o0 |= (uint32_t)lo; kb.o1 |= (uint32_t)(lo >> 32); kb.o2 |= (uint32_t)hi;
    kb.a0 &= (uint32_t)lo; kb.a1 &= (uint32_t)(lo >> 32); kb.a2 &= (uint32_t)hi;
}

// The reference's ReadJunctionExtractor::parse_read over one CIGAR.  EMIT=false counts
// the records that pass junction_qc (and reports a CIGAR error); EMIT=true stores them
// at slot0, slot0+1, ...  Returns the number of QC-passed records.
struct Pool {                    // LDS staging of a tile's records (k_scan_pool)
    uint4 *item, *pay;
    uint32_t *n;
    uint32_t cap;
};

// MODE 0: count; MODE 1: store at slot0 + ordinal in the global record arrays; MODE 2: append to the LDS pool
template <int MODE>
__device__ __forceinline__ uint32_t run_read(const ScanParams &p, const ReadCtx &c,
                                             const uint32_t *__restrict__ cig, uint32_t nc,
                                             uint64_t slot0, uint64_t capacity,
                                             uint4 *__restrict__ items, uint4 *__restrict__ pays,
                                             uint8_t *__restrict__ digs, KeyBits &kb,
                                             unsigned long long &err, const Pool *pool = nullptr) {
    uint32_t state = ST_LIMBO;
    int64_t target_pos = c.pos, anchor_start = -1, intron_start = -1, intron_end = -1, anchor_end = -1;
    uint32_t lmm = 0, rmm = 0, lind = 0, rind = 0, cur_indel = 0, ordinal = 0;

    auto emit = [&]() {
        const Rec r{anchor_start, intron_start, intron_end, anchor_end, lmm, rmm, lind, rind};
        if (qc_pass(p, c, r)) {
            if (MODE == 1) {
                const uint64_t idx = slot0 + ordinal;
                if (idx < capacity && ordinal <= IPX_ORD_MAX_ORDINAL) store_record(p, c, r, ordinal, idx, items, pays, digs, kb);
            } else if (MODE == 2) {
                const uint32_t slot = atomicAdd(pool->n, 1u);
                if (slot < pool->cap && ordinal <= IPX_ORD_MAX_ORDINAL) {
                    uint8_t dummy_dig;
                    KeyBits dummy_kb;
                    store_record(p, c, r, ordinal, 0, pool->item + slot, pool->pay + slot, &dummy_dig, dummy_kb);
                }
            }
            ++ordinal;
        }
    };

    for (uint32_t k = 0; k < nc; ++k) {
        const uint32_t w = cig[k];
        const uint32_t op = w & 0xfu, len = w >> 4;
        // BAM_CIGAR_STR "MIDNSHP=XB": M0 I1 D2 N3 S4 H5 P6 =7 X8 B9
        if (op == 3) {                                  // N   :208-229
            if (state == ST_RIGHT) {
                emit();
                anchor_start = intron_end;
                lmm = rmm; rmm = 0; lind = rind; rind = 0;
                state = ST_LEFT;
            }
            if (state == ST_LEFT) {
                intron_start = target_pos;
                intron_end = target_pos + len;
                state = ST_INTRON;
            }
            cur_indel = 0;
            target_pos += len;
        } else if (op == 0 || op == 7 || op == 8) {     // M = X  :232-254
            if (state == ST_LIMBO) {
                anchor_start = target_pos;
                anchor_end = target_pos + len;
                lmm = rmm = 0;                           // indel counters keep their value
                state = ST_LEFT;
            } else if (state & (ST_LEFT | ST_RIGHT)) {
                anchor_end = target_pos + len;
            } else {                                    // ST_INTRON
                intron_end = target_pos;
                anchor_end = target_pos + len;
                state = ST_RIGHT;
            }
            if (op == 8) {
                if (state == ST_LEFT) lmm += len;
                else if (state == ST_RIGHT) rmm += len;
            }
            cur_indel = 0;
            target_pos += len;
        } else if (op == 1 || op == 2) {                // I, D  :257-275
            if (state == ST_LEFT) lind += len;
            else if (state == ST_RIGHT) rind += len;
            cur_indel += len;
            if (cur_indel > p.max_anchor_indel_size) {
                if (state == ST_RIGHT) emit();
                state = ST_LIMBO;
            }
            if (op == 2) target_pos += len;
        } else if (op == 4 || op == 5) {                // S, H  :306-311 -> enter_limbo
            if (state == ST_RIGHT) emit();
            state = ST_LIMBO;
        } else {                                        // :312-313: the exception ends the read
            const uint32_t ch = (op == 6) ? 'P' : (op == 9 ? 'B' : '?');
            const unsigned long long e = (c.read_index << IPX_ORD_SHIFT_READ) |
                                         ((unsigned long long)ordinal << IPX_ORD_SHIFT_ORDINAL) |
                                         (IPX_ERRK_CIGAR << 8) | ch;
            err = e < err ? e : err;
            return ordinal;
        }
    }
    if (state == ST_RIGHT) emit();                      // finish()  :278-282
    if (ordinal > IPX_ORD_MAX_ORDINAL) {
        const unsigned long long e = (c.read_index << IPX_ORD_SHIFT_READ) | (IPX_ERRK_TARGET << 8) | 1u;
        err = e < err ? e : err;
    }
    return ordinal;
}

// One block = one tile of 256*RPT consecutive alignments, thread t takes alignments
// tile + r*256 + t.  Pass 1 counts each alignment's records, a block scan turns the
// counts into slots (round-major, so that at a fixed r neighbouring lanes own
// neighbouring slots and the record stores coalesce), ONE atomicAdd per tile reserves
// the tile's range of the record arrays, pass 2 re-runs the state machines and stores.
// (One returning atomic per wave and step on a single counter serialises at ~17 ns each:
// that version spent 5.4 ms on 10 M alignments; the CIGAR words of a tile are re-read
// from L1/L2.)
template <uint32_t RPT, uint32_t POOL>
__global__ __launch_bounds__(256) void k_scan(ScanParams p, uint64_t n, uint64_t n_words, uint64_t base_read_index,
                                              const int32_t *__restrict__ a_tid,
                                              const int32_t *__restrict__ a_pos,
                                              const uint16_t *__restrict__ a_flag,
                                              const uint8_t *__restrict__ a_xs,
                                              const uint8_t *__restrict__ a_nh,
                                              const uint32_t *__restrict__ a_off,
                                              const uint32_t *__restrict__ a_nc,
                                              const uint32_t *__restrict__ cigar,
                                              const uint32_t *__restrict__ rank_of_tid,
                                              uint4 *__restrict__ items, uint4 *__restrict__ pays,
                                              uint8_t *__restrict__ digs,
                                              uint64_t capacity, Counters *counters) {
    __shared__ uint32_t wsum[RPT][4];
    __shared__ uint32_t kb_s[4][6];
    KeyBits kb{0u, 0u, 0u, ~0u, ~0u, ~0u};
    __shared__ uint32_t cnt_s[RPT][256], base_s[RPT][256];
    __shared__ unsigned long long tile_slot;
    // POOL > 0: a tile's records are first staged in LDS by ONE run of the state machines and then
    // copied out as a block (no second run); a tile that emits more than POOL records falls back to
    // the count / reserve / re-run scheme below
    __shared__ uint4 pool_item[POOL ? POOL : 1], pool_pay[POOL ? POOL : 1];
    __shared__ uint32_t pool_n;
    const Pool pool{pool_item, pool_pay, &pool_n, POOL};
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const uint64_t n_tiles = (n + 256ull * RPT - 1) / (256ull * RPT);

    // per-alignment attributes; re-read in pass 2 (L1/L2 hits) instead of held in registers
    auto load_ctx = [&](uint64_t i, ReadCtx &c, uint32_t &nc, uint32_t &off, unsigned long long &err) {
        c.pos = 0; c.flag = 0; c.nhc = 0; c.rank = 0; c.strand = '.'; c.read_index = base_read_index + i;
        nc = 0; off = 0;
        if (i < n) {
            const int32_t tid = a_tid[i];
            c.flag = a_flag[i];
            nc = a_nc[i];
            // junctions_extractor.cc:536
            if (nc <= 1 || tid < 0 || (c.flag & (F_QCFAIL | F_DUP))) nc = 0;
            if (nc) {
                if ((uint32_t)tid >= p.n_targets) {
                    const unsigned long long e = (c.read_index << IPX_ORD_SHIFT_READ) | (IPX_ERRK_TARGET << 8);
                    err = e < err ? e : err;
                    nc = 0;
                } else {
                    c.pos = a_pos[i];
                    off = a_off[i];
                    if ((uint64_t)off + nc > n_words) {          // words outside the batch: never dereferenced
                        const unsigned long long e = (c.read_index << IPX_ORD_SHIFT_READ) | (IPX_ERRK_TARGET << 8) | 3u;
                        err = e < err ? e : err;
                        nc = 0;
                    }
                    c.nhc = a_nh[i];
                    c.rank = rank_of_tid[tid];
                    const uint32_t xs = a_xs[i];
                    c.strand = (p.strandness != IPX_UNSTRANDED) ? strand_from_flag(c.flag, p.strandness)
                                                                : (xs ? xs : (uint32_t)'.');
                }
            }
        }
        c.cat = read_category(c.flag, c.nhc);
        c.read_rejected = (c.flag & F_SECONDARY) || (p.exclude_multi && c.nhc == 2);
    };

    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        unsigned long long my_err = IPX_ERRW_NONE;
        if (POOL) {
            if (t == 0) pool_n = 0;
            __syncthreads();
#pragma unroll 1
            for (uint32_t r = 0; r < RPT; ++r) {
                const uint64_t i = tile * (256ull * RPT) + (uint64_t)r * 256 + t;
                ReadCtx c;
                uint32_t nc, off;
                load_ctx(i, c, nc, off, my_err);
                if (nc) (void)run_read<2>(p, c, cigar + off, nc, 0, 0, nullptr, nullptr, nullptr, kb, my_err, &pool);
            }
            __syncthreads();
            const uint32_t total = pool_n;
            if (total <= POOL) {
                if (t == 0) tile_slot = total ? atomicAdd(&counters->n_records, (unsigned long long)total) : 0ull;
                __syncthreads();
                const uint64_t slot = tile_slot;
                for (uint32_t k = t; k < total; k += 256) {
                    const uint64_t idx = slot + k;
                    if (idx < capacity) {
                        uint4 it = pool_item[k];
                        it.w = (uint32_t)idx;
                        items[idx] = it;
                        pays[idx] = pool_pay[k];
                        digs[idx] = (uint8_t)it.x;
                        kb.o0 |= it.x; kb.o1 |= it.y; kb.o2 |= it.z;
                        kb.a0 &= it.x; kb.a1 &= it.y; kb.a2 &= it.z;
                    }
                }
                unsigned long long e = my_err;
                for (int d = 32; d >= 1; d >>= 1) {
                    const unsigned long long o = __shfl_xor(e, d);
                    e = o < e ? o : e;
                }
                if (lane == 0 && e != IPX_ERRW_NONE) atomicMin(&counters->err_word, e);
                __syncthreads();
                continue;
            }
        }
        // pass 1: count.  Loops over r stay rolled (one copy of the state machine in the
        // code, low register pressure); per-round counts live in LDS.
#pragma unroll 1
        for (uint32_t r = 0; r < RPT; ++r) {
            const uint64_t i = tile * (256ull * RPT) + (uint64_t)r * 256 + t;
            ReadCtx c;
            uint32_t nc, off;
            load_ctx(i, c, nc, off, my_err);
            const uint32_t cn = nc ? run_read<0>(p, c, cigar + off, nc, 0, 0, nullptr, nullptr, nullptr, kb, my_err) : 0;
            cnt_s[r][t] = cn;
            // inclusive scan inside the wave
            uint32_t v = cn;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t x = __shfl_up(v, o);
                if ((int)lane >= o) v += x;
            }
            base_s[r][t] = v - cn;                   // exclusive inside the wave, completed below
            if (lane == 63) wsum[r][wave] = v;
        }
        __syncthreads();
        uint32_t tile_total = 0;
#pragma unroll 1
        for (uint32_t r = 0; r < RPT; ++r) {
            uint32_t before = tile_total;
            for (uint32_t w = 0; w < 4; ++w) {
                const uint32_t sw = wsum[r][w];
                if (w < wave) before += sw;
                tile_total += sw;
            }
            base_s[r][t] += before;
        }
        if (t == 0) tile_slot = tile_total ? atomicAdd(&counters->n_records, (unsigned long long)tile_total) : 0ull;
        __syncthreads();
        const uint64_t slot = tile_slot;
        if (tile_total) {
#pragma unroll 1
            for (uint32_t r = 0; r < RPT; ++r) {
                if (cnt_s[r][t]) {
                    const uint64_t i = tile * (256ull * RPT) + (uint64_t)r * 256 + t;
                    unsigned long long dummy = IPX_ERRW_NONE;
                    ReadCtx c;
                    uint32_t nc, off;
                    load_ctx(i, c, nc, off, dummy);
                    (void)run_read<1>(p, c, cigar + off, nc, slot + base_s[r][t], capacity, items, pays, digs, kb, dummy);
                }
            }
        }
        // earliest error of the wave -> one atomic
        unsigned long long e = my_err;
        for (int d = 32; d >= 1; d >>= 1) {
            const unsigned long long o = __shfl_xor(e, d);
            e = o < e ? o : e;
        }
        if (lane == 0 && e != IPX_ERRW_NONE) atomicMin(&counters->err_word, e);
        __syncthreads();          // tile_slot / wsum are reused by the next tile
    }
    // OR / AND of every key this block stored -> six atomics per block
    uint32_t v[6] = {kb.o0, kb.o1, kb.o2, kb.a0, kb.a1, kb.a2};
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        for (int dlt = 32; dlt >= 1; dlt >>= 1) {
            const uint32_t o = __shfl_xor(v[k], dlt);
            v[k] = (k < 3) ? (v[k] | o) : (v[k] & o);
        }
        if (lane == 0) kb_s[wave][k] = v[k];
    }
    __syncthreads();
    if (t < 6) {
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

hipError_t ipx_launch_scan(hipStream_t s, const ScanParams &p, const ipx_batch &b,
                           const uint32_t *rank_of_tid, uint4 *items, uint4 *pays, uint8_t *digs,
                           uint64_t capacity, Counters *counters) {
    if (b.n_alignments == 0) return hipSuccess;
    // short CIGARs: 2 alignments per thread and tile (512 per atomic), records staged in LDS; long ones: 1
    const bool short_reads = b.n_cigar_words <= 16 * b.n_alignments;
    const uint64_t per_tile = short_reads ? 512 : 256;
    uint64_t blocks = ipx_div_up(b.n_alignments, per_tile);
    if (blocks > 256ull * 16) blocks = 256ull * 16;
    if (short_reads)
        hipLaunchKernelGGL((k_scan<2, 704>), dim3((uint32_t)blocks), dim3(256), 0, s, p, b.n_alignments,
                           b.n_cigar_words, b.base_read_index, b.tid, b.pos, b.flag, b.xs, b.nh, b.cigar_off, b.n_cigar,
                           b.cigar, rank_of_tid, items, pays, digs, capacity, counters);
    else
        hipLaunchKernelGGL((k_scan<1, 0>), dim3((uint32_t)blocks), dim3(256), 0, s, p, b.n_alignments,
                           b.n_cigar_words, b.base_read_index, b.tid, b.pos, b.flag, b.xs, b.nh, b.cigar_off, b.n_cigar,
                           b.cigar, rank_of_tid, items, pays, digs, capacity, counters);
    ipx_prof_mark(s, IPXK_SCAN);
    return hipGetLastError();
}
