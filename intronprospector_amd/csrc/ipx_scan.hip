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
#include <algorithm>

#include <cstdlib>

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

// ---- where records go (ScanParams::base / n_stripes, ipx_common.h) -------------------------------
__device__ __forceinline__ uint32_t my_stripe(const ScanParams &p, uint32_t sub = 0) {
    // HW_REG_XCC_ID (id 20), bits 3:0: the XCD this wave runs on.  Any value is correct; the XCD's own keeps a stripe's
    // counter line in one place.  Four stripes per XCD (`sub`: the wave's number in its block): thirty-two counters, so
    // that a launch whose waves reserve often -- input in random order, where nothing folds -- does not queue on eight.
    return p.n_stripes ? ipx_stripe_of((uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | 20), sub) : 0u;
}
__device__ __forceinline__ unsigned long long *append_counter(Counters *c, const ScanParams &p, uint32_t stripe) {
    return p.n_stripes ? reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(c) + IPX_CNT_STRIPE0 + 128 * stripe)
                       : &c->n_records;
}
// record index of the i-th record of a stripe (i as returned by the stripe's counter)
__device__ __forceinline__ uint64_t append_index(const ScanParams &p, uint32_t stripe, uint64_t i) {
    if (!p.n_stripes) return i;
    const uint64_t chunk = (i >> IPX_STRIPE_CHUNK_BITS) * IPX_STRIPES + stripe;
    return p.base + ((chunk << IPX_STRIPE_CHUNK_BITS) | (i & ((1u << IPX_STRIPE_CHUNK_BITS) - 1)));
}

// key (item.x/y/z; item.w is left 0) and payload words of one record
__device__ __forceinline__ void pack_record(const ScanParams &p, const ReadCtx &c, const Rec &r,
                                            uint32_t ordinal, uint4 &item, uint4 &pay) {
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
    item = make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, 0u);
    pay = make_uint4((uint32_t)order, (uint32_t)(order >> 32), (uint32_t)r.anchor_start, (uint32_t)r.anchor_end);
}

// the same from 32-bit offsets relative to the alignment's position pos32 (k_scan_long)
__device__ __forceinline__ void pack_record_rel(const ScanParams &p, const ReadCtx &c, const Rec &r, uint32_t pos32,
                                                uint32_t ordinal, uint4 &item, uint4 &pay) {
    const uint32_t start = pos32 + (uint32_t)r.intron_start;
    const uint32_t len = (uint32_t)r.intron_end - (uint32_t)r.intron_start;
    const uint32_t off16 = (uint32_t)r.intron_start & 0xffffu;
    const uint32_t s1 = 16 + p.key.len_bits;
    const uint32_t s2 = s1 + 32;
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
    item = make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, 0u);
    pay = make_uint4((uint32_t)order, (uint32_t)(order >> 32), pos32 + (uint32_t)r.anchor_start,
                     pos32 + (uint32_t)r.anchor_end);
}

__device__ __forceinline__ void store_record(const ScanParams &p, const ReadCtx &c, const Rec &r,
                                             uint32_t ordinal, uint64_t idx, uint4 *__restrict__ items,
                                             uint4 *__restrict__ pays, uint32_t *__restrict__ cnts,
                                             uint8_t *__restrict__ digs, KeyBits &kb) {
    uint4 it, pay;
    pack_record(p, c, r, ordinal, it, pay);
    it.w = (uint32_t)idx;
    items[idx] = it;
    pays[idx] = pay;
    cnts[idx] = IPX_CNT_ONE(c.cat);
    digs[idx] = (uint8_t)ipx_digit_of(it, p.first);  // digit of the first sort pass
    kb.o0 |= it.x; kb.o1 |= it.y; kb.o2 |= it.z;
    kb.a0 &= it.x; kb.a1 &= it.y; kb.a2 &= it.z;
}

// The reference's ReadJunctionExtractor::parse_read over one CIGAR.  EMIT=false counts
// the records that pass junction_qc (and reports a CIGAR error); EMIT=true stores them
// at slot0, slot0+1, ...  Returns the number of QC-passed records.
// MODE 0: count; MODE 1: store at slot0 + ordinal of the stripe;
// MODE 3: records with ordinal >= slot0 each reserve their own slot of the stripe (k_scan_long's rare re-run)
template <int MODE>
__device__ __forceinline__ uint32_t run_read(const ScanParams &p, const ReadCtx &c,
                                             const uint32_t *__restrict__ cig, uint32_t nc,
                                             uint64_t slot0, uint64_t capacity,
                                             uint4 *__restrict__ items, uint4 *__restrict__ pays,
                                             uint32_t *__restrict__ cnts,
                                             uint8_t *__restrict__ digs, KeyBits &kb,
                                             unsigned long long &err,
                                             unsigned long long *n_records = nullptr, uint32_t stripe = 0) {
    uint32_t state = ST_LIMBO;
    int64_t target_pos = c.pos, anchor_start = -1, intron_start = -1, intron_end = -1, anchor_end = -1;
    uint32_t lmm = 0, rmm = 0, lind = 0, rind = 0, cur_indel = 0, ordinal = 0;

    auto emit = [&]() {
        const Rec r{anchor_start, intron_start, intron_end, anchor_end, lmm, rmm, lind, rind};
        if (qc_pass(p, c, r)) {
            if (MODE == 1) {
                const uint64_t idx = append_index(p, stripe, slot0 + ordinal);
                if (idx < capacity && ordinal <= IPX_ORD_MAX_ORDINAL) store_record(p, c, r, ordinal, idx, items, pays, cnts, digs, kb);
            } else if (MODE == 3) {
                if (ordinal >= slot0 && ordinal <= IPX_ORD_MAX_ORDINAL) {
                    const uint64_t idx = append_index(p, stripe, atomicAdd(n_records, 1ull));
                    if (idx < capacity) store_record(p, c, r, ordinal, idx, items, pays, cnts, digs, kb);
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

// ---- long CIGARs ---------------------------------------------------------------------------------
// k_scan_long is bound by the number of instructions per CIGAR op (a wave64 vector instruction takes
// four cycles on a SIMD) and by the latency of each lane's private stream of CIGAR words, so:
//   * every wave owns a contiguous range of alignments and works through it on its own (no block
//     barriers); a lane that finishes its alignment takes the next one at the top of the loop, so lanes
//     stay busy although CIGAR lengths differ.  The attributes of the next 64 alignments are staged in
//     LDS by one round of coalesced loads, so handing a lane its next alignment costs no memory round
//     trip;
//   * CIGAR words arrive as 16-byte loads (4 ops) into a ring of LONG_RING register quads; the loop body
//     is unrolled over the ring, every slot is refilled right after its four ops -- LONG_RING-1 chunks
//     ahead of use, unconditionally (lanes that need no more words re-read their first chunk), and the
//     ring is written by nothing else: a lane that was handed an alignment sits out one turn of the
//     ring ("warm") while the refills bring its first chunks.  So no load result passes through a select
//     or a copy, which would make the compiler wait for the load where it is issued.  (A 4-byte load per
//     op re-fetches the lane's 128-byte line from L2 for every op: the lines of all resident lanes do
//     not fit the 32 KB L1.)  A chunk that would reach past the end of the CIGAR array is read from a
//     16-word copy of the array's tail instead;
//   * positions are 32-bit offsets from the alignment's start.  The first op that would carry out of
//     32 bits parks the alignment: its earlier records stay (they are exact), the rest comes from the
//     64-bit run_read<3> (never seen on real data; tests/test_gpu_parity.py builds such CIGARs);
//   * the op is applied with selects, not branches; an op that ends a junction only snapshots six
//     registers.  junction_qc, key packing and the store run for many lanes at once when enough
//     snapshots wait (or one is about to be overwritten): a ballot hands out slots of the wave's LDS
//     pool, and a full pool goes to the record arrays with ONE global atomicAdd.
// Records of one alignment are told apart by their ordinal, so their place in the arrays is free.
constexpr uint32_t LONG_POOL = 256;          // records per wave pool (8 KB)
constexpr uint32_t LONG_DRAIN = 20;          // waiting snapshots that trigger a drain
constexpr uint32_t LONG_RING = 3;            // 16-byte chunks in flight per lane

struct LongStage {                            // next 64 alignments of a wave, one column per lane
    uint32_t pos[64], nc[64], meta[64], rank[64], off[64];
};

struct __attribute__((packed, aligned(4))) Words4 { uint32_t x, y, z, w; };

// copy of the last (up to) 8 CIGAR words followed by zero words (see ipx_launch_scan): a 16-byte load
// that would reach past the end of the array reads here instead
// (also starts the launch's list of deferred alignments at zero, see k_scan_rest)
__global__ void k_cigar_tail(const uint32_t *__restrict__ cigar, uint64_t n_words, uint32_t *__restrict__ tail,
                             uint32_t *__restrict__ defer_n, unsigned long long *__restrict__ merged_n) {
    const uint32_t t = threadIdx.x;
    const uint64_t tb = n_words >= 8 ? n_words - 8 : 0;
    if (t < 16) tail[t] = (tb + t < n_words) ? cigar[tb + t] : 0u;
    if (t == 16) *defer_n = 0;
    if (t == 17) *merged_n = 0;
}

// four words starting at word w of the CIGAR array (w < n_words); only the address is selected
__device__ __forceinline__ uint4 load_cigar4(const uint32_t *__restrict__ cigar, const uint32_t *__restrict__ tail,
                                             uint64_t n_words, uint64_t w) {
    const uint64_t tb = n_words >= 8 ? n_words - 8 : 0;
    const uint32_t *ptr = (w + 4 <= n_words) ? cigar + w : tail + (w - tb);
    const Words4 v = *reinterpret_cast<const Words4 *>(ptr);
    return make_uint4(v.x, v.y, v.z, v.w);
}

__global__ __launch_bounds__(256, 4) void k_scan_long(ScanParams p, uint64_t n, uint64_t n_words, uint64_t base_read_index,
                                                   uint32_t reads_per_wave,
                                                   const int32_t *__restrict__ a_tid,
                                                   const int32_t *__restrict__ a_pos,
                                                   const uint16_t *__restrict__ a_flag,
                                                   const uint8_t *__restrict__ a_xs,
                                                   const uint8_t *__restrict__ a_nh,
                                                   const uint32_t *__restrict__ a_off,
                                                   const uint32_t *__restrict__ a_nc,
                                                   const uint32_t *__restrict__ cigar,
                                                   const uint32_t *__restrict__ tail,
                                                   const uint32_t *__restrict__ rank_of_tid,
                                                   uint4 *__restrict__ items, uint4 *__restrict__ pays,
                                                   uint32_t *__restrict__ cnts, uint8_t *__restrict__ digs,
                                                   uint64_t capacity, Counters *counters) {
    __shared__ uint4 pool_item[4][LONG_POOL], pool_pay[4][LONG_POOL];
    __shared__ LongStage stage_s[4];
    __shared__ uint32_t kb_s[4][6];
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    uint4 *const my_item = pool_item[wave], *const my_pay = pool_pay[wave];
    LongStage &stage = stage_s[wave];
    KeyBits kb{0u, 0u, 0u, ~0u, ~0u, ~0u};
    unsigned long long my_err = IPX_ERRW_NONE;

    auto load4 = [&](uint64_t w) { return load_cigar4(cigar, tail, n_words, w); };

    const uint64_t gw = (uint64_t)blockIdx.x * 4 + wave;
    uint64_t next = gw * reads_per_wave;          // wave-uniform from here on
    uint64_t range_end = next + reads_per_wave;
    if (range_end > n) range_end = n;
    if (next > n) next = n;
    uint32_t staged = 0, staged_used = 0;         // alignments in the stage / already handed out
    uint32_t pool_n = 0;
    const uint32_t stripe = my_stripe(p, threadIdx.x >> 6);

    auto flush = [&]() {
        __threadfence_block();
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(append_counter(counters, p, stripe), (unsigned long long)pool_n);
        base = __shfl(base, 0);
        for (uint32_t k = lane; k < pool_n; k += 64) {
            const uint64_t idx = append_index(p, stripe, base + k);
            if (idx < capacity) {
                uint4 it = my_item[k];
                it.w = (uint32_t)idx;
                items[idx] = it;
                const uint4 py = my_pay[k];
                pays[idx] = py;
                cnts[idx] = IPX_CNT_ONE(py.x & 3u);
                digs[idx] = (uint8_t)ipx_digit_of(it, p.first);
                kb.o0 |= it.x; kb.o1 |= it.y; kb.o2 |= it.z;
                kb.a0 &= it.x; kb.a1 &= it.y; kb.a2 &= it.z;
            }
        }
        __threadfence_block();
        pool_n = 0;
    };

    // ---- per-lane state of the alignment being walked
    bool active = false, warm = false, pending = false, redo = false;
    bool fin = false;                             // the waiting junction is the alignment's last: it sits in the
                                                  // walk's own registers, not in the snapshot
    uint64_t my_i = 0;                            // alignment index in the batch
    uint32_t pos32 = 0, meta = 0, rank = 0;       // meta: strand | cat << 8 | read_rejected << 10
    uint32_t left = 0;                            // ops not yet applied
    uint32_t off0 = 0;                            // first word of the alignment's CIGAR
    uint64_t wpos = 0;                            // word index of ring[0].x at the top of the loop
    uint4 ring[LONG_RING];
#pragma unroll
    for (uint32_t k = 0; k < LONG_RING; ++k) ring[k] = make_uint4(0, 0, 0, 0);
    uint32_t state = ST_LIMBO;
    uint32_t tp = 0, as = 0, is = 0, ie = 0, ae = 0;          // offsets from the alignment's pos
    uint32_t lmm = 0, rmm = 0, lind = 0, rind = 0, cur_indel = 0, ordinal = 0;
    uint32_t s_as = 0, s_is = 0, s_ie = 0, s_ae = 0, s_l = 0, s_r = 0;   // snapshot of a finished junction

    // junction_qc, junctions_extractor.cc:392-397, on a snapshot (offsets are exact and ordered)
    auto snap_ok = [&]() {
        const uint32_t q_as = fin ? as : s_as, q_is = fin ? is : s_is, q_ie = fin ? ie : s_ie, q_ae = fin ? ae : s_ae;
        const uint32_t q_l = fin ? lmm + lind : s_l, q_r = fin ? rmm + rind : s_r;
        const uint32_t ilen = q_ie - q_is;
        return !(meta & 0x400u) && ilen >= p.min_intron_length && ilen <= p.max_intron_length &&
               (uint64_t)(q_is - q_as) >= (uint64_t)q_l + p.min_anchor_length &&
               (uint64_t)(q_ae - q_ie) >= (uint64_t)q_r + p.min_anchor_length;
    };

    // ---- one op per active lane, written as selects: branches on per-lane conditions cost more than
    // the few operations they would skip
    auto step = [&](uint32_t w) {
        const bool run = active && left != 0;
        const uint32_t op = w & 0xfu, len = w >> 4;
        // BAM_CIGAR_STR "MIDNSHP=XB": M0 I1 D2 N3 S4 H5 P6 =7 X8 B9; classes 1 = M/=/X, 2 = N, 4 = I/D, 8 = S/H
        const uint32_t cls = (uint32_t)(0x0000000110882441ull >> (op * 4u)) & 0xfu;
        const bool c_m = run && cls == 1u, c_n = run && cls == 2u, c_id = run && cls == 4u, c_sh = run && cls == 8u;   // one-hot
        const bool in_l = state == ST_LEFT, in_r = state == ST_RIGHT;
        // I/D :257-275: counters first, then the test
        lind += (c_id && in_l) ? len : 0u;
        rind += (c_id && in_r) ? len : 0u;
        const uint32_t indel2 = cur_indel + len;
        const bool over = c_id && indel2 > p.max_anchor_indel_size;
        // the op ends a junction: N :208-229, S/H :306-311, I/D beyond the limit
        const bool trig = in_r && (c_n || c_sh || over);
        const bool last = run && left == 1;                             // finish() :278-282 follows this op

        // ---- store waiting snapshots: when one is about to be overwritten or orphaned, or enough wait
        const unsigned long long pm = ipx_ballot(pending);
        if (pm && (ipx_ballot(pending && (trig || last || !active)) || __popcll(pm) >= (int)LONG_DRAIN)) {
            const bool pass = pending && snap_ok();
            const unsigned long long em = ipx_ballot(pass);
            if (pass) {
                const uint32_t slot = pool_n + __builtin_amdgcn_mbcnt_hi((uint32_t)(em >> 32),
                                                                         __builtin_amdgcn_mbcnt_lo((uint32_t)em, 0u));
                const unsigned long long ri = base_read_index + my_i;
                if (ordinal <= IPX_ORD_MAX_ORDINAL) {
                    ReadCtx c;
                    c.pos = 0; c.rank = rank; c.strand = meta & 0xffu; c.cat = (meta >> 8) & 3u; c.read_index = ri;
                    uint4 it, pay;
                    const Rec r{(int64_t)(fin ? as : s_as), (int64_t)(fin ? is : s_is), (int64_t)(fin ? ie : s_ie),
                                (int64_t)(fin ? ae : s_ae), 0, 0, 0, 0};
                    pack_record_rel(p, c, r, pos32, ordinal, it, pay);
                    my_item[slot] = it;
                    my_pay[slot] = pay;
                }
                if (++ordinal > IPX_ORD_MAX_ORDINAL) {
                    const unsigned long long e = (ri << IPX_ORD_SHIFT_READ) | (IPX_ERRK_TARGET << 8) | 1u;
                    my_err = e < my_err ? e : my_err;
                }
            }
            pending = false;
            fin = false;
            pool_n += (uint32_t)__popcll(em);
            if (pool_n > LONG_POOL - 64) flush();
        }
        s_as = trig ? as : s_as; s_is = trig ? is : s_is; s_ie = trig ? ie : s_ie; s_ae = trig ? ae : s_ae;
        s_l = trig ? lmm + lind : s_l; s_r = trig ? rmm + rind : s_r;
        pending = pending || trig;

        // ---- apply the op
        const uint32_t tpn = tp + len;
        const bool adv = c_m || c_n || (c_id && op == 2);               // M = X N D move along the reference
        const bool carry = adv && tpn < tp;
        const bool bad = run && cls == 0;
        if (ipx_ballot(carry || bad)) {                                    // rare, wave-uniform
            if (carry) {                                                 // offsets leave 32 bits: 64-bit re-run
                if (trig) pending = false;                               // this op's junction comes from the re-run too
                redo = true;
            } else if (bad) {                                            // :312-313: the exception ends the read
                const uint32_t ch = (op == 6) ? 'P' : (op == 9 ? 'B' : '?');
                const uint32_t ord_e = ordinal + ((pending && snap_ok()) ? 1u : 0u);
                const unsigned long long e = ((base_read_index + my_i) << IPX_ORD_SHIFT_READ) |
                                             ((unsigned long long)ord_e << IPX_ORD_SHIFT_ORDINAL) |
                                             (IPX_ERRK_CIGAR << 8) | ch;
                my_err = e < my_err ? e : my_err;
            }
        }
        const bool n_r = c_n && in_r;                                   // N closes a junction's right anchor ...
        const bool n_lr = c_n && (in_l || in_r);                        // ... and opens the next intron
        const bool m_limbo = c_m && state == ST_LIMBO, m_intron = c_m && state == ST_INTRON;
        const uint32_t as2 = n_r ? ie : (m_limbo ? tp : as);
        const uint32_t ie2 = n_lr ? tpn : (m_intron ? tp : ie);
        is = n_lr ? tp : is;
        as = as2;
        ie = ie2;
        ae = c_m ? tpn : ae;
        uint32_t lmm2 = n_r ? rmm : (m_limbo ? 0u : lmm);               // M from LIMBO zeroes the mismatch counters
        uint32_t rmm2 = (n_r || m_limbo) ? 0u : rmm;                    // only: indel counters keep their value
        lind = n_r ? rind : lind;
        rind = n_r ? 0u : rind;
        state = n_lr ? ST_INTRON : m_limbo ? ST_LEFT : m_intron ? ST_RIGHT : (over || c_sh) ? ST_LIMBO : state;
        const bool x = c_m && op == 8;                                  // state is LEFT or RIGHT after an M
        lmm2 += (x && state == ST_LEFT) ? len : 0u;
        rmm2 += (x && state == ST_RIGHT) ? len : 0u;
        lmm = lmm2;
        rmm = rmm2;
        cur_indel = (c_m || c_n) ? 0u : (c_id ? indel2 : cur_indel);
        tp = adv ? tpn : tp;
        // finish(): the alignment's last op leaves a right anchor open (never together with `trig`: an op
        // that ends a junction does not leave the walk in RIGHT).  The lane stops here, so the junction
        // stays where it is until the next drain -- which is the next step's, the lane being inactive
        fin = last && !carry && !bad && state == ST_RIGHT;
        pending = pending || fin;
        left -= run ? 1u : 0u;
        active = run && !last && !carry && !bad;
    };

    for (;;) {
        // ---- 64-bit re-run of alignments whose offsets left 32 bits (after their snapshot is stored)
        if (ipx_ballot(redo && !pending)) {
            if (redo && !pending) {
                ReadCtx c;
                c.pos = (int32_t)pos32; c.flag = 0; c.nhc = 0; c.rank = rank;
                c.strand = meta & 0xffu; c.cat = (meta >> 8) & 3u; c.read_rejected = (meta & 0x400u) != 0;
                c.read_index = base_read_index + my_i;
                (void)run_read<3>(p, c, cigar + a_off[my_i], a_nc[my_i], ordinal, capacity, items, pays, cnts, digs, kb,
                                  my_err, append_counter(counters, p, stripe), stripe);
                redo = false;
            }
        }
        // ---- hand out alignments
        const unsigned long long idle = ipx_ballot(!active && !warm && !pending && !redo);
        if (idle) {
            if (next >= range_end) {
                if (idle == ~0ull) break;
            } else {
                if (staged_used == staged) {
                    // stage the next 64 alignments: process_alignment gate (junctions_extractor.cc:533-536),
                    // strand, category, and the first chunks of the CIGAR
                    const uint64_t rem = range_end - next;
                    staged = rem < 64 ? (uint32_t)rem : 64u;
                    staged_used = 0;
                    if (lane < staged) {
                        const uint64_t i = next + lane;
                        const int32_t tid = a_tid[i];
                        uint32_t nc = a_nc[i];
                        const uint32_t flag = a_flag[i];
                        const unsigned long long ri = base_read_index + i;
                        if (nc <= 1 || tid < 0 || (flag & (F_QCFAIL | F_DUP))) nc = 0;
                        if (nc && (uint32_t)tid >= p.n_targets) {
                            const unsigned long long e = (ri << IPX_ORD_SHIFT_READ) | (IPX_ERRK_TARGET << 8);
                            my_err = e < my_err ? e : my_err;
                            nc = 0;
                        }
                        const uint32_t off = nc ? a_off[i] : 0u;
                        if (nc && (uint64_t)off + nc > n_words) {            // words outside the batch
                            const unsigned long long e = (ri << IPX_ORD_SHIFT_READ) | (IPX_ERRK_TARGET << 8) | 3u;
                            my_err = e < my_err ? e : my_err;
                            nc = 0;
                        }
                        stage.nc[lane] = nc;
                        if (nc) {
                            const uint32_t nhc = a_nh[i], xs = a_xs[i];
                            const uint32_t strand = (p.strandness != IPX_UNSTRANDED) ? strand_from_flag(flag, p.strandness)
                                                                                     : (xs ? xs : (uint32_t)'.');
                            const uint32_t rej = ((flag & F_SECONDARY) || (p.exclude_multi && nhc == 2)) ? 0x400u : 0u;
                            stage.pos[lane] = (uint32_t)a_pos[i];
                            stage.meta[lane] = strand | (read_category(flag, nhc) << 8) | rej;
                            stage.rank[lane] = p.rank_is_tid ? (uint32_t)tid : rank_of_tid[tid];
                            stage.off[lane] = off;
                        }
                    }
                    __threadfence_block();
                }
                const uint32_t avail = staged - staged_used;
                if (!active && !warm && !pending && !redo) {
                    const uint32_t e = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32),
                                                                 __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
                    if (e < avail) {
                        const uint32_t k = staged_used + e;
                        const uint32_t nc = stage.nc[k];
                        if (nc) {
                            my_i = next + e;
                            pos32 = stage.pos[k];
                            meta = stage.meta[k];
                            rank = stage.rank[k];
                            off0 = stage.off[k];
                            wpos = (uint64_t)off0 - 4 * LONG_RING;        // this turn's refills fetch chunks 0..LONG_RING-1
                            left = nc;
                            state = ST_LIMBO;
                            tp = as = is = ie = ae = 0;
                            lmm = rmm = lind = rind = cur_indel = ordinal = 0;
                            warm = true;
                        }
                    }
                }
                const uint32_t want = (uint32_t)__popcll(idle);
                const uint32_t took = want < avail ? want : avail;
                staged_used += took;
                next += took;
            }
        }
        // ---- 4 * LONG_RING ops; slot k is refilled LONG_RING - 1 chunks ahead of its next use
#pragma unroll
        for (uint32_t k = 0; k < LONG_RING; ++k) {
            const uint4 q = ring[k];
            step(q.x);
            step(q.y);
            step(q.z);
            step(q.w);
            // `left` counts from the first word of slot k + 1 now (from the alignment's first word if warm)
            const bool need = (active && left > 4 * (LONG_RING - 1)) || (warm && left > 4 * k);
            ring[k] = load4(need ? wpos + 4 * (k + LONG_RING) : (uint64_t)off0);
        }
        wpos += 4 * LONG_RING;
        active = active || warm;
        warm = false;
    }
    if (pool_n) flush();

    unsigned long long e = my_err;
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long o = __shfl_xor(e, d);
        e = o < e ? o : e;
    }
    if (lane == 0 && e != IPX_ERRW_NONE) atomicMin(&counters->err_word, e);

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

// ---- long CIGARs, third formulation: plain ops at a few instructions each, events on a fixed beat ----
// Nineteen ops in twenty of a long read only lengthen the anchor the walk is in (M = X, and I / D below the
// indel limit, while the state is LEFT or RIGHT; junctions_extractor.cc:232-275): they move tp / ae and add to
// one counter.  k_scan_long pays the whole state machine (~90 vector instructions) for each of them.  Here a
// TURN of the wave is one step of the full state machine, one step that knows the M after an N (or after a clip:
// the op that opens an anchor, ~25 instructions; lanes that are inside an anchor take a plain op in it), and
// LONG3_FAST "plain" steps (~30 instructions: no state, no snapshot, no drain).  A lane that meets anything else in
// a plain step (N, a clip, an indel run over the limit, its alignment's last op) stands still until the next
// turn's full step -- a few steps' wait once per junction (an earlier two-speed kernel let it wait for 40 other
// lanes and lost).
// Lanes advance at their own pace, so the CIGAR words go through a per-lane ring in LDS (32 words a lane, word-
// major: no bank conflicts; row = the word's index in the CIGAR array mod 32, so the ring holds two ALIGNED 64-byte
// pieces of the array).  A lane asks for its next piece -- four 16-byte loads issued back to back -- as soon as
// the piece two before it is used up; the loads land in registers and move to the ring at the top of the next
// turn.  The memory side is why: with a 16-byte load per lane every four ops (k_scan_long) each lane's 128-byte
// line is asked for eight times over ~30 us, and the lines of the 32 k lanes an XCD runs are exactly its 4 MB of
// L2 -- PMC: 24 GB from the fabric per 2 M alignments for 6.6 GB of CIGAR words, 4.5 TB/s, which is what bounds
// that kernel (profiles/long_occupancy_experiment.sh: fewer waves per CU = proportionally less traffic).  Here a
// line is asked for twice, once per half.
constexpr uint32_t LONG3_POOL = 96;          // records per wave pool (1.5 KB + 1.5 KB)
constexpr uint32_t LONG3_RINGW = 32;         // CIGAR words per lane in the LDS ring (two aligned pieces of 16)
constexpr uint32_t LONG3_DRAIN = 40;         // waiting snapshots that trigger a drain
constexpr uint32_t LONG3_FAST = 6;           // plain steps per turn: LONG3_FAST + 2 ops = half a piece, a piece arrives within a turn

// (a struct out of the argument block: member by member, the copy constructor does not take the constant address space)
template <class T>
__device__ __forceinline__ T load_arg(const __attribute__((address_space(4))) T *src) {
    T v;
    __builtin_memcpy(&v, src, sizeof(T));
    return v;
}
// k_scan_long3's ONE parameter (the struct is the signature: the kernarg segment is this struct, nothing mirrors
// anything).  The kernel holds more uniform values than a wave has scalar registers; what only the staging of new
// alignments, the packing of finished junctions and the pool's copy-out need is read from the kernarg segment at those
// points (fresh_long3_args(): the pointer is made opaque so that the scalar loads stay where they are written).
struct Long3Args {
    ScanParams p;
    uint64_t n, n_words, base_read_index;
    uint32_t reads_per_wave;
    const int32_t *a_tid, *a_pos;
    const uint16_t *a_flag;
    const uint8_t *a_xs, *a_nh;
    const uint32_t *a_off, *a_nc, *cigar, *tail, *rank_of_tid;
    uint4 *items, *pays;
    uint32_t *cnts;
    uint8_t *digs;
    uint64_t capacity;
    Counters *counters;
};
typedef const __attribute__((address_space(4))) Long3Args *Long3ArgsPtr;
__device__ __forceinline__ Long3ArgsPtr fresh_long3_args() {
    Long3ArgsPtr k = (Long3ArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(k));
    return k;
}

__global__ __launch_bounds__(256, 3) void k_scan_long3(const Long3Args args) {
    const ScanParams &p = args.p;
    const uint64_t n = args.n, n_words = args.n_words;
    const uint32_t reads_per_wave = args.reads_per_wave;
    const uint32_t *const __restrict__ cigar = args.cigar, *const __restrict__ tail = args.tail;
    static_assert(LONG3_FAST + 2 <= LONG3_RINGW / 4, "a piece asked for at the top of a turn must be there before the lane needs it");
    __shared__ uint4 pool_item[4][LONG3_POOL], pool_pay[4][LONG3_POOL];
    __shared__ LongStage stage_s[4];
    __shared__ uint32_t ring_s[4][LONG3_RINGW][64];   // per lane (column): its next CIGAR words, word w of the array in row w % 32
    __shared__ uint32_t kb_s[4][6];
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    uint4 *const my_item = pool_item[wave], *const my_pay = pool_pay[wave];
    LongStage &stage = stage_s[wave];
    uint32_t (*const my_ring)[64] = ring_s[wave];
    KeyBits kb{0u, 0u, 0u, ~0u, ~0u, ~0u};
    unsigned long long my_err = IPX_ERRW_NONE;

    auto load4 = [&](uint64_t w) { return load_cigar4(cigar, tail, n_words, w); };

    const uint64_t gw = (uint64_t)blockIdx.x * 4 + wave;
    uint64_t next = gw * reads_per_wave;          // wave-uniform from here on
    uint64_t range_end = next + reads_per_wave;
    if (range_end > n) range_end = n;
    if (next > n) next = n;
    uint32_t staged = 0, staged_used = 0;         // alignments in the stage / already handed out
    uint32_t pool_n = 0;
    const uint32_t stripe = my_stripe(p, threadIdx.x >> 6);

    auto flush = [&]() {
        __threadfence_block();
        const Long3ArgsPtr ka = fresh_long3_args();
        const ScanParams fp = load_arg(&ka->p);
        uint4 *const ka_items = ka->items, *const ka_pays = ka->pays;
        uint32_t *const ka_cnts = ka->cnts;
        uint8_t *const ka_digs = ka->digs;
        const uint64_t ka_capacity = ka->capacity;
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(append_counter(ka->counters, fp, stripe), (unsigned long long)pool_n);
        base = __shfl(base, 0);
        for (uint32_t k = lane; k < pool_n; k += 64) {
            const uint64_t idx = append_index(fp, stripe, base + k);
            if (idx < ka_capacity) {
                uint4 it = my_item[k];
                it.w = (uint32_t)idx;
                ka_items[idx] = it;
                const uint4 py = my_pay[k];
                ka_pays[idx] = py;
                ka_cnts[idx] = IPX_CNT_ONE(py.x & 3u);
                ka_digs[idx] = (uint8_t)ipx_digit_of(it, fp.first);
                kb.o0 |= it.x; kb.o1 |= it.y; kb.o2 |= it.z;
                kb.a0 &= it.x; kb.a1 &= it.y; kb.a2 &= it.z;
            }
        }
        __threadfence_block();
        pool_n = 0;
    };

    // ---- per-lane state of the alignment being walked
    bool active = false, pending = false, redo = false;
    bool fin = false;                             // the waiting junction is the alignment's last: it sits in the
                                                  // walk's own registers, not in the snapshot
    uint64_t my_i = 0;                            // alignment index in the batch
    uint32_t pos32 = 0, meta = 0, rank = 0;       // meta: strand | cat << 8 | read_rejected << 10
    uint32_t cw = 0, end_w = 0;                   // word (of the CIGAR array) of the next op to apply / one past the alignment's last
    uint32_t off0 = 0;                            // first word of the alignment's CIGAR
    uint32_t hf = 0;                              // next 16-word piece of the array to ask for; pieces before it are in the ring or used up
    uint4 land[4];                                // the piece in flight
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) land[k] = make_uint4(0, 0, 0, 0);
    bool in_flight = false;
    uint32_t state = ST_LIMBO;
    uint32_t tp = 0, as = 0, is = 0, ie = 0, ae = 0;          // offsets from the alignment's pos
    uint32_t lmm = 0, rmm = 0, lind = 0, rind = 0, cur_indel = 0, ordinal = 0;
    uint32_t s_as = 0, s_is = 0, s_ie = 0, s_ae = 0, s_l = 0, s_r = 0;   // snapshot of a finished junction
#ifdef IPX_LONG3_STATS
    uint32_t st_turns = 0, st_plain = 0, st_full = 0, st_entry = 0, st_active = 0, st_starved = 0;
#endif

    // junction_qc, junctions_extractor.cc:392-397, on a snapshot (offsets are exact and ordered)
    auto snap_ok = [&]() {
        const uint32_t q_as = fin ? as : s_as, q_is = fin ? is : s_is, q_ie = fin ? ie : s_ie, q_ae = fin ? ae : s_ae;
        const uint32_t q_l = fin ? lmm + lind : s_l, q_r = fin ? rmm + rind : s_r;
        const uint32_t ilen = q_ie - q_is;
        return !(meta & 0x400u) && ilen >= p.min_intron_length && ilen <= p.max_intron_length &&
               (uint64_t)(q_is - q_as) >= (uint64_t)q_l + p.min_anchor_length &&
               (uint64_t)(q_ae - q_ie) >= (uint64_t)q_r + p.min_anchor_length;
    };

    // QC, pack and store the waiting snapshots of the whole wave (pool -> record arrays when the pool fills up)
    auto drain = [&]() {
        const bool pass = pending && snap_ok();
        const unsigned long long em = ipx_ballot(pass);
        if (pass) {
            const uint32_t slot = pool_n + __builtin_amdgcn_mbcnt_hi((uint32_t)(em >> 32),
                                                                     __builtin_amdgcn_mbcnt_lo((uint32_t)em, 0u));
            const Long3ArgsPtr ka = fresh_long3_args();
            const unsigned long long ri = ka->base_read_index + my_i;
            if (ordinal <= IPX_ORD_MAX_ORDINAL) {
                ReadCtx c;
                c.pos = 0; c.rank = rank; c.strand = meta & 0xffu; c.cat = (meta >> 8) & 3u; c.read_index = ri;
                uint4 it, pay;
                const Rec r{(int64_t)(fin ? as : s_as), (int64_t)(fin ? is : s_is), (int64_t)(fin ? ie : s_ie),
                            (int64_t)(fin ? ae : s_ae), 0, 0, 0, 0};
                pack_record_rel(load_arg(&ka->p), c, r, pos32, ordinal, it, pay);
                my_item[slot] = it;
                my_pay[slot] = pay;
            }
            if (++ordinal > IPX_ORD_MAX_ORDINAL) {
                const unsigned long long e = (ri << IPX_ORD_SHIFT_READ) | (IPX_ERRK_TARGET << 8) | 1u;
                my_err = e < my_err ? e : my_err;
            }
        }
        pending = false;
        fin = false;
        pool_n += (uint32_t)__popcll(em);
        if (pool_n > LONG3_POOL - 64) flush();
    };

    // ---- the full state machine: one op for every lane with `go` (k_scan_long's step)
    auto step = [&](uint32_t w, bool go) {
        const uint32_t left = end_w - cw;
        const bool run = go && active && left != 0;
        const uint32_t op = w & 0xfu, len = w >> 4;
        // BAM_CIGAR_STR "MIDNSHP=XB": M0 I1 D2 N3 S4 H5 P6 =7 X8 B9; classes 1 = M/=/X, 2 = N, 4 = I/D, 8 = S/H
        const uint32_t cls = (uint32_t)(0x0000000110882441ull >> (op * 4u)) & 0xfu;
        const bool c_m = run && cls == 1u, c_n = run && cls == 2u, c_id = run && cls == 4u, c_sh = run && cls == 8u;   // one-hot
        const bool in_l = state == ST_LEFT, in_r = state == ST_RIGHT;
        // I/D :257-275: counters first, then the test
        lind += (c_id && in_l) ? len : 0u;
        rind += (c_id && in_r) ? len : 0u;
        const uint32_t indel2 = cur_indel + len;
        const bool over = c_id && indel2 > p.max_anchor_indel_size;
        // the op ends a junction: N :208-229, S/H :306-311, I/D beyond the limit
        const bool trig = in_r && (c_n || c_sh || over);
        const bool last = run && left == 1;                             // finish() :278-282 follows this op

        // ---- store waiting snapshots: when one is about to be overwritten or orphaned, or enough wait
        const unsigned long long pm = ipx_ballot(pending);
        if (pm && (ipx_ballot(pending && (trig || last || !active)) || __popcll(pm) >= (int)LONG3_DRAIN)) drain();
        s_as = trig ? as : s_as; s_is = trig ? is : s_is; s_ie = trig ? ie : s_ie; s_ae = trig ? ae : s_ae;
        s_l = trig ? lmm + lind : s_l; s_r = trig ? rmm + rind : s_r;
        pending = pending || trig;

        // ---- apply the op
        const uint32_t tpn = tp + len;
        const bool adv = c_m || c_n || (c_id && op == 2);               // M = X N D move along the reference
        const bool carry = adv && tpn < tp;
        const bool bad = run && cls == 0;
        if (ipx_ballot(carry || bad)) {                                    // rare, wave-uniform
            if (carry) {                                                 // offsets leave 32 bits: 64-bit re-run
                if (trig) pending = false;                               // this op's junction comes from the re-run too
                redo = true;
            } else if (bad) {                                            // :312-313: the exception ends the read
                const uint32_t ch = (op == 6) ? 'P' : (op == 9 ? 'B' : '?');
                const uint32_t ord_e = ordinal + ((pending && snap_ok()) ? 1u : 0u);
                const unsigned long long e = ((fresh_long3_args()->base_read_index + my_i) << IPX_ORD_SHIFT_READ) |
                                             ((unsigned long long)ord_e << IPX_ORD_SHIFT_ORDINAL) |
                                             (IPX_ERRK_CIGAR << 8) | ch;
                my_err = e < my_err ? e : my_err;
            }
        }
        const bool n_r = c_n && in_r;                                   // N closes a junction's right anchor ...
        const bool n_lr = c_n && (in_l || in_r);                        // ... and opens the next intron
        const bool m_limbo = c_m && state == ST_LIMBO, m_intron = c_m && state == ST_INTRON;
        const uint32_t as2 = n_r ? ie : (m_limbo ? tp : as);
        const uint32_t ie2 = n_lr ? tpn : (m_intron ? tp : ie);
        is = n_lr ? tp : is;
        as = as2;
        ie = ie2;
        ae = c_m ? tpn : ae;
        uint32_t lmm2 = n_r ? rmm : (m_limbo ? 0u : lmm);               // M from LIMBO zeroes the mismatch counters
        uint32_t rmm2 = (n_r || m_limbo) ? 0u : rmm;                    // only: indel counters keep their value
        lind = n_r ? rind : lind;
        rind = n_r ? 0u : rind;
        state = n_lr ? ST_INTRON : m_limbo ? ST_LEFT : m_intron ? ST_RIGHT : (over || c_sh) ? ST_LIMBO : state;
        const bool x = c_m && op == 8;                                  // state is LEFT or RIGHT after an M
        lmm2 += (x && state == ST_LEFT) ? len : 0u;
        rmm2 += (x && state == ST_RIGHT) ? len : 0u;
        lmm = lmm2;
        rmm = rmm2;
        cur_indel = (c_m || c_n) ? 0u : (c_id ? indel2 : cur_indel);
        tp = adv ? tpn : tp;
        // finish(): the alignment's last op leaves a right anchor open (never together with `trig`: an op
        // that ends a junction does not leave the walk in RIGHT).  The lane stops here, so the junction
        // stays where it is until the next drain -- which is the next full step's, the lane being inactive
        fin = go ? (last && !carry && !bad && state == ST_RIGHT) : fin;
        pending = pending || fin;
        cw += run ? 1u : 0u;
        active = go ? (run && !last && !carry && !bad) : active;
    };

    for (;;) {
        // ---- 64-bit re-run of alignments whose offsets left 32 bits (after their snapshot is stored)
        if (ipx_ballot(redo && !pending)) {
            if (redo && !pending) {
                ReadCtx c;
                c.pos = (int32_t)pos32; c.flag = 0; c.nhc = 0; c.rank = rank;
                c.strand = meta & 0xffu; c.cat = (meta >> 8) & 3u; c.read_rejected = (meta & 0x400u) != 0;
                const Long3ArgsPtr ka = fresh_long3_args();
                const ScanParams rp = load_arg(&ka->p);
                c.read_index = ka->base_read_index + my_i;
                (void)run_read<3>(rp, c, ka->cigar + ka->a_off[my_i], ka->a_nc[my_i], ordinal, ka->capacity, ka->items, ka->pays,
                                  ka->cnts, ka->digs, kb, my_err, append_counter(ka->counters, rp, stripe), stripe);
                redo = false;
            }
        }
        // ---- hand out alignments
        const unsigned long long idle = ipx_ballot(!active && !pending && !redo);
        if (idle) {
            if (next >= range_end) {
                if (idle == ~0ull) break;
            } else {
                if (staged_used == staged) {
                    // stage the next 64 alignments: process_alignment gate (junctions_extractor.cc:533-536),
                    // strand and category
                    const uint64_t rem = range_end - next;
                    staged = rem < 64 ? (uint32_t)rem : 64u;
                    staged_used = 0;
                    if (lane < staged) {
                        const Long3ArgsPtr ka = fresh_long3_args();
                        const int32_t *const a_tid = ka->a_tid, *const a_pos = ka->a_pos;
                        const uint16_t *const a_flag = ka->a_flag;
                        const uint8_t *const a_xs = ka->a_xs, *const a_nh = ka->a_nh;
                        const uint32_t *const a_off = ka->a_off, *const a_nc = ka->a_nc, *const rank_of_tid = ka->rank_of_tid;
                        const uint64_t i = next + lane;
                        const int32_t tid = a_tid[i];
                        uint32_t ncl = a_nc[i];
                        const uint32_t flag = a_flag[i];
                        const unsigned long long ri = ka->base_read_index + i;
                        if (ncl <= 1 || tid < 0 || (flag & (F_QCFAIL | F_DUP))) ncl = 0;
                        if (ncl && (uint32_t)tid >= p.n_targets) {
                            const unsigned long long e = (ri << IPX_ORD_SHIFT_READ) | (IPX_ERRK_TARGET << 8);
                            my_err = e < my_err ? e : my_err;
                            ncl = 0;
                        }
                        const uint32_t off = ncl ? a_off[i] : 0u;
                        if (ncl && (uint64_t)off + ncl > n_words) {          // words outside the batch
                            const unsigned long long e = (ri << IPX_ORD_SHIFT_READ) | (IPX_ERRK_TARGET << 8) | 3u;
                            my_err = e < my_err ? e : my_err;
                            ncl = 0;
                        }
                        stage.nc[lane] = ncl;
                        if (ncl) {
                            const uint32_t nhc = a_nh[i], xs = a_xs[i];
                            const uint32_t strand = (p.strandness != IPX_UNSTRANDED) ? strand_from_flag(flag, p.strandness)
                                                                                     : (xs ? xs : (uint32_t)'.');
                            const uint32_t rej = ((flag & F_SECONDARY) || (p.exclude_multi && nhc == 2)) ? 0x400u : 0u;
                            stage.pos[lane] = (uint32_t)a_pos[i];
                            stage.meta[lane] = strand | (read_category(flag, nhc) << 8) | rej;
                            stage.rank[lane] = p.rank_is_tid ? (uint32_t)tid : rank_of_tid[tid];
                            stage.off[lane] = off;
                        }
                    }
                    __threadfence_block();
                }
                const uint32_t avail = staged - staged_used;
                if (!active && !pending && !redo) {
                    const uint32_t e = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32),
                                                                 __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
                    if (e < avail) {
                        const uint32_t k = staged_used + e;
                        const uint32_t ncl = stage.nc[k];
                        if (ncl) {
                            my_i = next + e;
                            pos32 = stage.pos[k];
                            meta = stage.meta[k];
                            rank = stage.rank[k];
                            off0 = stage.off[k];
                            cw = off0;
                            end_w = off0 + ncl;
                            hf = off0 >> 4;
                            in_flight = false;
                            state = ST_LIMBO;
                            tp = as = is = ie = ae = 0;
                            lmm = rmm = lind = rind = cur_indel = ordinal = 0;
                            active = true;
                        }
                    }
                }
                const uint32_t want = (uint32_t)__popcll(idle);
                const uint32_t took = want < avail ? want : avail;
                staged_used += took;
                next += took;
            }
        }
        // ---- (1) the piece the landing registers took up a turn ago goes to its ring rows
        if (in_flight) {
            const uint32_t r0 = ((hf - 1u) & 1u) * 16u;
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k) {
                my_ring[r0 + 4 * k][lane] = land[k].x; my_ring[r0 + 4 * k + 1][lane] = land[k].y;
                my_ring[r0 + 4 * k + 2][lane] = land[k].z; my_ring[r0 + 4 * k + 3][lane] = land[k].w;
            }
        }
        const uint32_t have_w = hf * 16u;                          // words [cw, have_w) of the array are in the ring
        // ---- (2) the next piece, if the alignment reaches into it and the piece whose rows it takes is used up
        {
            const bool want = active && have_w < end_w && cw + 16u >= have_w;
            if (want) {
                if ((uint64_t)have_w + 16u <= n_words) {                     // the whole piece is inside the array (it is 16-byte aligned)
                    const uint4 *src = reinterpret_cast<const uint4 *>(cigar + have_w);
#pragma unroll
                    for (uint32_t k = 0; k < 4; ++k) land[k] = src[k];
                } else {
#pragma unroll
                    for (uint32_t k = 0; k < 4; ++k) {
                        const uint64_t w = (uint64_t)have_w + 4u * k;
                        land[k] = load4(w < n_words ? w : (uint64_t)off0);   // (a chunk past the array's end: any words will do)
                    }
                }
                hf += 1u;
            }
            in_flight = want;
        }
        // ---- (3) one full step: whatever made a lane stop in the last turn's plain steps (for the others, one more op)
        {
            const uint32_t w = my_ring[cw & (LONG3_RINGW - 1u)][lane];
            const bool go = active && cw < have_w;
#ifdef IPX_LONG3_STATS
            st_full += __popcll(ipx_ballot(go)); st_starved += __popcll(ipx_ballot(active && !go));
#endif
            step(w, go);
        }
        // the ops the lane may take without the full step this turn: words in the ring, never the alignment's last
        // op (finish() follows it), and only while the turn's ops cannot carry out of 32 bits (lengths are 28 bits)
        const uint32_t lim1 = (active && tp < 0x40000000u) ? (have_w < end_w - 1u ? have_w : end_w - 1u) : 0u;
        // ---- (4) the op after an N (or after a clip, or the alignment's first): M = X from INTRON opens the right
        // anchor, from LIMBO the left one (:232-255) -- k_scan_long's step for exactly this case.  Lanes that are in
        // LEFT / RIGHT (three in four: they did not stop) take a plain op in the same step
        uint32_t ind_acc = 0, mm_acc = 0;                           // what the plain ops add to the current side's counters
        // An op's classes as masks (all ones / zero), one v_bfe_i32 each: bit `op` of a 16-bit pattern (BAM's "MIDNSHP=X":
        // M 0, I 1, D 2, = 7, X 8) -- the pattern stands twice in the word, so the WHOLE CIGAR word can be the bit offset
        // (the instruction reads its bits 4:0; bit 4 is the length's lowest).  A plain op is one with m_adv | m_id.
        // (Rounds 2-4 shifted a 64-bit table of four-bit codes by 4 * op and took the code apart: ten instructions for
        // these four, and the steps below hung on a branch each)
        constexpr uint32_t K_ADV = 0x01850185u,     // moves along the reference: M D = X
                           K_ID = 0x00060006u,      // indel: I D
                           K_M = 0x01810181u,       // M = X
                           K_X = 0x01000100u;       // X
        {
            const uint32_t w = my_ring[cw & (LONG3_RINGW - 1u)][lane];
            const uint32_t len = w >> 4;
            const uint32_t m_adv = (uint32_t)__builtin_amdgcn_sbfe((int)K_ADV, w, 1), m_id = (uint32_t)__builtin_amdgcn_sbfe((int)K_ID, w, 1),
                           m_m = (uint32_t)__builtin_amdgcn_sbfe((int)K_M, w, 1), m_x = (uint32_t)__builtin_amdgcn_sbfe((int)K_X, w, 1);
            const bool may = cw < lim1 && (m_adv | m_id) != 0u;
            const bool ok_e = may && m_m != 0u && (state & (ST_LIMBO | ST_INTRON)) != 0u;
            const uint32_t il = len & m_id;
            const uint32_t ci = (cur_indel + il) & ~m_m;
            const bool ok_p = may && (state & (ST_LEFT | ST_RIGHT)) != 0u && ci <= p.max_anchor_indel_size;
            if (ok_e) {
                const bool from_limbo = state == ST_LIMBO;
                const uint32_t xl = len & m_x;
                as = from_limbo ? tp : as;
                ie = from_limbo ? ie : tp;
                lmm = from_limbo ? xl : lmm;                            // M from LIMBO zeroes the mismatch counters only
                rmm = from_limbo ? 0u : rmm + xl;
                state = from_limbo ? ST_LEFT : ST_RIGHT;
                tp += len;
                ae = tp;
                cur_indel = 0;
                cw += 1u;
            }
            if (ok_p) {
                tp += len & m_adv;
                ae = (tp & m_m) | (ae & ~m_m);
                cur_indel = ci;
                ind_acc = il;
                mm_acc = len & m_x;
                cw += 1u;
            }
#ifdef IPX_LONG3_STATS
            st_entry += __popcll(ipx_ballot(ok_e)); st_plain += __popcll(ipx_ballot(ok_p));
#endif
        }
        // ---- (5) plain steps: only from LEFT / RIGHT, which no plain op leaves
        {
            const bool lr = (state & (ST_LEFT | ST_RIGHT)) != 0u;
            const uint32_t lim = lr ? lim1 : 0u;
            const uint32_t m_l = state == ST_LEFT ? ~0u : 0u;
            // The words of all the plain steps are read up front, from cw on: a step that does not apply leaves cw where it
            // is, and so do all the steps behind it (the same word fails the same way), so step k applies exactly if all
            // before it did (`ok` runs through the steps) -- with the word read at the step, from an address that depends on
            // the step before, a turn was six dependent trips to LDS (round 5).  The steps are straight-line code: what an
            // op adds is masked by its classes and by `ok`
            const uint32_t cw0 = cw;
            uint32_t wk[LONG3_FAST];
#pragma unroll
            for (uint32_t k = 0; k < LONG3_FAST; ++k) wk[k] = my_ring[(cw0 + k) & (LONG3_RINGW - 1u)][lane];
            const uint32_t nmax = lim > cw0 ? lim - cw0 : 0u;            // plain steps the lane may take this turn
            bool ok = true;
#pragma unroll
            for (uint32_t k = 0; k < LONG3_FAST; ++k) {
                const uint32_t w = wk[k];
                const uint32_t len = w >> 4;
                const uint32_t m_adv = (uint32_t)__builtin_amdgcn_sbfe((int)K_ADV, w, 1), m_id = (uint32_t)__builtin_amdgcn_sbfe((int)K_ID, w, 1),
                               m_m = (uint32_t)__builtin_amdgcn_sbfe((int)K_M, w, 1), m_x = (uint32_t)__builtin_amdgcn_sbfe((int)K_X, w, 1);
                const uint32_t ci = (cur_indel + (len & m_id)) & ~m_m;   // I/D :257-275 add up, M = X :232-255 start over
                // a step that does not apply stops all the steps behind it
                ok = ok && k < nmax && (m_adv | m_id) != 0u && ci <= p.max_anchor_indel_size;
                const uint32_t le = ok ? len : 0u;
                tp += le & m_adv;
                ae = (ok && m_m != 0u) ? tp : ae;
                cur_indel = ok ? ci : cur_indel;
                ind_acc += le & m_id;
                mm_acc += le & m_x;
                cw += ok ? 1u : 0u;
#ifdef IPX_LONG3_STATS
                st_plain += __popcll(ipx_ballot(ok));
#endif
            }
            lind += ind_acc & m_l; rind += ind_acc & ~m_l;
            lmm += mm_acc & m_l; rmm += mm_acc & ~m_l;
        }
#ifdef IPX_LONG3_STATS
        st_turns++; st_active += __popcll(ipx_ballot(active));
#endif
    }
    if (pool_n) flush();
#ifdef IPX_LONG3_STATS
    if (lane == 0 && (gw % 997) == 0)
        printf("wave %llu: turns %u plain-ops %u full-ops %u entry-ops %u active-lanes %u starved %u\n",
               (unsigned long long)gw, st_turns, st_plain, st_full, st_entry, st_active, st_starved);
#endif

    unsigned long long e = my_err;
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long o = __shfl_xor(e, d);
        e = o < e ? o : e;
    }
    Counters *const counters_ = fresh_long3_args()->counters;
    if (lane == 0 && e != IPX_ERRW_NONE) atomicMin(&counters_->err_word, e);

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
            if (o) atomicOr(&counters_->key_or[t], o);
        } else {
            const uint32_t a = x0 & x1 & x2 & x3;
            if (a != ~0u) atomicAnd(&counters_->key_and[t - 3], a);
        }
    }
}

// ---- short CIGARs: what k_scan_short3 (ipx_scan_short.inc) and k_scan_rest share ---------------------------------
// (rounds 1-4's k_scan_short -- a 512-alignment tile per block, an LDS pool behind block barriers, the hash-table
// combiner -- is in the history at 0be155e; ipx_scan_short.inc says what replaced it and why)
struct AlnArrays {
    const int32_t *__restrict__ tid;
    const int32_t *__restrict__ pos;
    const uint16_t *__restrict__ flag;
    const uint8_t *__restrict__ xs;
    const uint8_t *__restrict__ nh;
    const uint32_t *__restrict__ off;
    const uint32_t *__restrict__ nc;
    const uint32_t *__restrict__ rank_of_tid;
    uint64_t n, n_words, base_read_index;
};

// An alignment's seven fields as they lie in the batch.  They are fetched for every lane, from a clamped index, in ONE
// round trip: written as `if (i < n) { ...; if (passes the gate) { pos, off, nh, xs } }` the compiler sinks each load
// into the branch that uses it and waits at the branch's end -- three dependent round trips per alignment and round
// (flag nc tid / pos off nh xs / rank), six per tile, which was what k_scan_short's tiles spent their first 15 k cycles
// on.  pin_raw() (an empty asm that names the registers) is where the loads of all rounds are waited for, together.
struct RawAln {
    int32_t tid, pos;
    uint32_t flag, nc, off, nh, xs;
};
__device__ __forceinline__ RawAln load_raw_aln(const AlnArrays &a, uint64_t i_clamped) {
    RawAln r;
    r.tid = a.tid[i_clamped];
    r.flag = a.flag[i_clamped];
    r.nc = a.nc[i_clamped];
    r.pos = a.pos[i_clamped];
    r.off = a.off[i_clamped];
    r.nh = a.nh[i_clamped];
    r.xs = a.xs[i_clamped];
    return r;
}
__device__ __forceinline__ void pin_raw(RawAln &r) {
    asm volatile("" : "+v"(r.tid), "+v"(r.pos), "+v"(r.flag), "+v"(r.nc), "+v"(r.off), "+v"(r.nh), "+v"(r.xs));
}
// the index rank_of_tid is read at (any lane may read: a target that exists)
__device__ __forceinline__ uint32_t rank_index(const ScanParams &p, const RawAln &r) {
    return (uint32_t)r.tid < p.n_targets ? (uint32_t)r.tid : 0u;
}

// the per-read gate (junctions_extractor.cc:536), strand and category of alignment i from its fields; nc = 0: nothing
// to walk.  rank: rank_of_tid[tid] (or tid itself in target scope)
__device__ __forceinline__ void read_ctx_of(const ScanParams &p, const AlnArrays &a, uint64_t i, const RawAln &raw, uint32_t rank,
                                            ReadCtx &c, uint32_t &nc, uint32_t &off, unsigned long long &err) {
    c.pos = 0; c.flag = 0; c.nhc = 0; c.rank = 0; c.strand = '.'; c.read_index = a.base_read_index + i;
    nc = 0; off = 0;
    if (i < a.n) {
        const int32_t tid = raw.tid;
        c.flag = raw.flag;
        nc = raw.nc;
        // junctions_extractor.cc:536
        if (nc <= 1 || tid < 0 || (c.flag & (F_QCFAIL | F_DUP))) nc = 0;
        if (nc) {
            if ((uint32_t)tid >= p.n_targets) {
                const unsigned long long e = (c.read_index << IPX_ORD_SHIFT_READ) | (IPX_ERRK_TARGET << 8);
                err = e < err ? e : err;
                nc = 0;
            } else {
                c.pos = raw.pos;
                off = raw.off;
                if ((uint64_t)off + nc > a.n_words) {          // words outside the batch: never dereferenced
                    const unsigned long long e = (c.read_index << IPX_ORD_SHIFT_READ) | (IPX_ERRK_TARGET << 8) | 3u;
                    err = e < err ? e : err;
                    nc = 0;
                }
                c.nhc = raw.nh;
                c.rank = rank;
                c.strand = (p.strandness != IPX_UNSTRANDED) ? strand_from_flag(c.flag, p.strandness)
                                                            : (raw.xs ? raw.xs : (uint32_t)'.');
            }
        }
    }
    c.cat = read_category(c.flag, c.nhc);
    c.read_rejected = (c.flag & F_SECONDARY) || (p.exclude_multi && c.nhc == 2);
}

// one alignment (a.n != 0)
__device__ __forceinline__ void load_read_ctx(const ScanParams &p, const AlnArrays &a, uint64_t i, ReadCtx &c,
                                              uint32_t &nc, uint32_t &off, unsigned long long &err) {
    RawAln raw = load_raw_aln(a, i < a.n ? i : a.n - 1);
    pin_raw(raw);
    uint32_t rank = (uint32_t)raw.tid;
    if (!p.rank_is_tid && p.n_targets) rank = a.rank_of_tid[rank_index(p, raw)];
    read_ctx_of(p, a, i, raw, rank, c, nc, off, err);
}

__device__ __forceinline__ uint32_t *defer_counter(Counters *c) {
    return reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(c) + IPX_CNT_DEFER);
}

#include "ipx_scan_short.inc"

// ---- what k_scan_short3 leaves out -----------------------------------------------------------------
// k_scan_rest: the alignments k_scan_short3 listed (more than 8 CIGAR ops in a batch of short ones), one lane each,
// op by op: count (run_read<0>), ONE reservation per wave, store (run_read<1>).  It reads the list's length on the
// device and leaves at once if it is zero; its grid follows what the previous batch listed (see the launch below).
__global__ __launch_bounds__(256) void k_scan_rest(ScanParams p, AlnArrays a, const uint32_t *__restrict__ cigar,
                                                   const uint32_t *__restrict__ defer,
                                                   uint4 *__restrict__ items, uint4 *__restrict__ pays,
                                                   uint32_t *__restrict__ cnts, uint8_t *__restrict__ digs,
                                                   uint64_t capacity, Counters *counters) {
    const uint32_t n_def = *defer_counter(counters);
    if (n_def == 0) return;
    __shared__ uint32_t kb_s[4][6];
    KeyBits kb{0u, 0u, 0u, ~0u, ~0u, ~0u};
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const uint32_t stripe = my_stripe(p, wave);
    unsigned long long my_err = IPX_ERRW_NONE;
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t e0 = ((uint64_t)blockIdx.x * 4 + wave) * 64; e0 < n_def; e0 += stride) {
        const bool mine = e0 + lane < n_def;
        const uint64_t i = mine ? defer[e0 + lane] : 0u;
        ReadCtx c;
        uint32_t nc = 0, off = 0;
        load_read_ctx(p, a, i, c, nc, off, my_err);
        if (!mine) nc = 0;
        const uint32_t cn = nc ? run_read<0>(p, c, cigar + off, nc, 0, 0, nullptr, nullptr, nullptr, nullptr, kb, my_err) : 0;
        uint32_t incl = cn;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t x = __shfl_up(incl, o);
            if ((int)lane >= o) incl += x;
        }
        const uint32_t total = __shfl(incl, 63);
        unsigned long long slot = 0;
        if (lane == 63 && total) slot = atomicAdd(append_counter(counters, p, stripe), (unsigned long long)total);
        slot = __shfl(slot, 63);
        if (cn) {
            unsigned long long dummy = IPX_ERRW_NONE;
            (void)run_read<1>(p, c, cigar + off, nc, slot + (incl - cn), capacity, items, pays, cnts, digs, kb, dummy,
                              nullptr, stripe);
        }
    }
    unsigned long long e = my_err;
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long o = __shfl_xor(e, d);
        e = o < e ? o : e;
    }
    if (lane == 0 && e != IPX_ERRW_NONE) atomicMin(&counters->err_word, e);
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
            const uint32_t a4 = x0 & x1 & x2 & x3;
            if (a4 != ~0u) atomicAnd(&counters->key_and[t - 3], a4);
        }
    }
}

// ---- closing the gaps of a striped launch ---------------------------------------------------------
// After a striped scan the launch's records sit in IPX_STRIPES interleaved stripes behind `base`, every stripe filled up
// to its own count: all positions below the shortest stripe's last full round of chunks are taken, the rest of the
// range -- the WINDOW -- has holes.  k_close moves the records at or above base + n into the holes below it (equally
// many), publishes base + n as the record count and zeroes the stripe counters for the next launch.  Rounds 1-4 did
// this with three kernels that scanned the window position by position (find / move / finish, 16 us).  But the holes
// have a closed form: a chunk of the window is taken up to its stripe's count and free behind it, so a block needs only
// a prefix sum over the window's CHUNKS (<= two per thread) to know where the i-th hole and the i-th record to move
// lie; every block works that out for itself and moves its share of the pairs, nothing is listed, and the block that
// finishes last (a counter) publishes.  A launch that ran out of capacity publishes the extent it would have needed
// instead (the caller grows the arrays and repeats the scan); a window of more than CLOSE_MAX_CHUNKS chunks (stripes
// that filled very unevenly) raises Counters::pad[1] and the caller repeats the scan without stripes.
constexpr uint32_t CLOSE_BT = 1024, CLOSE_BLOCKS = 64;
constexpr uint32_t CLOSE_MAX_CHUNKS = 2 * CLOSE_BT;
constexpr uint32_t CHUNK = 1u << IPX_STRIPE_CHUNK_BITS;

__global__ __launch_bounds__(CLOSE_BT) void k_close(Counters *counters, uint64_t base, uint64_t capacity,
                                                    uint4 *__restrict__ items, uint4 *__restrict__ pays,
                                                    uint32_t *__restrict__ cnts, uint8_t *__restrict__ digs) {
    __shared__ unsigned long long cnt_s[IPX_STRIPES];
    __shared__ uint32_t hpre[CLOSE_MAX_CHUNKS + 1], tpre[CLOSE_MAX_CHUNKS + 1];      // holes / records to move before chunk c
    __shared__ uint32_t wsum_h[CLOSE_BT / 64], wsum_t[CLOSE_BT / 64];
    __shared__ uint32_t last_s;
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    if (t < IPX_STRIPES)
        cnt_s[t] = *reinterpret_cast<const unsigned long long *>(reinterpret_cast<const char *>(counters) + IPX_CNT_STRIPE0 + 128 * t);
    __syncthreads();
    unsigned long long n = 0, extent = 0, rounds = ~0ull;
    for (uint32_t k = 0; k < IPX_STRIPES; ++k) {
        const unsigned long long c = cnt_s[k];
        n += c;
        if (c) {
            const unsigned long long last = c - 1;
            const unsigned long long end = ((((last >> IPX_STRIPE_CHUNK_BITS) * IPX_STRIPES + k) << IPX_STRIPE_CHUNK_BITS) | (last & (CHUNK - 1))) + 1;
            extent = end > extent ? end : extent;
        }
        const unsigned long long full = c >> IPX_STRIPE_CHUNK_BITS;
        rounds = full < rounds ? full : rounds;
    }
    const unsigned long long window_lo = (rounds * IPX_STRIPES) << IPX_STRIPE_CHUNK_BITS;
    // (block-uniform, like everything computed from cnt_s)
    const bool no_room = base + extent > capacity;
    const uint64_t n_chunks = (extent - window_lo + CHUNK - 1) >> IPX_STRIPE_CHUNK_BITS;
    const bool too_wide = !no_room && extent != n && n_chunks > CLOSE_MAX_CHUNKS;
    if (!no_room && !too_wide && extent != n) {
        // chunk c of the window: positions P0 .. P0 + CHUNK, taken up to P0 + tk
        uint32_t nh[2], nt[2];
#pragma unroll
        for (uint32_t e = 0; e < 2; ++e) {
            const uint64_t c = (uint64_t)t * 2 + e;
            nh[e] = nt[e] = 0;
            if (c < n_chunks) {
                const uint64_t P0 = window_lo + (c << IPX_STRIPE_CHUNK_BITS);
                const uint64_t r = rounds + c / IPX_STRIPES, L0 = r << IPX_STRIPE_CHUNK_BITS;
                const unsigned long long ck = cnt_s[c % IPX_STRIPES];
                const uint64_t tk = ck <= L0 ? 0 : (ck - L0 >= CHUNK ? CHUNK : ck - L0);
                const uint64_t free_lo = P0 + tk, free_hi = P0 + CHUNK < n ? P0 + CHUNK : n;       // free positions below n
                nh[e] = free_hi > free_lo ? (uint32_t)(free_hi - free_lo) : 0u;
                const uint64_t mv_lo = P0 > n ? P0 : n;                                              // taken positions at or above n
                nt[e] = P0 + tk > mv_lo ? (uint32_t)(P0 + tk - mv_lo) : 0u;
            }
        }
        uint32_t ih = nh[0] + nh[1], it = nt[0] + nt[1];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t xh = __shfl_up(ih, o), xt = __shfl_up(it, o);
            if ((int)lane >= o) { ih += xh; it += xt; }
        }
        if (lane == 63) { wsum_h[wave] = ih; wsum_t[wave] = it; }
        __syncthreads();
        uint32_t bh = 0, bt = 0, n_pairs = 0;
        for (uint32_t w = 0; w < CLOSE_BT / 64; ++w) {
            if (w < wave) { bh += wsum_h[w]; bt += wsum_t[w]; }
            n_pairs += wsum_h[w];
        }
        const uint32_t eh = bh + ih - nh[0] - nh[1], et = bt + it - nt[0] - nt[1];
        hpre[2 * t] = eh; hpre[2 * t + 1] = eh + nh[0];
        tpre[2 * t] = et; tpre[2 * t + 1] = et + nt[0];
        if (t == CLOSE_BT - 1) { hpre[CLOSE_MAX_CHUNKS] = eh + nh[0] + nh[1]; tpre[CLOSE_MAX_CHUNKS] = et + nt[0] + nt[1]; }
        __syncthreads();
        // the i-th hole, the i-th record to move: the last chunk whose prefix is <= i
        auto chunk_of = [&](const uint32_t *pre, uint32_t i) {
            uint32_t lo = 0, hi = CLOSE_MAX_CHUNKS;                        // pre[lo] <= i < pre[hi]
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (pre[mid] <= i) lo = mid; else hi = mid;
            }
            return lo;
        };
        for (uint32_t i = blockIdx.x * CLOSE_BT + t; i < n_pairs; i += gridDim.x * CLOSE_BT) {
            const uint32_t ch = chunk_of(hpre, i), ct = chunk_of(tpre, i);
            uint64_t dst, src;
            {
                const uint64_t P0 = window_lo + ((uint64_t)ch << IPX_STRIPE_CHUNK_BITS);
                const uint64_t L0 = (rounds + ch / IPX_STRIPES) << IPX_STRIPE_CHUNK_BITS;
                const unsigned long long ck = cnt_s[ch % IPX_STRIPES];
                const uint64_t tk = ck <= L0 ? 0 : (ck - L0 >= CHUNK ? CHUNK : ck - L0);
                dst = base + P0 + tk + (i - hpre[ch]);
            }
            {
                const uint64_t P0 = window_lo + ((uint64_t)ct << IPX_STRIPE_CHUNK_BITS);
                src = base + (P0 > n ? P0 : n) + (i - tpre[ct]);
            }
            uint4 v = items[src];
            v.w = (v.w & IPX_ITEM_FOLDED) | (uint32_t)dst;
            items[dst] = v;
            pays[dst] = pays[src];
            cnts[dst] = cnts[src];
            digs[dst] = digs[src];
        }
    }
    // the block that finishes last publishes and resets (every block has read the stripe counters by then)
    __syncthreads();
    unsigned int *const done = reinterpret_cast<unsigned int *>(reinterpret_cast<char *>(counters) + IPX_CNT_CLOSE_DONE);
    if (t == 0) last_s = (atomicAdd(done, 1u) == gridDim.x - 1) ? 1u : 0u;
    __syncthreads();
    if (!last_s) return;
    if (t == 0) {
        counters->n_records = no_room ? base + extent : base + n;            // out of room: what it would take
        if (too_wide) counters->pad[1] = 1;                                  // repeat without stripes
        *done = 0;
    }
    if (t < IPX_STRIPES)
        *reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(counters) + IPX_CNT_STRIPE0 + 128 * t) = 0;
}

}  // namespace

hipError_t ipx_launch_scan(hipStream_t s, ScanParams p, const ipx_batch &b,
                           const uint32_t *rank_of_tid, uint4 *items, uint4 *pays, uint32_t *cnts, uint8_t *digs,
                           uint64_t capacity, Counters *counters, uint64_t n_records_before,
                           unsigned long long *close_ws, uint64_t close_cap, uint32_t *defer, uint32_t rest_hint) {
    if (b.n_alignments == 0) return hipSuccess;
    // Which kernel: the short-CIGAR kernel walks at most 8 ops per alignment itself and hands longer ones to a
    // slow second stage, so it is for batches where those are rare.  Measured on 2 M random CIGARs
    // (profiles/mid_cigar_timing.py): 2-8 ops short 0.12 ms / long 0.16 ms; 4-12 ops 0.37 / 0.18; 6-16 ops
    // 0.40 / 0.20; 10-20 ops 0.44 / 0.21 -> up to 6 words per alignment on average go to the short kernel.
    // (the long kernel needs >= 8 words in the array)
    const char *sma = getenv("IPX_SHORT_MAX_AVG");             // tests route either way
    const long sma_v = sma ? atol(sma) : 0;
    const uint64_t short_max_avg = (uint64_t)(sma_v >= 1 && sma_v <= 16 ? sma_v : 6);
    const bool short_reads = b.n_cigar_words <= short_max_avg * b.n_alignments || b.n_cigar_words < 8;
    static const uint32_t n_cus = [] {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        return (uint32_t)cus;
    }();
    uint32_t *tail = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(counters) + IPX_CNT_TAIL);
    ipx_prof_pre(s, IPXK_SCAN);
    hipLaunchKernelGGL(k_cigar_tail, dim3(1), dim3(64), 0, s, b.cigar, b.n_cigar_words, tail,
                       reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(counters) + IPX_CNT_DEFER),
                       reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(counters) + IPX_CNT_MERGED));
    // stripes pay once many blocks queue on the record counter
    p.base = n_records_before;
    p.n_stripes = 0;
    if (short_reads) {
        // k_scan_short3 (ipx_scan_short.inc): a wave per 128 alignments, six blocks of four waves per CU
        uint64_t blocks3 = ipx_div_up(b.n_alignments, (uint64_t)S3_WT * S3_NW);
        const uint64_t resident3 = (uint64_t)n_cus * 6u;
        if (blocks3 > resident3) blocks3 = resident3;
        p.n_stripes = (blocks3 >= 64 && close_ws && close_cap) ? IPX_STRIPES : 0;
        Short3Args a{};
        a.p = p;
        a.n = b.n_alignments; a.n_words = b.n_cigar_words; a.base_read_index = b.base_read_index;
        a.a_tid = b.tid; a.a_pos = b.pos; a.a_flag = b.flag; a.a_xs = b.xs; a.a_nh = b.nh; a.a_off = b.cigar_off; a.a_nc = b.n_cigar;
        a.cigar = b.cigar; a.cigar8 = b.n_cigar_words >= 8 ? b.cigar : tail; a.tail = tail; a.rank_of_tid = rank_of_tid;
        a.items = items; a.pays = pays; a.cnts = cnts; a.digs = digs; a.capacity = capacity; a.counters = counters;
        a.defer = defer;
        a.comb_on = p.combine;
        if (p.combine == 2u) {
            hipLaunchKernelGGL((k_scan_short3<1>), dim3((uint32_t)blocks3), dim3(64 * S3_NW), 0, s, a);
        } else {
            hipLaunchKernelGGL((k_scan_short3<0>), dim3((uint32_t)blocks3), dim3(64 * S3_NW), 0, s, a);
        }
    }
    else {
        // long CIGARs: waves work through ranges of >= 64 alignments
        constexpr uint32_t min_range = 64;
        const char *lv = getenv("IPX_LONG");
        // k_scan_long3 reads aligned 64-byte pieces of the CIGAR array and counts its words in 32 bits
        const bool v3 = !(lv && lv[0] == '1') && ((uintptr_t)b.cigar & 15u) == 0 && b.n_cigar_words < (1ull << 31);
        const uint64_t max_waves = 256ull * 4 * (v3 ? 3 : 4);      // all resident at once (128 VGPRs: 4 waves per SIMD; k_scan_long3: 3, by its LDS)
        uint64_t per_wave = ipx_div_up(b.n_alignments, max_waves);
        if (per_wave < min_range) per_wave = min_range;
        // few alignments per lane: whole multiples of 64, or one lane's second alignment doubles the wave's time
        if (per_wave < 256) per_wave = (per_wave + 63) / 64 * 64;
        const uint64_t waves = ipx_div_up(b.n_alignments, per_wave);
        p.n_stripes = (waves >= 256 && close_ws && close_cap) ? IPX_STRIPES : 0;
        // IPX_LONG=1: k_scan_long instead of k_scan_long3 (tests run both).  Round 2 also built a two-speed walk
        // (k_scan_long2: chunks of four plain ops on a fast path, lanes that met anything else waited for 40 others;
        // exact, 6.5 ms against 5.1 at 2 M alignments) -- DESIGN.md section 4 has the measurement, the history has the code
        // (IPX_LONG_PAD, the round-2 occupancy experiment's extra LDS per block, retired in round 4: NOTEBOOK.md)
        if (v3) {
            const Long3Args la{p, b.n_alignments, b.n_cigar_words, b.base_read_index, (uint32_t)per_wave, b.tid, b.pos, b.flag, b.xs,
                               b.nh, b.cigar_off, b.n_cigar, b.cigar, tail, rank_of_tid, items, pays, cnts, digs, capacity, counters};
            hipLaunchKernelGGL(k_scan_long3, dim3((uint32_t)ipx_div_up(waves, 4)), dim3(256), 0, s, la);
        }
        else
            hipLaunchKernelGGL(k_scan_long, dim3((uint32_t)ipx_div_up(waves, 4)), dim3(256), 0, s, p, b.n_alignments,
                               b.n_cigar_words, b.base_read_index, (uint32_t)per_wave, b.tid, b.pos, b.flag, b.xs, b.nh,
                               b.cigar_off, b.n_cigar, b.cigar, tail, rank_of_tid, items, pays, cnts, digs, capacity, counters);
    }
    if (short_reads) {
        // Second stage, in the same breath: the alignments k_scan_short3 only listed (more than 8 ops) are walked
        // op by op by k_scan_rest, which reads the list's length on the
        // device and leaves at once if it is zero.  Its grid follows what the previous batch listed (rest_hint):
        // round 1 had the host read the count back and launch it with a fitting grid -- a second read-back and a
        // second closing of the stripes on every batch that listed anything (S: 35 us of a 0.74 ms step for 512
        // alignments).  A batch that lists far more than its predecessor runs the stage on too small a grid
        // (slower, not wrong) and corrects the hint for the next one.
        const uint32_t rest_blocks = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(ipx_div_up((uint64_t)rest_hint * 2, 256), 32), 2048);
        const AlnArrays aln{b.tid, b.pos, b.flag, b.xs, b.nh, b.cigar_off, b.n_cigar, rank_of_tid,
                            b.n_alignments, b.n_cigar_words, b.base_read_index};
        hipLaunchKernelGGL(k_scan_rest, dim3(rest_blocks), dim3(256), 0, s, p, aln, b.cigar, defer, items, pays, cnts, digs, capacity,
                           counters);
    }
    if (p.n_stripes)
        hipLaunchKernelGGL(k_close, dim3(CLOSE_BLOCKS), dim3(CLOSE_BT), 0, s, counters, p.base, capacity, items, pays, cnts, digs);
    ipx_prof_mark(s, IPXK_SCAN);
    return hipGetLastError();
}
