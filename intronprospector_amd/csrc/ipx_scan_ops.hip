// ipx_scan_ops.hip -- CIGAR scan, one lane per CIGAR OPERATION.
//
// Same function as ipx_scan.hip (alignments -> QC-passed junction records; reference
// src/junctions_extractor.cc:114-319, 384-401, 533-549), different mapping: a wavefront
// walks a contiguous span of the flat CIGAR array 64 words at a time (one coalesced
// 256-byte load) and every lane owns one operation.  The reference's sequential state
// machine is restated as prefix operations over the operations of an alignment:
//
//   tokens      A (M,=,X)  N  g (I/D inside the allowed run)  B (S, H, or an I/D whose run
//               since the last A/N exceeds max_anchor_indel_size)
//   state       LIMBO --A--> LEFT --N--> INTRON --A--> RIGHT --N--> INTRON ...; B -> LIMBO.
//               Inside a run without B the state is a function of three facts (was there an
//               A; which came last, A or N; was there an N after the first A), all read off
//               ballot masks with bit operations: no scan.
//   sums        target_pos, X lengths left/right, I/D lengths left/right, I/D run length:
//               six inclusive wave scans, made per-alignment by subtracting the value at the
//               alignment's first lane.
//   events      "value at the last lane where X happened" (start of the left anchor, end of
//               the intron, last shift of the anchor counters, ...) is one ballot, a
//               count-leading-zeros and ONE shuffle.
//   the quirk   the indel counters survive a LIMBO: lind/rind are the I/D
//               sums since the last *shift* (an N met in RIGHT), not since the anchor began.
//
// Values that outlive a 64-word chunk are carried in wave-uniform registers.  A junction
// is recorded by the lane of the operation that closes its right anchor (or by the last
// lane of the alignment).  Pass 1 counts, one atomicAdd per block reserves the record
// slots, pass 2 recomputes and stores -- as in ipx_scan.hip.
//
// Requires the batch's CIGAR words to be laid out back to back in alignment order
// (cigar_off[i+1] == cigar_off[i] + n_cigar[i]); a wave that sees anything else raises
// Counters::layout_flag and the caller falls back to the lane-per-alignment kernel.
#include "ipx_common.h"

namespace {

enum : uint32_t { S_LIMBO = 0, S_LEFT = 1, S_INTRON = 2, S_RIGHT = 3 };
enum : uint32_t { T_A = 0, T_N = 1, T_G = 2, T_B = 3 };
enum : uint32_t {
    F_PAIRED = 0x1, F_PROPER = 0x2, F_UNMAP = 0x4, F_REVERSE = 0x10, F_MREVERSE = 0x20, F_READ1 = 0x40,
    F_READ2 = 0x80, F_SECONDARY = 0x100, F_QCFAIL = 0x200, F_DUP = 0x400, F_SUPPLEMENTARY = 0x800
};

struct OpsArgs {
    ScanParams p;
    const int32_t *__restrict__ tid;
    const int32_t *__restrict__ pos;
    const uint16_t *__restrict__ flag;
    const uint8_t *__restrict__ xs;
    const uint8_t *__restrict__ nh;
    const uint32_t *__restrict__ off;
    const uint32_t *__restrict__ nc;
    const uint32_t *__restrict__ cigar;
    const uint32_t *__restrict__ rank_of_tid;
    uint64_t n, base_read_index, n_words;
    uint32_t reads_per_wave;
    uint4 *__restrict__ items;
    uint4 *__restrict__ pays;
    uint32_t *__restrict__ cnts;
    uint8_t *__restrict__ digs;
    uint64_t capacity;
    Counters *counters;
};

__device__ __forceinline__ uint64_t le_mask(uint32_t k) { return ~0ull >> (63 - k); }
__device__ __forceinline__ uint64_t lt_mask(uint32_t k) { return k ? (~0ull >> (64 - k)) : 0ull; }
__device__ __forceinline__ int top_bit(uint64_t m) { return m ? 63 - __clzll((long long)m) : -1; }
__device__ __forceinline__ int low_bit(uint64_t m) { return m ? __ffsll((unsigned long long)m) - 1 : -1; }

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, uint32_t lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t x = __shfl_up(v, o);
        if ((int)lane >= o) v += x;
    }
    return v;
}

// value carried by an alignment across chunk boundaries (all wave-uniform)
struct Carry {
    uint32_t live, dead, state, ord;
    uint32_t off, nc;                  // the alignment's words
    uint32_t tp, xl, xr, gl, gr, cur;  // sums after the last processed word
    uint32_t toLF_tp, toLF_xl, toRT_tp, toRT_xr, lastA_end, sh_gr, sh_gl, rs;
    uint32_t sn_is, sn_as, sn_lmm, sn_lind;
    uint64_t rid;
};

// state before lane k, from the tokens of its run [run_lo, k) and the state the run started in
__device__ __forceinline__ uint32_t state_from_masks(uint64_t mA, uint64_t mN, uint32_t s0) {
    // mA / mN: A and N tokens of the run, restricted to the lanes before k
    if (s0 == S_LIMBO) {
        if (!mA) return S_LIMBO;
        const int a0 = low_bit(mA);
        const uint64_t after_first = ~le_mask((uint32_t)a0);
        const uint64_t n_after = mN & after_first;
        const int la = top_bit(mA), ln = top_bit(n_after);
        if (ln > la) return S_INTRON;
        return (n_after & lt_mask((uint32_t)la)) ? S_RIGHT : S_LEFT;
    }
    const int la = top_bit(mA), ln = top_bit(mN);
    if (la < 0 && ln < 0) return s0;
    if (ln > la) return S_INTRON;
    // last significant token is an A
    if (s0 == S_INTRON || s0 == S_RIGHT) return S_RIGHT;
    return (mN & lt_mask((uint32_t)la)) ? S_RIGHT : S_LEFT;        // s0 == LEFT
}

template <bool EMIT>
__device__ __forceinline__ uint32_t wave_span(const OpsArgs &a, uint64_t ra, uint64_t rb, uint64_t slot_base,
                                              uint32_t *head_s /* 64 words of LDS for this wave */,
                                              unsigned long long &err_out, uint32_t (&kb)[6], bool &layout_bad) {
    const uint32_t lane = threadIdx.x & 63u;
    const ScanParams &p = a.p;
    const uint64_t o_begin = a.off[ra];
    const uint64_t o_end = (uint64_t)a.off[rb - 1] + a.nc[rb - 1];
    if (o_end < o_begin || o_end > a.n_words) { layout_bad = true; return 0; }

    Carry c{};
    c.rid = ra;               // "the alignment before the first head": nothing live
    c.live = 0;
    c.off = (uint32_t)o_begin; c.nc = 0;
    uint64_t next_rid = ra;   // next alignment whose head has not been placed yet
    uint64_t next_off = o_begin;                         // its offset (uniform), ~0 when there is none
    uint32_t emitted = 0;     // QC-passed records of this span so far
    uint32_t w_next = (o_begin + lane < o_end) ? a.cigar[o_begin + lane] : 0u;

    for (uint64_t c0 = o_begin; c0 < o_end; c0 += 64) {
        const uint64_t o = c0 + lane;
        const bool valid = o < o_end;
        const uint32_t w = w_next;
        // the next chunk's words are requested now and waited for a whole chunk later
        w_next = (o + 64 < o_end) ? a.cigar[o + 64] : 0u;
        // ---- which alignment does each word belong to: heads of the chunk -------------------
        const bool heads_here = next_off < c0 + 64;                      // uniform: a head may fall in this chunk
        if (heads_here) head_s[lane] = 0;
        __builtin_amdgcn_wave_barrier();
        while (next_rid < rb && next_off < c0 + 64) {
            // 64 candidate alignments at a time; the ones that start inside [c0, c0+64) form a prefix
            const uint64_t r = next_rid + lane;
            bool consumed = false;
            if (r < rb) {
                const uint32_t r_nc = a.nc[r];
                const uint64_t ro = a.off[r];
                if (r_nc == 0) {
                    consumed = true;                                   // no words: nothing to place
                } else if (ro >= c0 && ro < c0 + 64) {
                    consumed = true;
                    atomicMax(&head_s[(uint32_t)(ro - c0)], (uint32_t)(r - ra) + 1u);
                } else if (ro < c0) {
                    layout_bad = true;                                 // offsets must ascend with the words
                    consumed = true;
                }
            }
            const uint64_t not_consumed = ~ipx_ballot(consumed);
            const uint32_t n_in = not_consumed ? (uint32_t)(__ffsll((unsigned long long)not_consumed) - 1) : 64u;
            next_rid += n_in;
            next_off = next_rid < rb ? (uint64_t)a.off[next_rid] : ~0ull;
            if (n_in < 64) break;
        }
        __builtin_amdgcn_wave_barrier();
        const uint32_t hv = heads_here ? head_s[lane] : 0u;
        const bool is_head = valid && hv != 0;
        const uint64_t mH = ipx_ballot(is_head);
        const int h_k = top_bit(mH & le_mask(lane));                      // my alignment's head lane, -1: carried
        const uint32_t seg_lo = h_k >= 0 ? (uint32_t)h_k : 0u;
        const uint64_t seg = le_mask(lane) & ~lt_mask(seg_lo);            // lanes of my alignment up to me
        const uint64_t seg_x = seg & ~(1ull << lane);                     // ... before me
        const bool fresh = h_k >= 0;

        // per-alignment attributes, loaded by the head lanes
        uint32_t h_live = 0, h_off = 0, h_nc = 0;
        uint64_t h_rid = 0;
        unsigned long long my_err = IPX_ERRW_NONE;
        if (is_head) {
            h_rid = ra + hv - 1;
            const int32_t tid = a.tid[h_rid];
            const uint32_t flag = a.flag[h_rid];
            h_nc = a.nc[h_rid];
            h_off = a.off[h_rid];
            h_live = !(h_nc <= 1 || tid < 0 || (flag & (F_QCFAIL | F_DUP)));   // junctions_extractor.cc:536
            if (h_live && (uint32_t)tid >= p.n_targets) {
                my_err = ((a.base_read_index + h_rid) << IPX_ORD_SHIFT_READ) | (IPX_ERRK_TARGET << 8);
                h_live = 0;
            }
            if (h_off != (uint32_t)o) layout_bad = true;
        }
        const uint32_t src = fresh ? (uint32_t)h_k : 0u;
        const uint64_t rid = fresh ? __shfl(h_rid, src) : c.rid;
        const uint32_t r_off = fresh ? __shfl(h_off, src) : c.off;
        const uint32_t r_nc = fresh ? __shfl(h_nc, src) : c.nc;
        uint32_t live = fresh ? __shfl(h_live, src) : c.live;
        // words that belong to no alignment of this span (holes) are a layout error
        if (valid && !(o >= r_off && o < (uint64_t)r_off + r_nc)) { if (r_nc || fresh) layout_bad = true; live = 0; }
        const bool last_of_read = valid && (o + 1 == (uint64_t)r_off + r_nc);

        // ---- tokens ----------------------------------------------------------------------
        const uint32_t op = w & 0xfu, len = w >> 4;
        const bool op_a = (op == 0) | (op == 7) | (op == 8), op_n = (op == 3), op_id = (op == 1) | (op == 2),
                   op_clip = (op == 4) | (op == 5);
        const bool op_bad = valid && live && !(op_a | op_n | op_id | op_clip);
        // the exception ends the alignment: nothing after the first bad op counts (junctions_extractor.cc:312-313)
        const uint64_t m_bad = ipx_ballot(op_bad);
        const bool dead = (fresh ? 0u : c.dead) || (m_bad & seg_x) != 0;
        const bool act = valid && live && !dead && !op_bad;

        // run of insertions/deletions since the last A/N (S/H do not reset it, :153-158, :227, :252)
        const uint32_t g_len = (act && op_id) ? len : 0u;
        const uint32_t gI = wave_incl_scan(g_len, lane);
        const uint64_t m_reset = ipx_ballot(valid && (op_a | op_n)) | mH;          // A, N or a new alignment reset cur
        const int rs_k = top_bit(m_reset & seg);                                  // last reset at or before me
        // value BEFORE the resetting lane's own contribution (a head that is itself an I/D keeps its length)
        const uint32_t g_at_reset = __shfl(gI - g_len, rs_k >= 0 ? rs_k : 0);
        const uint32_t cur = (rs_k >= 0) ? gI - g_at_reset : c.cur + gI;          // no reset in sight: carried run
        const bool tok_b = act && (op_clip | (op_id && cur > p.max_anchor_indel_size));
        const bool tok_a = act && op_a, tok_n = act && op_n;
        const uint64_t mA = ipx_ballot(tok_a), mN = ipx_ballot(tok_n), mB = ipx_ballot(tok_b);

        // ---- state before every op ---------------------------------------------------------------
        const int b_k = top_bit(mB & seg_x);                                       // last break before me
        const uint32_t run_lo = b_k >= 0 ? (uint32_t)b_k + 1 : seg_lo;
        const uint64_t run = lt_mask(lane) & ~lt_mask(run_lo);
        const uint32_t s0 = (b_k >= 0 || fresh) ? (uint32_t)S_LIMBO : c.state;
        const uint32_t sb = state_from_masks(mA & run, mN & run, s0);
        uint32_t sa = sb;                                                          // state after my op
        if (tok_a) sa = (sb == S_LIMBO || sb == S_LEFT) ? (uint32_t)S_LEFT : (uint32_t)S_RIGHT;
        else if (tok_n) sa = (sb == S_LIMBO) ? (uint32_t)S_LIMBO : (uint32_t)S_INTRON;
        else if (tok_b) sa = S_LIMBO;

        // ---- sums, made per alignment -------------------------------------------------------------
        const uint32_t ref_len = (act && (op_a | op_n | (op == 2))) ? len : 0u;
        const uint32_t xl_v = (tok_a && op == 8 && sa == S_LEFT) ? len : 0u;       // count_mismatchs :167-173
        const uint32_t xr_v = (tok_a && op == 8 && sa == S_RIGHT) ? len : 0u;
        const uint32_t gl_v = (act && op_id && sb == S_LEFT) ? len : 0u;           // count_indels :175-181
        const uint32_t gr_v = (act && op_id && sb == S_RIGHT) ? len : 0u;
        auto seg_excl = [&](uint32_t v, uint32_t carry) -> uint32_t {
            const uint32_t e = wave_incl_scan(v, lane) - v;                        // exclusive over the chunk
            const uint32_t e0 = __shfl(e, seg_lo);
            return fresh ? e - e0 : carry + e;
        };
        const uint32_t tp = seg_excl(ref_len, c.tp);          // target_pos - pos before my op
        const uint32_t xl = seg_excl(xl_v, c.xl), xr = seg_excl(xr_v, c.xr);
        const uint32_t gl = seg_excl(gl_v, c.gl), gr = seg_excl(gr_v, c.gr);

        // ---- events: value at the last lane (of my alignment) where something happened ----------------
        // incl: lanes <= me, excl: lanes < me; falls back to the carried value, or to `none` for a fresh alignment
        auto last_at = [&](uint64_t m, bool incl, uint32_t v, uint32_t carry) -> uint32_t {
            const int j = top_bit(m & (incl ? seg : seg_x));
            const uint32_t got = __shfl(v, j >= 0 ? j : 0);
            return j >= 0 ? got : (fresh ? 0u : carry);
        };
        const uint64_t m_toLF = ipx_ballot(tok_a && sb == S_LIMBO);
        const uint64_t m_toRT = ipx_ballot(tok_a && sb == S_INTRON);
        const uint64_t m_shift = ipx_ballot(tok_n && sb == S_RIGHT);
        const uint64_t m_toIN = ipx_ballot(tok_n && (sb == S_LEFT || sb == S_RIGHT));
        const uint32_t toLF_tp = last_at(m_toLF, true, tp, c.toLF_tp), toLF_xl = last_at(m_toLF, true, xl, c.toLF_xl);
        const uint32_t toRT_tp = last_at(m_toRT, true, tp, c.toRT_tp), toRT_xr = last_at(m_toRT, true, xr, c.toRT_xr);
        const uint32_t lastA_end = last_at(mA, true, tp + len, c.lastA_end);
        // shift (N met in RIGHT): lind := rind, rind := 0 (:160-165); sums restart there
        const uint32_t sh_gr_x = last_at(m_shift, false, gr, c.sh_gr), sh_gl_x = last_at(m_shift, false, gl, c.sh_gl);
        const uint32_t rind_now = gr - sh_gr_x;                                      // rind before my op
        const uint32_t rs_x = last_at(m_shift, false, rind_now, c.rs);              // rind at the last shift = lind base
        const uint32_t lind_now = rs_x + (gl - sh_gl_x);
        const uint32_t rmm_now = xr - toRT_xr;
        // snapshot taken by the N that opens an intron: what the junction's left side is
        const bool from_rt = sb == S_RIGHT;
        const uint32_t my_is = tp;
        const uint32_t my_as = from_rt ? toRT_tp : toLF_tp;
        const uint32_t my_lmm = from_rt ? rmm_now : xl - toLF_xl;
        const uint32_t my_lind = from_rt ? rind_now : lind_now;
        const uint32_t sn_is = last_at(m_toIN, false, my_is, c.sn_is), sn_as = last_at(m_toIN, false, my_as, c.sn_as);
        const uint32_t sn_lmm = last_at(m_toIN, false, my_lmm, c.sn_lmm), sn_lind = last_at(m_toIN, false, my_lind, c.sn_lind);

        // ---- record_junction: closed right anchors -------------------------------------------------
        const bool e1 = act && sb == S_RIGHT && (tok_n | tok_b);        // closed by N / limbo (:145-158, :213-219)
        const bool e2 = act && last_of_read && sa == S_RIGHT;           // finish() (:278-282)
        bool rec = e1 | e2;
        uint32_t r_as = sn_as, r_is = sn_is, r_ie = toRT_tp, r_ae = lastA_end;
        // my own op still counts: an X that extends the right anchor, an I/D counted before the limbo test
        uint32_t r_lmm = sn_lmm, r_lind = sn_lind, r_rmm = rmm_now + xr_v, r_rind = rind_now + gr_v;
        // an alignment's attributes are needed only where a junction is recorded
        bool pass = false;
        uint32_t r_pos = 0, r_flag = 0, r_nhc = 0, r_rank = 0, r_strand = '.';
        if (rec) {
            r_flag = a.flag[rid];
            r_nhc = a.nh[rid];
            const bool rejected = (r_flag & F_SECONDARY) || (p.exclude_multi && r_nhc == 2);       // :388-391
            const int64_t ilen = (int64_t)(uint32_t)(r_ie - r_is);
            const int64_t left = (int64_t)(uint32_t)(r_is - r_as) - (int64_t)(uint32_t)(r_lmm + r_lind);
            const int64_t right = (int64_t)(uint32_t)(r_ae - r_ie) - (int64_t)(uint32_t)(r_rmm + r_rind);
            pass = !rejected && ilen >= (int64_t)p.min_intron_length && ilen <= (int64_t)p.max_intron_length &&
                   left >= (int64_t)p.min_anchor_length && right >= (int64_t)p.min_anchor_length;   // :392-397
        }
        const uint64_t m_pass = ipx_ballot(pass);
        const uint32_t ord = (fresh ? 0u : c.ord) + (uint32_t)__popcll(m_pass & seg_x);
        if (op_bad && !dead) {
            const unsigned long long e = ((a.base_read_index + rid) << IPX_ORD_SHIFT_READ) |
                                         ((unsigned long long)ord << IPX_ORD_SHIFT_ORDINAL) | (IPX_ERRK_CIGAR << 8) |
                                         (uint32_t)((op == 6) ? 'P' : (op == 9 ? 'B' : '?'));
            my_err = e < my_err ? e : my_err;
        }
        if (EMIT && pass) {
            const uint64_t idx = slot_base + emitted + (uint32_t)__popcll(m_pass & lt_mask(lane));
            if (idx < a.capacity) {
                r_pos = (uint32_t)a.pos[rid];
                const uint32_t xsb = a.xs[rid];
                r_rank = a.rank_of_tid[a.tid[rid]];
                // get_junction_strand :362-371 / :333-359
                if (p.strandness != IPX_UNSTRANDED) {
                    const uint32_t reversed = (r_flag & F_REVERSE) != 0, mrev = (r_flag & F_MREVERSE) != 0;
                    uint32_t fs = ((r_flag & F_READ1) != 0) ^ reversed, ss = ((r_flag & F_READ2) != 0) ^ mrev;
                    const uint32_t flip = (p.strandness - 1) ? 0u : 1u;
                    fs ^= flip; ss ^= flip;
                    r_strand = (fs == ss) ? (fs ? '+' : '-') : '.';
                } else {
                    r_strand = xsb ? xsb : (uint32_t)'.';
                }
                // get_read_category :71-95
                const uint32_t by_tag = (r_nhc == 0) ? 2u : (r_nhc == 1 ? 0u : 1u);
                uint32_t cat;
                if (r_flag & (F_QCFAIL | F_UNMAP | F_DUP)) cat = 3u;
                else if (r_flag & (F_SECONDARY | F_SUPPLEMENTARY)) cat = 1u;
                else if (r_flag & F_PAIRED) cat = (r_flag & F_PROPER) ? by_tag : 2u;
                else cat = by_tag;
                const uint32_t start = r_pos + r_is, jlen = r_ie - r_is, off16 = r_is & 0xffffu;
                const uint32_t s1 = 16 + p.key.len_bits, s2 = s1 + 32;
                uint64_t lo = (uint64_t)off16 | ((uint64_t)jlen << 16) | ((uint64_t)start << s1);
                uint64_t hi = (s1 > 32) ? ((uint64_t)start >> (64 - s1)) : 0;
                if (s2 < 64) {
                    lo |= (uint64_t)r_rank << s2;
                    if (s2 > 32) hi |= (uint64_t)r_rank >> (64 - s2);
                } else {
                    hi |= (uint64_t)r_rank << (s2 - 64);
                }
                const unsigned long long order = ((a.base_read_index + rid) << IPX_ORD_SHIFT_READ) |
                                                 ((unsigned long long)ord << IPX_ORD_SHIFT_ORDINAL) | (r_strand << 2) | cat;
                const uint4 it = make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)idx);
                a.items[idx] = it;
                a.pays[idx] = make_uint4((uint32_t)order, (uint32_t)(order >> 32), r_pos + r_as, r_pos + r_ae);
                a.cnts[idx] = IPX_CNT_ONE((uint32_t)order & 3u);
                a.digs[idx] = (uint8_t)ipx_digit_of(it, a.p.first);
                kb[0] |= (uint32_t)lo; kb[1] |= (uint32_t)(lo >> 32); kb[2] |= (uint32_t)hi;
                kb[3] &= (uint32_t)lo; kb[4] &= (uint32_t)(lo >> 32); kb[5] &= (uint32_t)hi;
            }
        }
        emitted += (uint32_t)__popcll(m_pass);
        // earliest error of the chunk
        {
            unsigned long long e = my_err;
            for (int d = 32; d >= 1; d >>= 1) {
                const unsigned long long x = __shfl_xor(e, d);
                e = x < e ? x : e;
            }
            err_out = e < err_out ? e : err_out;
        }

        // ---- carry: everything as of after the chunk's last valid word ---------------------------
        const uint64_t m_valid = ipx_ballot(valid);
        const int L = top_bit(m_valid);                    // last valid lane (>= 0)
#define RL(x) ((uint32_t)__builtin_amdgcn_readlane((int)(x), L))
        c.rid = ((uint64_t)RL((uint32_t)(rid >> 32)) << 32) | RL((uint32_t)rid);
        c.off = RL(r_off);
        c.nc = RL(r_nc);
        c.live = RL(live);
        c.dead = RL((uint32_t)(dead | op_bad));
        c.state = RL(sa);
        c.ord = RL(ord + (pass ? 1u : 0u));
        c.tp = RL(tp + ref_len);
        c.xl = RL(xl + xl_v); c.xr = RL(xr + xr_v);
        c.gl = RL(gl + gl_v); c.gr = RL(gr + gr_v);
        c.cur = RL((op_a | op_n) ? 0u : cur);       // A / N reset the run (:227, :252)
        c.toLF_tp = RL(toLF_tp); c.toLF_xl = RL(toLF_xl);
        c.toRT_tp = RL(toRT_tp); c.toRT_xr = RL(toRT_xr);
        c.lastA_end = RL(lastA_end);
        // inclusive versions of the exclusive event values: did lane L itself shift / open an intron?
        const bool L_shift = (m_shift >> L) & 1ull, L_toIN = (m_toIN >> L) & 1ull;
        c.sh_gr = L_shift ? RL(gr) : RL(sh_gr_x);
        c.sh_gl = L_shift ? RL(gl) : RL(sh_gl_x);
        c.rs = L_shift ? RL(rind_now) : RL(rs_x);
        c.sn_is = L_toIN ? RL(my_is) : RL(sn_is);
        c.sn_as = L_toIN ? RL(my_as) : RL(sn_as);
        c.sn_lmm = L_toIN ? RL(my_lmm) : RL(sn_lmm);
        c.sn_lind = L_toIN ? RL(my_lind) : RL(sn_lind);
    }
#undef RL
    return emitted;
}

__global__ __launch_bounds__(256) void k_scan_ops(OpsArgs a) {
    __shared__ uint32_t head_s[4][64];
    __shared__ uint32_t wcount[4];
    __shared__ uint32_t kb_s[4][6];
    __shared__ unsigned long long block_slot;
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const uint64_t waves_total = (a.n + a.reads_per_wave - 1) / a.reads_per_wave;
    const uint64_t n_blocks = (waves_total + 3) / 4;
    uint32_t kb[6] = {0u, 0u, 0u, ~0u, ~0u, ~0u};
    bool layout_bad = false;
    unsigned long long err = IPX_ERRW_NONE, dummy = IPX_ERRW_NONE;

    for (uint64_t blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
        const uint64_t wv = blk * 4 + wave;
        const uint64_t ra = wv * a.reads_per_wave;
        uint64_t rb = ra + a.reads_per_wave;
        if (rb > a.n) rb = a.n;
        uint32_t cnt = 0;
        if (ra < a.n) cnt = wave_span<false>(a, ra, rb, 0, head_s[wave], err, kb, layout_bad);
        if (lane == 0) wcount[wave] = cnt;
        __syncthreads();
        uint32_t before = 0, total = 0;
        for (uint32_t w = 0; w < 4; ++w) {
            if (w < wave) before += wcount[w];
            total += wcount[w];
        }
        if (t == 0) block_slot = total ? atomicAdd(&a.counters->n_records, (unsigned long long)total) : 0ull;
        __syncthreads();
        if (ra < a.n && cnt) (void)wave_span<true>(a, ra, rb, block_slot + before, head_s[wave], dummy, kb, layout_bad);
        __syncthreads();
    }
    if (lane == 0 && err != IPX_ERRW_NONE) atomicMin(&a.counters->err_word, err);
    if (ipx_ballot(layout_bad) && lane == 0) atomicOr(&a.counters->pad[0], 1u);
    // OR / AND of the stored keys
    uint32_t v[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        v[k] = kb[k];
        for (int d = 32; d >= 1; d >>= 1) {
            const uint32_t o = __shfl_xor(v[k], d);
            v[k] = (k < 3) ? (v[k] | o) : (v[k] & o);
        }
        if (lane == 0) kb_s[wave][k] = v[k];
    }
    __syncthreads();
    if (t < 6) {
        const uint32_t x0 = kb_s[0][t], x1 = kb_s[1][t], x2 = kb_s[2][t], x3 = kb_s[3][t];
        if (t < 3) {
            const uint32_t o = x0 | x1 | x2 | x3;
            if (o) atomicOr(&a.counters->key_or[t], o);
        } else {
            const uint32_t an = x0 & x1 & x2 & x3;
            if (an != ~0u) atomicAnd(&a.counters->key_and[t - 3], an);
        }
    }
}

}  // namespace

hipError_t ipx_launch_scan_ops(hipStream_t s, const ScanParams &p, const ipx_batch &b, const uint32_t *rank_of_tid,
                               uint4 *items, uint4 *pays, uint32_t *cnts, uint8_t *digs, uint64_t capacity, Counters *counters) {
    if (b.n_alignments == 0) return hipSuccess;
    OpsArgs a{};
    a.p = p;
    a.tid = b.tid; a.pos = b.pos; a.flag = b.flag; a.xs = b.xs; a.nh = b.nh; a.off = b.cigar_off; a.nc = b.n_cigar;
    a.cigar = b.cigar; a.rank_of_tid = rank_of_tid;
    a.n = b.n_alignments; a.base_read_index = b.base_read_index; a.n_words = b.n_cigar_words;
    // ~4096 words per wave
    const double avg = (double)b.n_cigar_words / (double)b.n_alignments;
    uint64_t rpw = (uint64_t)(4096.0 / (avg < 1.0 ? 1.0 : avg));
    if (rpw < 1) rpw = 1;
    if (rpw > 2048) rpw = 2048;
    a.reads_per_wave = (uint32_t)rpw;
    a.items = items; a.pays = pays; a.cnts = cnts; a.digs = digs; a.capacity = capacity; a.counters = counters;
    const uint64_t waves = ipx_div_up(b.n_alignments, rpw);
    uint64_t blocks = ipx_div_up(waves, 4);
    if (blocks > 256ull * 32) blocks = 256ull * 32;
    hipLaunchKernelGGL(k_scan_ops, dim3((uint32_t)blocks), dim3(256), 0, s, a);
    return hipGetLastError();
}

// This file is NOT part of libipx.so (round 4): the per-operation scan is an alternative formulation that measured 3-4
// times slower than the lane-per-alignment kernels; it is built into libipx_scanops.so, which
// libipx.so loads at ipx_create when IPX_SCAN=ops asks for it (tests/test_gpu_parity.py keeps its parity test).
extern "C" hipError_t ipx_scanops_launch(hipStream_t s, const ScanParams *p, const ipx_batch *b, const uint32_t *rank_of_tid,
                                         uint4 *items, uint4 *pays, uint32_t *cnts, uint8_t *digs, uint64_t capacity,
                                         Counters *counters) {
    return ipx_launch_scan_ops(s, *p, *b, rank_of_tid, items, pays, cnts, digs, capacity, counters);
}
