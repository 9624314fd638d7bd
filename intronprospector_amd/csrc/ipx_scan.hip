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
// Mapping: one lane per alignment, 64 consecutive alignments per wavefront, so
// that a wave's CIGAR words form one contiguous span of the flat array.  The
// op loop is wave-uniform (runs to the longest CIGAR of the wave) because record
// slots are handed out with one atomic per wave and step (ballot + popcount).
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

__global__ __launch_bounds__(256) void k_scan(ScanParams p, uint64_t n, uint64_t base_read_index,
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
                                              uint64_t capacity, Counters *counters) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t lane_lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const uint64_t wave0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_chunks = (n + 63) >> 6;

    for (uint64_t chunk = wave0; chunk < n_chunks; chunk += n_waves) {
        const uint64_t i = (chunk << 6) + lane;
        uint32_t nc = 0, off = 0, flag = 0, nhc = 0, rank = 0, strand = '.';
        int32_t tid = -1;
        int64_t pos = 0;
        unsigned long long my_err = IPX_ERRW_NONE;
        const uint64_t read_index = base_read_index + i;
        if (i < n) {
            tid = a_tid[i];
            flag = a_flag[i];
            nc = a_nc[i];
            // junctions_extractor.cc:536
            if (nc <= 1 || tid < 0 || (flag & (F_QCFAIL | F_DUP))) nc = 0;
            if (nc) {
                if ((uint32_t)tid >= p.n_targets) {
                    my_err = (read_index << IPX_ORD_SHIFT_READ) | (IPX_ERRK_TARGET << 8);
                    nc = 0;
                } else {
                    pos = a_pos[i];
                    off = a_off[i];
                    nhc = a_nh[i];
                    rank = rank_of_tid[tid];
                    uint32_t xs = a_xs[i];
                    strand = (p.strandness != IPX_UNSTRANDED) ? strand_from_flag(flag, p.strandness)
                                                              : (xs ? xs : (uint32_t)'.');
                }
            }
        }
        const uint32_t cat = read_category(flag, nhc);
        // junction_qc's per-read part, junctions_extractor.cc:388-391
        const bool read_rejected = (flag & F_SECONDARY) || (p.exclude_multi && nhc == 2);

        uint32_t state = ST_LIMBO;
        int64_t target_pos = pos, anchor_start = -1, intron_start = -1, intron_end = -1, anchor_end = -1;
        uint32_t lmm = 0, rmm = 0, lind = 0, rind = 0, cur_indel = 0, ordinal = 0;

        // longest CIGAR in the wave
        uint32_t max_nc = nc;
        for (int d = 32; d >= 1; d >>= 1) {
            uint32_t o = __shfl_xor(max_nc, d);
            max_nc = o > max_nc ? o : max_nc;
        }

        for (uint32_t k = 0; k <= max_nc; ++k) {
            bool emit = false;
            Rec r;
            if (k < nc) {
                const uint32_t w = cigar[(uint64_t)off + k];
                const uint32_t op = w & 0xfu, len = w >> 4;
                // BAM_CIGAR_STR "MIDNSHP=XB": M0 I1 D2 N3 S4 H5 P6 =7 X8 B9
                if (op == 3) {                                  // N   :208-229
                    if (state == ST_RIGHT) {
                        emit = true;
                        r = {anchor_start, intron_start, intron_end, anchor_end, lmm, rmm, lind, rind};
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
                        if (state == ST_RIGHT) {
                            emit = true;
                            r = {anchor_start, intron_start, intron_end, anchor_end, lmm, rmm, lind, rind};
                        }
                        state = ST_LIMBO;
                    }
                    if (op == 2) target_pos += len;
                } else if (op == 4 || op == 5) {                // S, H  :306-311 -> enter_limbo
                    if (state == ST_RIGHT) {
                        emit = true;
                        r = {anchor_start, intron_start, intron_end, anchor_end, lmm, rmm, lind, rind};
                    }
                    state = ST_LIMBO;
                } else {                                        // :312-313
                    const uint32_t ch = (op == 6) ? 'P' : (op == 9 ? 'B' : '?');
                    my_err = (read_index << IPX_ORD_SHIFT_READ) |
                             ((unsigned long long)ordinal << IPX_ORD_SHIFT_ORDINAL) |
                             (IPX_ERRK_CIGAR << 8) | ch;
                    nc = 0;                                     // the exception ends this read
                    state = ST_LIMBO;
                }
            } else if (k == nc && nc != 0) {                    // finish()  :278-282
                if (state == ST_RIGHT) {
                    emit = true;
                    r = {anchor_start, intron_start, intron_end, anchor_end, lmm, rmm, lind, rind};
                }
                state = ST_LIMBO;
            }

            // junction_qc, junctions_extractor.cc:392-397 (int64 arithmetic, uint32 sums wrap)
            if (emit) {
                const int64_t ilen = r.intron_end - r.intron_start;
                const int64_t left = (r.intron_start - r.anchor_start) - (int64_t)(uint32_t)(r.lmm + r.lind);
                const int64_t right = (r.anchor_end - r.intron_end) - (int64_t)(uint32_t)(r.rmm + r.rind);
                emit = !read_rejected && ilen >= (int64_t)p.min_intron_length &&
                       ilen <= (int64_t)p.max_intron_length && left >= (int64_t)p.min_anchor_length &&
                       right >= (int64_t)p.min_anchor_length;
            }

            const uint64_t m = __ballot(emit);
            if (m) {
                const int leader = __ffsll((unsigned long long)m) - 1;
                unsigned long long base = 0;
                if ((int)lane == leader)
                    base = atomicAdd(&counters->n_records, (unsigned long long)__popcll(m));
                base = __shfl(base, leader);
                if (emit) {
                    const uint64_t idx = base + __popcll(m & lane_lt);
                    if (idx < capacity) {
                        const uint32_t start = (uint32_t)r.intron_start;
                        const uint32_t len = (uint32_t)(r.intron_end - r.intron_start);
                        const uint32_t off16 = (uint32_t)(r.intron_start - pos) & 0xffffu;
                        const uint32_t s1 = 16 + p.key.len_bits;        // <= 48
                        const uint32_t s2 = s1 + 32;                    // <= 80
                        uint64_t lo = (uint64_t)off16 | ((uint64_t)len << 16) | ((uint64_t)start << s1);
                        uint64_t hi = (s1 > 32) ? ((uint64_t)start >> (64 - s1)) : 0;
                        if (s2 < 64) {
                            lo |= (uint64_t)rank << s2;
                            if (s2 > 32) hi |= (uint64_t)rank >> (64 - s2);
                        } else {
                            hi |= (uint64_t)rank << (s2 - 64);
                        }
                        const unsigned long long order =
                            (read_index << IPX_ORD_SHIFT_READ) |
                            ((unsigned long long)ordinal << IPX_ORD_SHIFT_ORDINAL) | (strand << 2) | cat;
                        items[idx] = make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)idx);
                        pays[idx] = make_uint4((uint32_t)order, (uint32_t)(order >> 32),
                                               (uint32_t)r.anchor_start, (uint32_t)r.anchor_end);
                    }
                    ordinal++;
                }
            }
        }

        // earliest error of the wave -> one atomic
        unsigned long long e = my_err;
        for (int d = 32; d >= 1; d >>= 1) {
            unsigned long long o = __shfl_xor(e, d);
            e = o < e ? o : e;
        }
        if (lane == 0 && e != IPX_ERRW_NONE) atomicMin(&counters->err_word, e);
    }
}

}  // namespace

hipError_t ipx_launch_scan(hipStream_t s, const ScanParams &p, const ipx_batch &b,
                           const uint32_t *rank_of_tid, uint4 *items, uint4 *pays,
                           uint64_t capacity, Counters *counters) {
    if (b.n_alignments == 0) return hipSuccess;
    const uint64_t chunks = ipx_div_up(b.n_alignments, 64);
    uint64_t blocks = ipx_div_up(chunks, 4);
    if (blocks > 256ull * 64) blocks = 256ull * 64;     // grid-stride beyond 64 blocks per CU
    hipLaunchKernelGGL(k_scan, dim3((uint32_t)blocks), dim3(256), 0, s, p, b.n_alignments,
                       b.base_read_index, b.tid, b.pos, b.flag, b.xs, b.nh, b.cigar_off, b.n_cigar,
                       b.cigar, rank_of_tid, items, pays, capacity, counters);
    return hipGetLastError();
}
