// ipx_bam.hip -- BAM alignment records -> the SoA batch of include/ipx.h, on the device.
//
// Stands where htslib's sam_read1 and the reference's field reads stand in front of the hot path
// (src/junctions_extractor.cc:533-549 process_alignment; :49-56 get_num_aligns = bam_aux_get("NH") + bam_aux2i;
// :322-330 get_junction_strand_XS = bam_aux_get("XS") + bam_aux2A), for callers that hold the records as the
// bytes of the BAM stream (SAM spec 4.2: block_size, refID, pos, l_read_name, mapq, bin, n_cigar_op, flag, l_seq,
// next_refID, next_pos, tlen, read_name, cigar, seq, qual, tags).  The host only inflates the BGZF blocks and cuts
// them into SPANS, runs of whole records; everything per record happens here:
//   k_bam_chain   one lane per span: follow the block_size words, note where every record starts
//   k_bam_fields  one lane per record: the fixed fields, the tag walk for XS / NH / CG, bounds checks
//   (exclusive scan of the CIGAR lengths = cigar_off)
//   k_bam_cigar   the CIGAR words (the CG tag's for a long CIGAR, SAM spec 4.2.2) into the batch's array
// A record that does not parse (the cases host/aln_reader.cc: parse_bam_record throws for) is reported as
// 1 + its index in *err (smallest wins); nothing it says about lengths is trusted before it is checked.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ipx_common.h"

namespace {

// four bytes at any byte offset of a 4-byte aligned buffer (the buffer is padded by 8 bytes)
__device__ __forceinline__ uint32_t ld32(const uint8_t *__restrict__ bytes, uint64_t off) {
    const uint32_t *w = reinterpret_cast<const uint32_t *>(bytes + (off & ~3ull));
    return __builtin_amdgcn_alignbyte(w[1], w[0], (uint32_t)off & 3u);
}

__device__ __forceinline__ void bam_error(uint32_t *err, uint64_t record) {
    atomicMin(err, (uint32_t)(record + 1 < 0xffffffffull ? record + 1 : 0xfffffffeull));
}

__global__ __launch_bounds__(64) void k_bam_chain(const uint8_t *__restrict__ bytes, const uint64_t *__restrict__ span_begin,
                                                  const uint64_t *__restrict__ span_first, uint32_t n_spans,
                                                  uint64_t *__restrict__ rec_off, uint32_t *__restrict__ err) {
    const uint32_t s = blockIdx.x * 64 + threadIdx.x;
    if (s >= n_spans) return;
    uint64_t off = span_begin[s];
    const uint64_t end = span_begin[s + 1];
    uint64_t k = span_first[s];
    const uint64_t k_end = span_first[s + 1];
    while (k < k_end) {
        if (off + 4 > end) break;
        const uint32_t bs = ld32(bytes, off);
        if (bs < 32 || off + 4 + (uint64_t)bs > end) break;
        rec_off[k++] = off;
        off += 4 + (uint64_t)bs;
    }
    if (k != k_end || off != end) {
        bam_error(err, k < k_end ? k : (k_end ? k_end - 1 : 0));
        for (; k < k_end; ++k) rec_off[k] = ~0ull;               // k_bam_fields leaves these alone
    }
}

__global__ __launch_bounds__(256) void k_bam_fields(const uint8_t *__restrict__ bytes, const uint64_t *__restrict__ rec_off,
                                                    uint64_t n, const int32_t *__restrict__ tid_map, uint32_t n_tid_map,
                                                    int32_t *__restrict__ o_tid, int32_t *__restrict__ o_pos,
                                                    uint16_t *__restrict__ o_flag, uint8_t *__restrict__ o_xs,
                                                    uint8_t *__restrict__ o_nh, uint32_t *__restrict__ o_nc,
                                                    uint64_t *__restrict__ cig_src, uint32_t *__restrict__ err) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint64_t off = rec_off[i];
    int32_t tid = -1, pos = 0;
    uint32_t flag = 0, xs = 0, nh = 0, n_cig = 0;
    uint64_t cig = 0;
    bool ok = off != ~0ull;
    if (ok) {
        const uint64_t block = ld32(bytes, off);                 // >= 32 and inside its span (k_bam_chain)
        const uint64_t b = off + 4;                              // the record's body
        tid = (int32_t)ld32(bytes, b);
        pos = (int32_t)ld32(bytes, b + 4);
        const uint64_t l_read_name = ld32(bytes, b + 8) & 0xffu;
        const uint32_t w12 = ld32(bytes, b + 12);
        const uint64_t n_cigar = w12 & 0xffffu;
        flag = w12 >> 16;
        const uint64_t l_seq = ld32(bytes, b + 16);
        uint64_t o = 32 + l_read_name;
        ok = o <= block;
        cig = b + o;
        n_cig = (uint32_t)n_cigar;
        o += 4 * n_cigar;
        ok = ok && o <= block;
        o += (l_seq + 1) / 2 + l_seq;
        ok = ok && o <= block;
        // tags (bam_aux_get walks them the same way; the first XS / NH counts)
        bool have_xs = false, have_nh = false;
        uint64_t cg = 0;
        uint32_t cg_n = 0;
        bool have_cg = false;
        while (ok && o + 3 <= block) {
            const uint32_t t0 = bytes[b + o], t1 = bytes[b + o + 1], type = bytes[b + o + 2];
            const uint64_t p = o + 3;
            uint64_t sz = 0;
            long long ival = 0;
            bool is_int = false;
            // (a value that would reach past the record is caught by the check below before it is used)
            switch (type) {
                case 'A': sz = 1; break;
                case 'c': sz = 1; ival = (int8_t)bytes[b + p]; is_int = true; break;
                case 'C': sz = 1; ival = bytes[b + p]; is_int = true; break;
                case 's': sz = 2; ival = (int16_t)(ld32(bytes, b + p) & 0xffffu); is_int = true; break;
                case 'S': sz = 2; ival = (uint16_t)(ld32(bytes, b + p) & 0xffffu); is_int = true; break;
                case 'i': sz = 4; ival = (int32_t)ld32(bytes, b + p); is_int = true; break;
                case 'I': sz = 4; ival = ld32(bytes, b + p); is_int = true; break;
                case 'f': sz = 4; break;
                case 'd': sz = 8; break;
                case 'Z': case 'H': {
                    uint64_t q = p;
                    while (q < block && bytes[b + q]) ++q;
                    sz = q - p + 1;
                    break;
                }
                case 'B': {
                    if (p + 5 > block) { ok = false; break; }
                    const uint32_t sub = bytes[b + p];
                    const uint32_t cnt = ld32(bytes, b + p + 1);
                    const uint64_t es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
                    sz = 5 + es * (uint64_t)cnt;
                    if (t0 == 'C' && t1 == 'G' && sub == 'I' && !have_cg) { cg = b + p + 5; cg_n = cnt; have_cg = true; }   // the first CG tag, as bam_aux_get
                    break;
                }
                default: ok = false; break;
            }
            if (!ok) break;
            if (p + sz > block) { ok = false; break; }
            if (t0 == 'X' && t1 == 'S' && !have_xs) {
                have_xs = true;
                const uint32_t c = (type == 'A') ? (uint32_t)bytes[b + p] : 0u;     // bam_aux2A: 0 unless type 'A'
                xs = c ? c : (uint32_t)'.';
            } else if (t0 == 'N' && t1 == 'H' && !have_nh) {
                have_nh = true;
                const int v = (int)ival;                                            // bam_aux2i, cast as :51 does
                nh = is_int ? (v <= 0 ? 0u : (v == 1 ? 1u : 2u)) : 0u;
            }
            o = p + sz;
        }
        // long CIGARs live in the CG tag behind a "<l_seq>S<ref_len>N" placeholder (SAM spec 4.2.2)
        if (ok && have_cg && n_cigar == 2) {
            const uint32_t c0 = ld32(bytes, cig), c1 = ld32(bytes, cig + 4);
            if ((c0 & 0xfu) == 4u && (uint64_t)(c0 >> 4) == l_seq && (c1 & 0xfu) == 3u) { cig = cg; n_cig = cg_n; }
        }
        if (!ok) bam_error(err, i);
    }
    if (!ok) { tid = -1; n_cig = 0; }
    if (tid >= 0 && (uint32_t)tid < n_tid_map) tid = tid_map[tid];
    o_tid[i] = tid;
    o_pos[i] = pos;
    o_flag[i] = (uint16_t)flag;
    o_xs[i] = (uint8_t)xs;
    o_nh[i] = (uint8_t)nh;
    o_nc[i] = n_cig;
    cig_src[i] = cig;
}

// one lane per record for short CIGARs; a record of more than 32 words is copied by its whole wave
__global__ __launch_bounds__(256) void k_bam_cigar(const uint8_t *__restrict__ bytes, const uint64_t *__restrict__ cig_src,
                                                   const uint32_t *__restrict__ nc, const uint32_t *__restrict__ cigar_off,
                                                   uint64_t n, const uint32_t *__restrict__ total, uint64_t n_words,
                                                   uint32_t *__restrict__ cigar, uint32_t *__restrict__ err) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    if (i == 0 && (uint64_t)*total != n_words) bam_error(err, 0);           // the caller's count and the records disagree
    const bool mine = i < n;
    const uint32_t cnt = mine ? nc[i] : 0u;
    const uint32_t dst = mine ? cigar_off[i] : 0u;
    const uint64_t src = mine ? cig_src[i] : 0ull;
    const bool fits = (uint64_t)dst + cnt <= n_words;                       // (never false when the counts agree)
    if (fits && cnt <= 32u)
        for (uint32_t k = 0; k < cnt; ++k) cigar[dst + k] = ld32(bytes, src + 4ull * k);
    unsigned long long big = ipx_ballot(fits && cnt > 32u);
    while (big) {
        const int l = __builtin_ctzll(big);
        big &= big - 1;
        const uint32_t c = __shfl(cnt, l), d = __shfl(dst, l);
        const uint64_t s = ((uint64_t)__shfl((uint32_t)(src >> 32), l) << 32) | __shfl((uint32_t)src, l);
        for (uint32_t k = lane; k < c; k += 64) cigar[d + k] = ld32(bytes, s + 4ull * k);
    }
}

__global__ void k_bam_begin(uint32_t *err) { *err = 0xffffffffu; }

}  // namespace

size_t ipx_bam_workspace_bytes(uint64_t n) {
    // rec_off, cig_src (8 bytes per record), the scan's block sums and its total
    return (size_t)(16 * n + 4 * (ipx_div_up(n, 4096) + 1) + 256 + 64);
}

// stage 1: record starts, fixed fields, tags, CIGAR lengths and their prefix (cigar_off); *total (device) = CIGAR words
hipError_t ipx_launch_bam_decode_fields(hipStream_t s, const uint8_t *bytes, const uint64_t *span_begin, const uint64_t *span_first,
                                        uint32_t n_spans, uint64_t n, const int32_t *tid_map, uint32_t n_tid_map,
                                        int32_t *tid, int32_t *pos, uint16_t *flag, uint8_t *xs, uint8_t *nh, uint32_t *cigar_off,
                                        uint32_t *n_cigar, void *workspace, uint32_t *err, uint32_t **total_out) {
    char *ws = static_cast<char *>(workspace);
    uint64_t *rec_off = reinterpret_cast<uint64_t *>(ws);
    uint64_t *cig_src = rec_off + n;
    uint32_t *sums = reinterpret_cast<uint32_t *>(cig_src + n);
    uint32_t *total = sums + ipx_div_up(n, 4096) + 1 + 32;
    hipLaunchKernelGGL(k_bam_begin, dim3(1), dim3(1), 0, s, err);
    hipLaunchKernelGGL(k_bam_chain, dim3((uint32_t)ipx_div_up(n_spans, 64)), dim3(64), 0, s, bytes, span_begin, span_first,
                       n_spans, rec_off, err);
    hipLaunchKernelGGL(k_bam_fields, dim3((uint32_t)ipx_div_up(n, 256)), dim3(256), 0, s, bytes, rec_off, n, tid_map, n_tid_map,
                       tid, pos, flag, xs, nh, n_cigar, cig_src, err);
    hipError_t e = ipx_launch_scan_u32(s, n_cigar, cigar_off, n, sums, total);
    if (e != hipSuccess) return e;
    if (total_out) *total_out = total;
    return hipGetLastError();
}

// stage 2: the CIGAR words; n_words = what *total must say (the caller's count, or what it read back from stage 1)
hipError_t ipx_launch_bam_decode_cigar(hipStream_t s, const uint8_t *bytes, uint64_t n, uint64_t n_words, const uint32_t *cigar_off,
                                       const uint32_t *n_cigar, uint32_t *cigar, void *workspace, uint32_t *err) {
    char *ws = static_cast<char *>(workspace);
    uint64_t *rec_off = reinterpret_cast<uint64_t *>(ws);
    uint64_t *cig_src = rec_off + n;
    uint32_t *sums = reinterpret_cast<uint32_t *>(cig_src + n);
    uint32_t *total = sums + ipx_div_up(n, 4096) + 1 + 32;
    hipLaunchKernelGGL(k_bam_cigar, dim3((uint32_t)ipx_div_up(n, 256)), dim3(256), 0, s, bytes, cig_src, n_cigar, cigar_off, n,
                       total, n_words, cigar, err);
    return hipGetLastError();
}

hipError_t ipx_launch_bam_decode(hipStream_t s, const uint8_t *bytes, const uint64_t *span_begin, const uint64_t *span_first,
                                 uint32_t n_spans, uint64_t n, uint64_t n_words, const int32_t *tid_map, uint32_t n_tid_map,
                                 int32_t *tid, int32_t *pos, uint16_t *flag, uint8_t *xs, uint8_t *nh, uint32_t *cigar_off,
                                 uint32_t *n_cigar, uint32_t *cigar, void *workspace, uint32_t *err) {
    hipError_t e = ipx_launch_bam_decode_fields(s, bytes, span_begin, span_first, n_spans, n, tid_map, n_tid_map, tid, pos, flag, xs,
                                                nh, cigar_off, n_cigar, workspace, err, nullptr);
    if (e != hipSuccess) return e;
    return ipx_launch_bam_decode_cigar(s, bytes, n, n_words, cigar_off, n_cigar, cigar, workspace, err);
}
