// ipx_common.h -- shared device-side types of libipx (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/ipx.h"

#define IPX_WAVE 64

// ---- junction record ------------------------------------------------------
// One QC-passed junction record is 32 bytes in HBM, split in two 16-byte
// halves so that the sort moves only the first:
//
//   item  (uint4)  k0,k1,k2 = 96-bit sort key, little end first; w = record index (31 bits) | folded << 31
//   pay   (uint4)  x,y = order64 ; z = anchor_start ; w = anchor_end
//
// key bits, low to high:  off16:16 | len:LB | start:SB | rank:RB     (<= 96)
//   off16 = uint16(intron_start - pos)            junctions.hh:163-167
//   len   = intron_end - intron_start  (<= max_intron_length after junction_qc)
//   start = uint32(intron_start)                  junctions.hh:57-67
//   rank  = tid (target scope) or rank of the chromosome name (global scope)
// key >> 16 is the JunctionKey, ordered as junctions.hh:77-100.
//
// order64 = read_index:34 << 30 | ordinal:20 << 10 | strand:8 << 2 | cat:2
//   read_index = position of the alignment in the run, ordinal = number of the
//   record inside the alignment: (read_index, ordinal) is the order in which the
//   reference's sequential loop would have called record_junction().
//   strand = the READ's junction strand (junctions_extractor.cc:362-371)
//   cat    = ReadCategory (junctions.hh:49-53) or 3 = get_read_category would
//            throw (junctions_extractor.cc:73-74)
//
// Next to the two halves every record has a 4-byte count word (cnts[], indexed like pays[]):
//   c0 | c1 << 10 | c2 << 20 = how many reads of category 0/1/2 the record stands for.
// A record straight from a CIGAR has one bit set (none when cat == 3).  The short-read scan FOLDS
// records that share the whole 96-bit key (same junction, same read offset) before they leave the
// kernel -- a coordinate-sorted pile-up on a hot junction hands the sort one record per (junction,
// offset) and wave tile instead of hundreds: order64 = the smallest of the group (with that record's
// strand), anchors = min / max, counts = sums.  See ipx_scan_short.inc.
// item.w's top bit says whether the count word is worth fetching: clear = the record stands for ONE read, its count
// word is IPX_CNT_ONE(cat) and cat is in the payload the reader fetches anyway.  In input that is not sorted almost
// nothing folds, and the reduce's second random line per record (cnts[]) was a third of its time (0.45 -> 0.30 ms for
// 9.85 M shuffled records) because a handful of records had folded somewhere in the table.
#define IPX_ITEM_FOLDED 0x80000000u
#define IPX_ITEM_INDEX(w) ((w) & 0x7fffffffu)
#define IPX_MAX_RECORDS 0x7fffffffull          // records per table: the index has 31 bits
#define IPX_CNT_BITS 10
#define IPX_CNT_MASK 0x3ffu
#define IPX_CNT_ONE(cat) ((cat) < 3u ? (1u << (IPX_CNT_BITS * (cat))) : 0u)
#define IPX_ORD_SHIFT_READ 30
#define IPX_ORD_SHIFT_ORDINAL 10
#define IPX_ORD_MAX_ORDINAL ((1u << 20) - 1)
#define IPX_ORD_MAX_READ ((1ull << 34) - 1)
#define IPX_ORD_SEQ(o) ((o) >> IPX_ORD_SHIFT_ORDINAL)      // (read_index, ordinal)
#define IPX_ORD_READ(o) ((o) >> IPX_ORD_SHIFT_READ)
#define IPX_ORD_STRAND(o) ((uint32_t)((o) >> 2) & 0xffu)
#define IPX_ORD_CAT(o) ((uint32_t)(o) & 3u)

// error word, same ordering as order64 so that min() is "earliest in input
// order":  read_index:34 << 30 | ordinal:20 << 10 | kind:2 << 8 | detail:8
// kind 0 sorts before a record's own errors at the same ordinal: a CIGAR error
// met after m records were emitted carries ordinal m.
#define IPX_ERRW_NONE 0xffffffffffffffffull
#define IPX_ERRK_CIGAR 0u
#define IPX_ERRK_GENOME 1u   // detail: 0 = missing sequence, 1 = short fetch
#define IPX_ERRK_CATEGORY 2u
#define IPX_ERRK_TARGET 3u

struct KeyLayout {
    uint32_t len_bits, start_bits, rank_bits;   // off16 is always 16
    uint32_t total_bits, n_digits;              // 8-bit digits
};

// One radix pass looks at up to 8 key bits taken from at most two runs of bits:
// digit = bits [lo1, lo1+w1) | bits [lo2, lo2+w2) << w1   (w1 + w2 <= 8, lo2 >= lo1 + w1).
struct DigitSpec {
    uint32_t lo1, w1, lo2, w2;
};
// what a key rank's genome sequence is, in one 32-byte load
struct RankSeq {
    long long len;                    // sequence length; -1: the genome has no sequence of that name
    unsigned long long word2, wordm;  // first twobit / mask word of the sequence
    unsigned long long pad;
};
struct GenomeDev {
    const uint32_t *twobit;      // 16 bases per word, all sequences concatenated (word aligned per sequence)
    const uint32_t *mask;        // 32 bases per word
    const int64_t *seq_len;      // per genome sequence
    const uint64_t *seq_word2;   // first twobit word of the sequence
    const uint64_t *seq_wordm;   // first mask word of the sequence
    const int32_t *seq_of_rank;  // per key rank: genome sequence index, -1 = missing, -2 = no genome
    const RankSeq *rank_seq;     // the same per key rank, resolved (one dependent load less per look-up)
};
struct ScanParams {
    uint32_t min_anchor_length, max_anchor_indel_size, min_intron_length, max_intron_length;
    int32_t strandness;
    uint32_t exclude_multi;
    uint32_t n_targets;
    KeyLayout key;
    DigitSpec first;      // window of the digit bytes written next to the records (the first sort pass)
    // Where a launch's records go.  n_stripes == 0: appended at Counters::n_records.  n_stripes == IPX_STRIPES: the
    // launch fills IPX_STRIPES interleaved stripes after the `base` records the handle already holds, one
    // counter per stripe (a wave uses a stripe of the XCD it runs on): one returning atomic per reservation
    // on a single word saturates at ~60-90 M/s, thirty-two words on thirty-two lines do not.  ipx_launch_scan then
    // closes the gaps the stripes leave, so everybody else sees base + n dense records.
    uint64_t base;
    uint32_t n_stripes;
    // 1: the short-read scan folds records of equal key (ipx_scan_short.inc; off when the caller wants every record:
    // ipx_records_export)
    // 2: the same under an active junction filter (genome + "canon"): a record is folded only if IT would pass the
    // filter against its junction's motif (looked up in the scan, once per run of equal keys); records that fail stay
    // single, because the reduce drops those that precede the junction's creator one by one
    uint32_t combine;
    int32_t skip_missing;
    // the key rank of a tid is the tid itself (IPX_SCOPE_TARGET): the scan then skips the dependent load of
    // rank_of_tid[tid], one round trip per tile of short reads
    uint32_t rank_is_tid;
    GenomeDev genome;
};
constexpr uint32_t IPX_STRIPES = 32;                 // four per XCD
constexpr uint32_t IPX_STRIPE_CHUNK_BITS = 9;        // 512 records: stripe s owns chunks s, s + 32, s + 64, ...
// the stripe of a wave: the XCD it runs on (HW_REG_XCC_ID) and a number the kernel picks (its wave in the block)
#ifdef __HIPCC__
__device__ __forceinline__ uint32_t ipx_stripe_of(uint32_t xcc_id, uint32_t sub) { return ((xcc_id & 7u) << 2) | (sub & 3u); }
#endif
// The Counters allocation (8 KiB): Counters at 0, the copy of the CIGAR array's tail at 128 (ipx_scan.hip), length of the
// list of alignments the short-read scan leaves to k_scan_rest at 1536, number of records the scan's folds merged into
// others at 1544 (the words behind it: ipx_api.hip), k_close's count of finished blocks at 1664, stripe counters at
// 2048 + 128 * s.
constexpr uint32_t IPX_COUNTERS_BYTES = 8192, IPX_CNT_TAIL = 128, IPX_CNT_DEFER = 1536, IPX_CNT_MERGED = 1544,
                   IPX_CNT_CLOSE_DONE = 1664, IPX_CNT_STRIPE0 = 2048;

// device-side counters of one handle (one cache line apart is not needed: they
// are touched once per wave at most)
struct Counters {
    unsigned long long n_records;      // allocated so far (may exceed capacity)
    unsigned long long err_word;       // earliest data error, IPX_ERRW_NONE if none
    uint32_t key_or[3], key_and[3];    // OR / AND of all sort keys: bits that never vary need no pass
    uint32_t pad[2];                   // [0]: ipx_scan_ops met a CIGAR layout it cannot walk
};

// passes of a sort, least significant first
struct SortPlan {
    DigitSpec digits[16];
    uint32_t n;
};
// Passes that cover exactly the bits set in vary[0..2] (bits on which the keys differ), at most
// max_bits of them.  first: a window that is to be the first pass whatever varies in it (its digit
// bytes were written together with the records), or NULL.  The caller checks with
// ipx_window_is_first() that no varying bit below the window's top lies outside it.
#ifdef __HIPCC__
__host__ __device__
#endif
static inline SortPlan ipx_plan_sort(const uint32_t vary[3], uint32_t max_bits, const DigitSpec *first) {
    SortPlan plan{};
    uint32_t bit = 0;
    if (first) {
        plan.digits[plan.n++] = *first;
        bit = first->w2 ? first->lo2 + first->w2 : first->lo1 + first->w1;
    }
    auto varies = [&](uint32_t b) { return b < max_bits && ((vary[b >> 5] >> (b & 31u)) & 1u); };
    while (bit < max_bits && plan.n < 16) {
        while (bit < max_bits && !varies(bit)) ++bit;
        if (bit >= max_bits) break;
        DigitSpec d{bit, 0, 0, 0};
        while (d.w1 < 8 && varies(bit)) { ++d.w1; ++bit; }
        if (d.w1 < 8) {                                  // room left: take the start of the next run too
            uint32_t nb = bit;
            while (nb < max_bits && !varies(nb)) ++nb;
            if (nb < max_bits) {
                d.lo2 = nb;
                while (d.w1 + d.w2 < 8 && varies(nb)) { ++d.w2; ++nb; }
                bit = nb;
            }
        }
        plan.digits[plan.n++] = d;
    }
    if (plan.n == 0) plan.digits[plan.n++] = DigitSpec{0, 8, 0, 0};
    return plan;
}
// May window d be the least significant pass?  (every varying bit below its top is inside it)
static inline bool ipx_window_is_first(const uint32_t vary[3], uint32_t max_bits, const DigitSpec &d) {
    const uint32_t top = d.w2 ? d.lo2 + d.w2 : d.lo1 + d.w1;
    for (uint32_t b = 0; b < top && b < max_bits; ++b) {
        const bool inside = (b >= d.lo1 && b < d.lo1 + d.w1) || (d.w2 && b >= d.lo2 && b < d.lo2 + d.w2);
        if (!inside && ((vary[b >> 5] >> (b & 31u)) & 1u)) return false;
    }
    return true;
}
#ifdef __HIPCC__
// Lane mask of a predicate.  HIP's __ballot(int) first materialises the predicate as 0/1 in a vector
// register and compares it again (two vector instructions per ballot); the builtin takes the condition's
// lane mask as it is.
__device__ __forceinline__ uint64_t ipx_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// bits [lo, lo+w) of the 96-bit key (w <= 8)
__device__ __forceinline__ uint32_t ipx_key_bits(const uint4 &it, uint32_t lo, uint32_t w) {
    // lo is wave-uniform wherever this is used (a kernel argument): the word is picked by a scalar branch and
    // v_alignbit shifts the 64-bit pair in one instruction (selects + a 64-bit shift were ~7)
    const uint32_t word = lo >> 5, sh = lo & 31u;
    uint32_t v;
    if (word == 0) v = __builtin_amdgcn_alignbit(it.y, it.x, sh);
    else if (word == 1) v = __builtin_amdgcn_alignbit(it.z, it.y, sh);
    else v = it.z >> sh;
    return v & ((1u << w) - 1u);
}
__device__ __forceinline__ uint32_t ipx_digit_of(const uint4 &it, const DigitSpec &d) {
    uint32_t v = ipx_key_bits(it, d.lo1, d.w1);
    if (d.w2) v |= ipx_key_bits(it, d.lo2, d.w2) << d.w1;
    return v;
}
#endif


static inline uint64_t ipx_div_up(uint64_t a, uint64_t b) { return (a + b - 1) / b; }

#ifdef __HIPCC__
// ---- splice sites on the device genome (used per junction by k_motif, per record by the scan's combiner) ----
// two neighbouring 32-bit words at a word-aligned address, as one 8-byte load
struct __attribute__((packed, aligned(4))) IpxWords2 { uint32_t x, y; };

// splice_juncs.cc:22-26 on 2-bit codes A0 C1 G2 T3: GT/AG, GC/AG, AT/AC
__device__ __forceinline__ bool ipx_canon4(uint32_t d0, uint32_t d1, uint32_t a0, uint32_t a1) {
    const bool gt_ag = (d0 == 2 && d1 == 3 && a0 == 0 && a1 == 2);
    const bool gc_ag = (d0 == 2 && d1 == 1 && a0 == 0 && a1 == 2);
    const bool at_ac = (d0 == 0 && d1 == 3 && a0 == 0 && a1 == 1);
    return gt_ag || gc_ag || at_ac;
}

// get_splice_sites, junctions_extractor.cc:405-428, as a word: IPX_MOTIF_* bits | the four 2-bit bases;
// error marker in bits 11..8: 0x100 = sequence missing and not skippable (:411-417, genome.cc:65-67),
// 0x200 = a fetch ran past the end of the sequence (genome.cc:68-70)
__device__ __forceinline__ uint32_t ipx_motif_of(const GenomeDev &g, int32_t skip_missing, uint32_t rank,
                                                 uint32_t start, uint32_t end) {
    if (!g.seq_of_rank) return 0u;                              // no genome: splice_sites == ""
    const RankSeq rs = g.rank_seq[rank];
    if (rs.len < 0) return skip_missing ? 0u : 0x100u;
    const int64_t L = rs.len;
    const int64_t s = (int64_t)(int32_t)start, e = (int64_t)(int32_t)end;   // Genome::fetch takes int
    // faidx clamps to the sequence; a fetch that returns fewer than 2 bases throws.
    // acceptor [e-2,e) is fetched first (see oracle/ipxo_core.c get_splice_sites)
    if (e - 2 < 0 || e > L || s < 0 || s + 2 > L) return 0x200u;
    const uint64_t w2 = rs.word2, wm = rs.wordm;
    // bases p, p + 1 lie in one 2-bit word or in two neighbours: ONE 8-byte load per pair (word-aligned is enough for a
    // dwordx2 load; the arrays end in a spare word) instead of two 4-byte ones -- the look-up per RECORD in the scan's
    // filter-aware combiner is bound by the number of scattered loads, not by their bytes.  The same for the mask bits.
    auto pair2 = [&](int64_t p, uint32_t &b0, uint32_t &b1) {
        const IpxWords2 w = *reinterpret_cast<const IpxWords2 *>(g.twobit + w2 + (uint64_t)(p >> 4));
        const uint64_t v = (uint64_t)w.x | ((uint64_t)w.y << 32);
        const uint32_t sh = (uint32_t)(p & 15) * 2u;
        b0 = (uint32_t)(v >> sh) & 3u;
        b1 = (uint32_t)(v >> (sh + 2u)) & 3u;
    };
    auto mask2 = [&](int64_t p) -> uint32_t {
        const IpxWords2 w = *reinterpret_cast<const IpxWords2 *>(g.mask + wm + (uint64_t)(p >> 5));
        const uint64_t v = (uint64_t)w.x | ((uint64_t)w.y << 32);
        return (uint32_t)(v >> (p & 31)) & 3u;
    };
    uint32_t d0, d1, a0, a1;
    pair2(s, d0, d1);
    pair2(e - 2, a0, a1);
    const uint32_t non = mask2(s) | mask2(e - 2);
    uint32_t m = IPX_MOTIF_PRESENT | (d0 << 6) | (d1 << 4) | (a0 << 2) | a1;
    if (non) {
        m |= IPX_MOTIF_NONACGT;
    } else {
        if (ipx_canon4(d0, d1, a0, a1)) m |= IPX_MOTIF_CANON_P;
        // reverse complement of "d0d1/a0a1" is  ~a1 ~a0 / ~d1 ~d0
        if (ipx_canon4(3 - a1, 3 - a0, 3 - d1, 3 - d0)) m |= IPX_MOTIF_CANON_RC;
    }
    return m;
}

// does a record with read strand s create the junction?  (add_junction + junction_filter_check,
// junctions_extractor.cc:488-506, splice_juncs.cc:52-61)
__device__ __forceinline__ bool ipx_record_passes(uint32_t strand, uint32_t motif, int32_t sj_filter) {
    if (!(motif & IPX_MOTIF_PRESENT)) return true;          // empty splice_sites pass any filter
    if (sj_filter != IPX_SJ_CANON) return true;
    const bool cp = motif & IPX_MOTIF_CANON_P, crc = motif & IPX_MOTIF_CANON_RC;
    if (strand == '-') return crc;
    if (strand == '.') return cp || crc;
    return cp;
}
#endif

#ifdef __HIPCC__
// fields of a packed sort key (see the record layout above)
struct KeyParts { uint32_t rank, start, end, off16; };

__device__ __forceinline__ KeyParts unpack_key(const uint4 &it, const KeyLayout &kl) {
    const uint64_t lo = (uint64_t)it.x | ((uint64_t)it.y << 32);
    const uint64_t hi = it.z;
    KeyParts k;
    k.off16 = (uint32_t)lo & 0xffffu;
    const uint32_t len = (uint32_t)((lo >> 16) & ((kl.len_bits >= 32) ? 0xffffffffull : ((1ull << kl.len_bits) - 1)));
    const uint32_t s1 = 16 + kl.len_bits, s2 = s1 + 32;
    uint64_t st = lo >> s1;
    if (s1 > 32) st |= hi << (64 - s1);
    k.start = (uint32_t)st;
    uint64_t rk = (s2 < 64) ? ((lo >> s2) | (hi << (64 - s2))) : (hi >> (s2 - 64));
    k.rank = (uint32_t)(rk & ((kl.rank_bits >= 32) ? 0xffffffffull : ((1ull << kl.rank_bits) - 1)));
    k.end = k.start + len;
    return k;
}

#endif

// ---- per-kernel event timing (ipx_profile / ipx_profile_read) -----------------
// Launchers call ipx_prof_mark(stream, id) right after a launch; the interval since
// the previous mark on the stream is charged to `id`.  A no-op unless profiling is on.
enum IpxKernelId {
    IPXK_BEGIN = 0,      // opens an interval (after a host sync); never charged
    IPXK_SCAN, IPXK_SORT_HIST, IPXK_SORT_SCAN, IPXK_SORT_SKIP, IPXK_SORT_SCATTER,
    IPXK_COUNT_HEADS, IPXK_SCAN_U32, IPXK_WRITE_HEADS, IPXK_MOTIF, IPXK_MEMSET, IPXK_PHASE_A,
    IPXK_PHASE_B, IPXK_FINISH_J, IPXK_BINS, IPXK_EMIT_BINS, IPXK_EMIT_J, IPXK_EMIT_IJUNC,
    IPXK_COPY, IPXK_PARTITION, IPXK_ADOPT, IPXK_RANK_SORT, IPXK_BS_SAMPLE, IPXK_BS_ASSIGN, IPXK_BS_BOUNDS, IPXK_BS_LEAF, IPXK_COUNT
};
void ipx_prof_mark(hipStream_t s, int id);
// before a launch of class id: opens the interval when that class alone is being timed (ipx_profile_only)
void ipx_prof_pre(hipStream_t s, int id);

// words of the `defer` workspace ipx_launch_scan needs for a batch of n alignments (k_scan_rest's list)
static inline size_t ipx_scan_list_words(uint64_t n) { return (size_t)n + 64; }

// ---- launchers implemented in the kernel translation units -------------------
// ipx_scan.hip
// p.n_stripes / p.base are set by the launcher (striped when the grid is large and close_ws is given;
// close_ws: room for 2 * close_cap 64-bit entries; defer: room for one uint32 per alignment of the batch)
hipError_t ipx_launch_scan(hipStream_t s, ScanParams p, const ipx_batch &b,
                           const uint32_t *rank_of_tid, uint4 *items, uint4 *pays, uint32_t *cnts, uint8_t *digs,
                           uint64_t capacity, Counters *counters, uint64_t n_records_before,
                           unsigned long long *close_ws, uint64_t close_cap, uint32_t *defer, uint32_t rest_hint);
// ipx_scan_ops.hip (built into libipx_scanops.so, not into libipx.so): the same scan with one lane per CIGAR operation
// (needs back-to-back CIGAR words; raises Counters::pad[0] otherwise and the caller re-runs ipx_launch_scan)
hipError_t ipx_launch_scan_ops(hipStream_t s, const ScanParams &p, const ipx_batch &b,
                               const uint32_t *rank_of_tid, uint4 *items, uint4 *pays, uint32_t *cnts, uint8_t *digs,
                               uint64_t capacity, Counters *counters);

// ipx_sort.hip
size_t ipx_sort_workspace_bytes(uint64_t n);
// result in buf_b when plan.n is odd, else in buf_a; dig_a/dig_b: digit-byte streams or NULL
// prof_all: when >= 0, every kernel of this sort is charged to that profiler id
// after_last: when given, the last pass also writes that digit's bytes next to the items it moves (dig_a for an
// even number of passes, dig_b for an odd one) -- the bucket sort reads its leaf boundaries from them
hipError_t ipx_launch_sort(hipStream_t s, uint4 *buf_a, uint4 *buf_b, uint8_t *dig_a, uint8_t *dig_b,
                           bool first_bytes_valid, uint64_t n, const SortPlan &plan, void *workspace,
                           size_t workspace_bytes, int prof_all = -1, const DigitSpec *after_last = nullptr);

// ipx_bsort.hip: the bucket sort (sampled splitters -> two radix passes on a 16-bit leaf id -> every leaf sorted
// in LDS).  Sorts buf_a in place (buf_b, dig_a, dig_b are scratch).  *overflow (device word) is set when a leaf
// did not fit LDS: buf_a then holds the records in some order, keys intact, and the caller runs ipx_launch_sort.
bool ipx_bsort_applicable(uint64_t n, const KeyLayout &kl);
size_t ipx_bsort_workspace_bytes(void);
hipError_t ipx_launch_bsort(hipStream_t s, uint4 *buf_a, uint4 *buf_b, uint8_t *dig_a, uint8_t *dig_b, uint64_t n,
                            const KeyLayout &kl, const uint32_t vary[3], void *sort_ws, size_t sort_ws_bytes,
                            void *bs_ws, uint32_t *overflow);
// in == out allowed; block_sums needs div_up(n,4096)+1 words; total may be NULL
hipError_t ipx_launch_scan_u32(hipStream_t s, const uint32_t *in, uint32_t *out, uint64_t n,
                               uint32_t *block_sums, uint32_t *total);
// two independent arrays in the same three launches (n2 == 0: one); block_sums: ipx_scan_sums_words(n1, n2) words
size_t ipx_scan_sums_words(uint64_t n1, uint64_t n2);
hipError_t ipx_launch_scan_u32_pair(hipStream_t s, const uint32_t *in1, uint32_t *out1, uint64_t n1, uint32_t *total1,
                                    const uint32_t *in2, uint32_t *out2, uint64_t n2, uint32_t *total2,
                                    uint32_t *block_sums);

// ipx_bam.hip: BAM alignment records (spans = runs of whole records) -> the batch arrays; *err: 0xffffffff, or
// 1 + the index of the first record that does not parse.  workspace: ipx_bam_workspace_bytes(n)
size_t ipx_bam_workspace_bytes(uint64_t n);
hipError_t ipx_launch_bam_decode(hipStream_t s, const uint8_t *bytes, const uint64_t *span_begin, const uint64_t *span_first,
                                 uint32_t n_spans, uint64_t n, uint64_t n_words, const int32_t *tid_map, uint32_t n_tid_map,
                                 int32_t *tid, int32_t *pos, uint16_t *flag, uint8_t *xs, uint8_t *nh, uint32_t *cigar_off,
                                 uint32_t *n_cigar, uint32_t *cigar, void *workspace, uint32_t *err);

// the same in two stages, for a caller that learns the number of CIGAR words from the records themselves (ipx_decode_bgzf)
hipError_t ipx_launch_bam_decode_fields(hipStream_t s, const uint8_t *bytes, const uint64_t *span_begin, const uint64_t *span_first,
                                        uint32_t n_spans, uint64_t n, const int32_t *tid_map, uint32_t n_tid_map,
                                        int32_t *tid, int32_t *pos, uint16_t *flag, uint8_t *xs, uint8_t *nh, uint32_t *cigar_off,
                                        uint32_t *n_cigar, void *workspace, uint32_t *err, uint32_t **total_out);
hipError_t ipx_launch_bam_decode_cigar(hipStream_t s, const uint8_t *bytes, uint64_t n, uint64_t n_words, const uint32_t *cigar_off,
                                       const uint32_t *n_cigar, uint32_t *cigar, void *workspace, uint32_t *err);

// ipx_inflate.hip: BGZF blocks inflated (and CRC-checked) on the device, records counted per block
size_t ipx_bgzf_tables_bytes(uint64_t n_blocks);
// order (device, n_blocks words, or null): the block the n-th lane takes -- longest first, so that a wave's lanes finish together
hipError_t ipx_launch_bgzf_inflate(hipStream_t s, const uint8_t *comp, const uint64_t *in_begin, const uint64_t *in_end,
                                   const uint32_t *isize, const uint64_t *out_off, const uint32_t *crc, uint64_t n_blocks,
                                   uint8_t *out, uint32_t *tables, uint32_t *status, const uint32_t *order);
hipError_t ipx_launch_bam_count_blocks(hipStream_t s, const uint8_t *bytes, const uint64_t *out_off, const uint32_t *isize,
                                       uint64_t n_blocks, uint64_t records_begin, uint32_t *count, uint32_t *prefix, uint32_t *sums,
                                       uint32_t *total, uint64_t *span_begin, uint64_t *span_first, uint32_t *unaligned);

// ipx_exchange.hip: key-range partitioning of the pending records for the multi-GPU exchange
hipError_t ipx_launch_sample(hipStream_t s, const uint4 *items, uint64_t n, const KeyLayout &kl,
                             uint32_t n_samples, unsigned long long *out /* device */);
uint32_t ipx_partition_blocks(uint64_t n);
// count, plan and scatter in one go: the records (item, payload pairs; the item's fourth word carries the
// record's count word on the way) of part p end up one after the other in out_records, parts in order;
// totals (device, 64 words) receives the part sizes.
// blk: device scratch, 64 words per block of ipx_partition_blocks(n)
hipError_t ipx_launch_partition(hipStream_t s, const uint4 *items, const uint4 *pays, const uint32_t *cnts, uint64_t n,
                                const KeyLayout &kl, uint32_t n_parts,
                                const unsigned long long *splitters /* device */, unsigned long long *blk,
                                unsigned long long *totals, uint4 *out_records);
hipError_t ipx_launch_adopt(hipStream_t s, const uint4 *records, uint64_t n, uint64_t first_index, uint4 *items,
                            uint4 *pays, uint32_t *cnts, uint8_t *digs, DigitSpec first, Counters *counters);
// the slotted form (equal splits, counts and flags in slot headers: see k_part_plan_slots / k_adopt_slots)
hipError_t ipx_launch_partition_slots(hipStream_t s, const uint4 *items, const uint4 *pays, const uint32_t *cnts, uint64_t n,
                                      const KeyLayout &kl, uint32_t n_parts, const unsigned long long *splitters,
                                      unsigned long long *blk, unsigned long long *totals, uint64_t slot_records,
                                      unsigned long long error_order, uint32_t error_status, uint4 *out_slots,
                                      uint4 *spill = nullptr, uint64_t spill_records = 0);
hipError_t ipx_launch_adopt_slots(hipStream_t s, const uint4 *slots, uint32_t n_slots, uint64_t slot_records, uint64_t first_index,
                                  uint4 *items, uint4 *pays, uint32_t *cnts, uint8_t *digs, DigitSpec first, Counters *counters,
                                  uint32_t *info /* device, 16 words */);
