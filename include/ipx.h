/* ipx.h -- C ABI of the MI355X junction-extraction library (libipx.so).
 *
 * This is the drop-in boundary for the one hot path of diekhans/intronProspector
 * v1.4.0 that runs on the GPU: alignments (already decoded into SoA batches) ->
 * junction records -> per-junction aggregate table.  The reference has no FFI;
 * its de-facto interface for this path is the C++ class JunctionsExtractor
 * (src/junctions_extractor.hh:73-256).  Each entry point below names the
 * reference member it stands for (paths relative to /root/reference).
 *
 * Conventions: plain C types only; every function returns 0 (IPX_OK) or an
 * IPX_E* code and never throws; one handle is used from one thread at a time;
 * different handles are independent.  The library has no CPU fallback: if no
 * gfx950 device can be opened ipx_create fails with IPX_ENODEVICE.
 */
#ifndef IPX_H
#define IPX_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IPX_ABI_VERSION 5

/* status codes */
enum {
    IPX_OK = 0,
    IPX_EINVAL = 1,        /* bad argument / unsupported configuration           */
    IPX_ENODEVICE = 2,     /* no usable HIP device (no CPU fallback exists)       */
    IPX_ENOMEM = 3,        /* device or pinned-host allocation failed             */
    IPX_EDEVICE = 4,       /* HIP runtime error; see ipx_last_error().detail      */
    IPX_EDATA = 5          /* the input raised one of the reference's exceptions;
                              see ipx_last_error() for which and where            */
};

/* data errors, ipx_error.kind: the reference exception each one stands for */
enum {
    IPX_DATA_NONE = 0,
    IPX_DATA_CIGAR = 1,        /* invalid_argument "Unknown cigar operation 'P'"
                                  junctions_extractor.cc:312-313; op char in .op  */
    IPX_DATA_CATEGORY = 2,     /* logic_error "should never look at FQCAIL, FUNMAP,
                                  FDUP reads"  junctions_extractor.cc:73-74        */
    IPX_DATA_GENOME_MISSING = 3, /* runtime_error "can't load genome sequence for:"
                                  genome.cc:65-67; coordinates in .start/.end      */
    IPX_DATA_GENOME_SHORT = 4, /* runtime_error "unexpected length returned for:"
                                  genome.cc:68-70; returned length in .retlen      */
    IPX_DATA_TARGET = 5        /* tid >= number of targets (undefined behaviour in
                                  the reference, junctions_extractor.cc:538)       */
};

/* splice_juncs.hh:12-16 */
enum { IPX_SJ_NULL = 0, IPX_SJ_ALL = 1, IPX_SJ_CANON = 2 };
/* junctions_extractor.hh:56-60 */
enum { IPX_UNSTRANDED = 0, IPX_RF_STRANDED = 1, IPX_FR_STRANDED = 2 };
/* junctions_extractor.hh:64-67 */
enum { IPX_EXCLUDE_NONE = 0, IPX_EXCLUDE_MULTI = 1 };
/* table scope: which JunctionTable lifetime the caller reproduces */
enum {
    IPX_SCOPE_TARGET = 0,  /* one table per target, intronProspector.cc:364-368:
                              key = (tid, start, end)                              */
    IPX_SCOPE_GLOBAL = 1   /* one table for the run (--unsorted / several inputs),
                              intronProspector.cc:375-401: key = (chrom NAME,
                              start, end), names ordered as std::string            */
};

/* by-value parameters of the JunctionsExtractor constructor,
 * junctions_extractor.hh:176-207 (pass-through/tag/trace members excluded) */
typedef struct ipx_params {
    uint32_t min_anchor_length;       /* default 8       intronProspector.cc:44 */
    uint32_t max_anchor_indel_size;   /* 0 unless --allow-anchor-indels :289-292 */
    uint32_t min_intron_length;       /* default 70      :45 */
    uint32_t max_intron_length;       /* default 500000  :46 */
    int32_t  sj_filter;               /* IPX_SJ_*  */
    int32_t  strandness;              /* IPX_UNSTRANDED / RF / FR */
    uint32_t excludes;                /* IPX_EXCLUDE_* bits */
    int32_t  scope;                   /* IPX_SCOPE_* */
    int32_t  skip_missing_targets;    /* --skip-missing-targets */
    int32_t  device;                  /* HIP device ordinal */
} ipx_params;

/* One batch of decoded alignments, structure of arrays, 20 bytes per alignment
 * plus 4 bytes per CIGAR operation.  This is what the reference reads from a
 * bam1_t in process_alignment (junctions_extractor.cc:49-56,71-95,184-205,
 * 285-290,322-371,533-536):
 *   tid        core.tid
 *   pos        core.pos (0-based)
 *   flag       core.flag
 *   xs         0 if the record has no XS tag, else the byte
 *              get_junction_strand_XS() returns ('.' when bam_aux2A yields 0)
 *   nh         get_num_aligns() class: 0 = no NH tag or value <= 0,
 *              1 = NH == 1, 2 = NH > 1
 *   cigar_off  index of the record's first word in cigar[]
 *   n_cigar    core.n_cigar
 *   cigar      bam_get_cigar() words, BAM packing (len << 4 | op)
 * The caller lists alignments in the order the reference's read loop would
 * hand them to process_alignment; base_read_index is the position of
 * alignment 0 in that order over the whole run.
 */
typedef struct ipx_batch {
    uint64_t n_alignments;
    uint64_t n_cigar_words;
    uint64_t base_read_index;
    const int32_t  *tid;
    const int32_t  *pos;
    const uint16_t *flag;
    const uint8_t  *xs;
    const uint8_t  *nh;
    const uint32_t *cigar_off;
    const uint32_t *n_cigar;
    const uint32_t *cigar;
    int32_t memory;            /* IPX_MEM_HOST or IPX_MEM_DEVICE */
} ipx_batch;

enum { IPX_MEM_HOST = 0, IPX_MEM_DEVICE = 1 };

/* motif byte of a junction: what get_splice_sites() fetched
 * (junctions_extractor.cc:405-428) */
#define IPX_MOTIF_PRESENT 0x8000u  /* genome consulted and sequence present        */
#define IPX_MOTIF_NONACGT 0x4000u  /* at least one of the four bases is not ACGT:
                                      the caller must take the bytes from its
                                      FASTA to print them                          */
#define IPX_MOTIF_CANON_P 0x2000u  /* is_canonical(donor/acceptor as read)         */
#define IPX_MOTIF_CANON_RC 0x1000u /* is_canonical(reverse complement)             */
/* bits 7..0: 2-bit codes (A0 C1 G2 T3) of genome[start], [start+1], [end-2],
 * [end-1], first base in bits 7..6 */

/* The aggregate table, structure of arrays, rows in std::map<JunctionKey,...>
 * order (junctions.hh:77-100).  Stands for JunctionTable / get_junctions()
 * (junctions.hh:225-246, junctions_extractor.hh:253-255) with the fields the
 * printers read (junctions.cc:189-225).  Memory is owned by the handle and
 * stays valid until the next ipx_finalize / ipx_clear / ipx_destroy. */
typedef struct ipx_table {
    uint64_t n_junctions;
    const uint32_t *target;        /* tid (IPX_SCOPE_TARGET) or the first tid that
                                      carries the chromosome name (IPX_SCOPE_GLOBAL) */
    const uint32_t *intron_start;
    const uint32_t *intron_end;
    const uint32_t *anchor_start;  /* min over counted records */
    const uint32_t *anchor_end;    /* max over counted records */
    const uint8_t  *strand;        /* Junction::strand after correct_strand() */
    const uint16_t *motif;         /* IPX_MOTIF_* */
    const uint64_t *read_counts;   /* 3 per junction: single, multi, unsure */
    const uint64_t *creator_order; /* creating record: read_index << 20 | number of the
                                      record inside its alignment; orders creations */
    const uint32_t *ijunc;         /* rank of creation inside this table; the
                                      caller adds the run's running ijunc_
                                      (junctions_extractor.hh:129,202)             */
    /* read_offsets multiset (junctions.hh:121) as a histogram, CSR:
     * bins hist_begin[j] .. hist_begin[j+1]-1, values ascending */
    const uint64_t *hist_begin;    /* n_junctions + 1 */
    const uint16_t *hist_value;
    const uint32_t *hist_count;
    uint64_t n_hist;
    /* statistics of the finalized batches */
    uint64_t n_alignments;
    uint64_t n_records;            /* junction records that passed junction_qc */
    int32_t  memory;               /* IPX_MEM_HOST or IPX_MEM_DEVICE */
} ipx_table;

typedef struct ipx_error {
    int32_t  status;      /* last non-zero status */
    int32_t  kind;        /* IPX_DATA_* when status == IPX_EDATA */
    uint64_t read_index;  /* global index of the alignment that raised it */
    uint32_t target;      /* its tid */
    uint32_t start, end;  /* genome errors: the [start,end) that was fetched */
    int32_t  retlen;      /* IPX_DATA_GENOME_SHORT: bytes the fetch returned */
    char     op;          /* IPX_DATA_CIGAR: the operation character */
    char     detail[235]; /* human-readable text (HIP error string, ...) */
    uint64_t order;       /* IPX_EDATA: where the error lies in input order (alignment, then record inside
                             the alignment, then kind); smaller = earlier, comparable between the handles
                             of one run: a multi-GPU caller raises the error with the smallest order
                             (junctions_extractor.cc throws at the first one its loop meets).
                             ~0 for every other status */
} ipx_error;

typedef struct ipx_handle ipx_handle;

uint32_t ipx_abi_version(void);

/* JunctionsExtractor::JunctionsExtractor, junctions_extractor.hh:176-207 */
int ipx_create(const ipx_params *params, ipx_handle **out);
/* JunctionsExtractor::~JunctionsExtractor, junctions_extractor.cc:375-380 */
int ipx_destroy(ipx_handle *h);

/* save_targets(), junctions_extractor.cc:564-568: REPLACES the handle's target
 * list with names[0..n) (the caller keeps the reference's append-per-open
 * concatenation and passes the whole vector).  lengths may be NULL; when given
 * (header LN) they only tighten the sort-key width. */
int ipx_set_targets(ipx_handle *h, uint32_t n, const char *const *names, const int64_t *lengths);

/* Genome (genome.hh:15-25), device-resident: n_seqs sequences, each given as
 * 2-bit codes (4 bases per byte, first base in the low bits) and a non-ACGT
 * mask (8 bases per byte, first base in bit 0).  Sequence names are matched to
 * target names the way Genome::has_seq/fetch are called with the chromosome
 * name (junctions_extractor.cc:411,419-420).  n_seqs == 0 removes the genome. */
int ipx_set_genome_2bit(ipx_handle *h, uint32_t n_seqs, const char *const *names,
                        const int64_t *lengths, const uint8_t *const *twobit,
                        const uint8_t *const *mask);

/* Handle options.
 *   IPX_OPT_COMBINE (default 1)  the short-read scan may merge, tile by tile, junction records that share
 *       junction AND read offset into one record that stands for all of them (ipx_table is unaffected: the
 *       merged record carries the group's counts, extreme anchors and earliest order).  Set it to 0 before
 *       the first push when the caller wants every single record back (ipx_records_export: --debug-trace,
 *       strand tags in the pass-through, junctions_extractor.cc:451-457,471-477,526-528). */
enum { IPX_OPT_COMBINE = 1 };
int ipx_set_option(ipx_handle *h, int32_t option, int64_t value);

/* Optional: size device buffers once, before timing. */
int ipx_reserve(ipx_handle *h, uint64_t n_alignments, uint64_t n_cigar_words, uint64_t n_records);

/* process_alignment() over a batch (junctions_extractor.cc:533-549): scans the
 * CIGAR streams on the device and appends the QC-passed junction records to
 * the handle.  Batch memory is borrowed until the call returns. */
int ipx_push_batch(ipx_handle *h, const ipx_batch *batch);

/* The same, overlapped with the caller's own work: ipx_push_batch_async returns as soon as the copies to the
 * device and the scan are QUEUED on the handle's stream; the batch memory stays borrowed until ipx_sync (or
 * any other call on the handle, each of which first books the batch in flight) returns.  At most one batch is
 * in flight: a second ipx_push_batch_async first completes the first.  For the copies to run asynchronously
 * and at full PCIe rate the batch arrays should live in pinned host memory from ipx_host_alloc: the caller
 * fills one set of arrays while the device reads the other (host/junctions_extractor.cc does that).
 * Stands for the same process_alignment loop (junctions_extractor.cc:533-549,656-664). */
int ipx_push_batch_async(ipx_handle *h, const ipx_batch *batch);
int ipx_sync(ipx_handle *h);

/* A batch as the caller's BAM reader holds it: alignment records of the BAM stream (SAM spec 4.2, each with its
 * block_size word: refID, pos, l_read_name, mapq, bin, n_cigar_op, flag, l_seq, next_refID, next_pos, tlen,
 * read_name, cigar, seq, qual, tags), back to back in `bytes`.  The device reads of each record what
 * process_alignment reads of a bam1_t (junctions_extractor.cc:533-549): core.tid, core.pos, core.flag, the CIGAR
 * (bam_get_cigar; the CG tag's words behind a "<l_seq>S<n>N" placeholder, SAM spec 4.2.2), bam_aux_get("XS") +
 * bam_aux2A (:322-330), bam_aux_get("NH") + bam_aux2i (:49-56) -- so a caller needs no per-record loop of its
 * own.  The caller only says where runs of whole records lie: span k is bytes [span_begin[k], span_begin[k+1])
 * and holds records span_first[k] .. span_first[k+1]-1 of the batch (an inflated BGZF block, minus the record
 * that straddles its end, is one span; both tables have n_spans + 1 entries, start at 0 and end at n_bytes /
 * n_alignments).  tid_map (or NULL): refID -> the tid the hot path is to see.  n_cigar_words: the CIGAR words
 * of all records (CG-tag CIGARs counted as such); a count the records do not bear out, and any record that does
 * not parse (what sam_read1 returns < -1 for), makes the completing call return IPX_EINVAL with the batch dropped.
 * memory: IPX_MEM_HOST (pinned memory from ipx_host_alloc for an asynchronous copy).  Otherwise as ipx_push_batch /
 * ipx_push_batch_async. */
typedef struct ipx_bam_batch {
    uint64_t n_alignments;
    uint64_t n_cigar_words;
    uint64_t base_read_index;
    const uint8_t  *bytes;
    uint64_t n_bytes;
    const uint64_t *span_begin;
    const uint64_t *span_first;
    uint32_t n_spans;
    const int32_t  *tid_map;
    uint32_t n_tid_map;
    int32_t memory;
} ipx_bam_batch;
int ipx_push_bam(ipx_handle *h, const ipx_bam_batch *batch);
int ipx_push_bam_async(ipx_handle *h, const ipx_bam_batch *batch);

/* The inflater of ipx_decode_bgzf on its own (tests, timing): BGZF blocks inflated on the device, one lane per block --
 * what htslib's bgzf.c does block by block under sam_read1 (junctions_extractor.cc:620-634).  comp[0, n_bytes):
 * the file's bytes (host memory); block b's raw DEFLATE stream is comp[in_off[b], in_off[b + 1]) (in_off has
 * n_blocks + 1 entries; the BGZF header and the CRC / ISIZE trailer lie outside), its ISIZE is isize[b] (<= 65536)
 * and its bytes go to out[out_off[b], out_off[b] + isize[b]) (host memory, out_bytes long).  status[b]: 0, or why
 * the block did not inflate to exactly isize[b] bytes.  CRCs are not checked.  kernel_ms (or NULL): the kernel alone,
 * without the copies.  Stateless: allocates and frees what it needs. */
int ipx_bgzf_inflate(int32_t device, const uint8_t *comp, uint64_t n_bytes, const uint64_t *in_off,
                     const uint32_t *isize, const uint64_t *out_off, uint64_t n_blocks, uint8_t *out,
                     uint64_t out_bytes, uint32_t *status, float *kernel_ms);

/* A whole BGZF-compressed BAM file decoded on the device (ABI 3) -- what sam_read1 + bgzf.c do record by record and
 * block by block in front of process_alignment (junctions_extractor.cc:620-634): the caller reads the FILE's bytes,
 * walks the block headers (payload_begin / payload_end: block b's raw DEFLATE stream; isize / crc32: its trailer) and
 * says where the alignment records begin in the inflated stream (records_begin: the header's length).  The device
 * inflates every block (one lane per block), checks every CRC-32, finds the records -- every block must begin at a
 * record boundary, which htslib's writers see to (bgzf_flush_try before every record); a file that lets records
 * straddle block ends is refused, the caller then reads it the serial way -- and decodes them as ipx_push_bam does,
 * into ONE device-resident SoA batch of all alignments: *out (memory = IPX_MEM_DEVICE, base_read_index = 0), owned by
 * the handle and valid until the next ipx_decode_bgzf / ipx_destroy.  Slices of it (pointers advanced, cigar and
 * n_cigar_words kept: cigar_off stays absolute) go to ipx_push_batch target by target.
 * Returns IPX_EINVAL with the reason in ipx_last_error().detail when anything about the file does not check out
 * (a block that does not inflate to its ISIZE, a CRC mismatch, an unaligned block, a record that does not parse):
 * nothing was pushed; the caller's own reader then reports the damage in its words. */
typedef struct ipx_bgzf_file {
    const uint8_t  *bytes;          /* the file, host memory */
    uint64_t n_bytes;
    uint64_t n_blocks;
    const uint64_t *payload_begin;  /* n_blocks offsets into bytes */
    const uint64_t *payload_end;
    const uint32_t *isize;
    const uint32_t *crc32;
    uint64_t records_begin;
    const int32_t  *tid_map;        /* refID -> tid, or NULL */
    uint32_t n_tid_map;
} ipx_bgzf_file;
int ipx_decode_bgzf(ipx_handle *h, const ipx_bgzf_file *file, ipx_batch *out);
/* pinned (page-locked) host memory, hipHostMalloc / hipHostFree */
int ipx_host_alloc(void **out, size_t bytes);
int ipx_host_free(void *p);

/* Sort + reduce every record pushed since the last ipx_clear into the junction
 * table.  memory selects where the returned arrays live.  If the input raised a
 * reference exception the call returns IPX_EDATA and ipx_last_error() tells
 * which alignment and why (the earliest one in input order).  The reference
 * throws at that alignment and its caller never sees a table; *out is then
 * filled from the records of all other alignments and has no reference
 * counterpart.
 * Which sort: LSD radix passes, or -- from 8 M records on -- a bucket sort whose leaves are sorted in LDS.  The leaf
 * sort takes an item's rank among the equal digits of its wave from the value an LDS returning atomic hands back, i.e.
 * it rests on lanes of one instruction that hit the same LDS word being served in lane order: measured (1.3 x 10^10
 * lane-operations without an exception, profiles/microbench/lds_atomic_order.hip), checked again at start-up in every
 * process (the bucket sort stays off if the check fails), NOT a documented property of the hardware.  Every sort is
 * followed by a check that neighbouring items ascend, and a table is never reduced from items that do not: they are
 * sorted again by the radix passes.  IPX_SORT=lsd in the environment never takes the bucket sort. */
int ipx_finalize(ipx_handle *h, int32_t memory, ipx_table *out);

/* JunctionsExtractor::clear(), junctions_extractor.hh:223-225: drops records and
 * table; targets, genome and parameters stay. */
int ipx_clear(ipx_handle *h);

int ipx_last_error(const ipx_handle *h, ipx_error *out);

/* --- multi-GPU support: partial tables over a contiguous shard
 * of the run's alignments, to be combined with RCCL by the caller ------------ */

/* ipx_finalize for ONE contiguous shard of a larger run.  A shard cannot know
 * whether an earlier shard already created a key, so two tables come back, rows
 * in the same key order:
 *   out_from_creator  what ipx_finalize returns: keys with a creating record in
 *                     this shard, tallies from that record on;
 *   out_all           every key of the shard with the tallies over ALL its records:
 *                     what the junction receives when an earlier shard created it.
 * When no record can fail the filter (no genome, or sj_filter != canon) the two
 * are the same table (same pointers).  The caller combines shards: the creator is
 * the smallest creator_order over shards; shards before it contribute nothing,
 * that shard its from_creator row, later shards their out_all row. */
int ipx_finalize_shard(ipx_handle *h, int32_t memory, ipx_table *out_from_creator, ipx_table *out_all);

/* Record exchange (the faster multi-GPU path when tables are not much smaller than the
 * records): every rank splits its pending records at the same splitters on the partition
 * key (chromosome rank << 32 | intron start; all records of a junction share it), the
 * caller exchanges the parts with one all_to_all over RCCL, each rank adopts what it
 * received and calls ipx_finalize on the key range it now owns.  Records carry their
 * global order, so the result does not depend on arrival order.
 *   ipx_records_sample     partition keys of n_samples (<= 4096) evenly spaced pending records
 *   ipx_records_partition  n_parts (<= 64) parts: part p holds the keys in
 *                          [splitters[p-1], splitters[p]); counts[p] records each; the
 *                          32-byte records are written part after part to dst_records
 *                          (device memory, room for ipx_records_pending records).  A
 *                          record may stand for several reads (IPX_OPT_COMBINE); its
 *                          counts travel inside the 32 bytes
 *   ipx_records_adopt      replaces the handle's pending records by n received ones */
int ipx_records_pending(const ipx_handle *h, uint64_t *n_records);
int ipx_records_sample(ipx_handle *h, uint32_t n_samples, uint64_t *out_keys);
int ipx_records_partition(ipx_handle *h, uint32_t n_parts, const uint64_t *splitters, uint64_t *counts,
                          void *dst_records);
int ipx_records_adopt(ipx_handle *h, uint64_t n, const void *records, uint64_t n_alignments);

/* The same exchange without a host in the loop (ABI 3): parts go to fixed-capacity SLOTS of slot_records 32-byte
 * records each, slot p at dst_slots + p * slot_records * 32.  A slot begins with a header of 1 + (n_parts + 7) / 8
 * records written on the device -- the part's true count, the sender's largest part, a flag "some part of this sender did not fit its slot", and the
 * sender's own error (error_order = ipx_error.order of what this rank found wrong with its input, ~0 if nothing;
 * error_status its ipx status) -- so one all-to-all with EQUAL splits carries everything the receivers need:
 *   ipx_records_partition_slots  queues the kernels and returns (no read-back); dst_slots: device memory, n_parts slots
 *   ipx_records_adopt_slots      n_slots received slots -> the handle's pending records; info: what the headers said.
 *                                info->overflow != 0: some part somewhere was cut short -- the table would be wrong;
 *                                repeat scan + exchange with slots of more than info->max_part records.
 *                                info->error_order != ~0: sender info->error_slot reported the earliest error of the run
 * Both take the handle's stream: with ipx_use_stream the caller's collective needs no host synchronisation between. */
typedef struct ipx_slots_info {
    uint64_t n_records;      /* adopted */
    uint64_t max_part;       /* the largest part any sender had for any receiver */
    uint32_t overflow;
    uint32_t error_slot;
    uint64_t error_order;
    int32_t error_status;
    int32_t reserved;
    uint64_t max_received;   /* records the busiest rank receives, and what all ranks receive together: the same */
    uint64_t all_received;   /* numbers on every rank (every header carries its sender's counts for all parts) */
} ipx_slots_info;
int ipx_records_partition_slots(ipx_handle *h, uint32_t n_parts, const uint64_t *splitters, uint64_t slot_records,
                                uint64_t error_order, int32_t error_status, void *dst_slots);
int ipx_records_adopt_slots(ipx_handle *h, uint32_t n_slots, uint64_t slot_records, const void *slots,
                            uint64_t n_alignments, ipx_slots_info *info);
/* Overflow without a second scan (ABI 4).  ipx_records_partition_slots_spill is ipx_records_partition_slots that also
 * keeps what a slot cannot hold: the records of part p beyond its slot's room (slot_records - header) go to dst_spill
 * (device memory, spill_records records of 32 bytes; room for ipx_records_pending records is always enough), the tail of
 * part 0 first, then part 1's, ...  After ipx_records_adopt_slots said overflow, the true counts of every sender for
 * every part stand in the received headers (32-bit words from byte 32 of a slot on), so every rank knows how long each
 * tail is: a second all-to-all with those lengths brings the tails to their owners and
 *   ipx_records_adopt_more   appends n received records to the pending ones (after ipx_records_adopt_slots or
 *                            ipx_records_adopt; key masks and read counts go on from there)
 * -- the table is then the same as if the slots had been large enough.  (src/junctions_extractor.cc:466-485: every
 * record is counted once, whichever way it travelled.) */
int ipx_records_partition_slots_spill(ipx_handle *h, uint32_t n_parts, const uint64_t *splitters, uint64_t slot_records,
                                      uint64_t error_order, int32_t error_status, void *dst_slots, void *dst_spill,
                                      uint64_t spill_records);
int ipx_records_adopt_more(ipx_handle *h, uint64_t n, const void *records);
/* Run everything the handle queues on the caller's HIP stream (hipStream_t; NULL: back to the handle's own), e.g. the
 * stream a torch.distributed collective is ordered on.  Waits for what the handle had queued before.  NULL never means
 * the null stream: a caller that wants the legacy null stream names it, hipStreamLegacy ((hipStream_t)1) -- a handle
 * left on its own non-blocking stream runs BESIDE the null stream, ordered against nothing queued there. */
int ipx_use_stream(ipx_handle *h, void *hip_stream);

/* The handle's pending junction records, decoded, in no particular order (sort by `order`
 * for the order in which the reference would have met them).  For --debug-trace
 * (src/junctions_extractor.cc:451-457,471-477): the trace is the list of records that a
 * junction counted, i.e. those with IPX_RECORD_SEQ(order) >= the creator_order of their row
 * in the table.  dst is host memory with room for `capacity` records; *n_records receives
 * the number the handle holds (nothing is written if that exceeds capacity).  The records must not have been
 * combined: set IPX_OPT_COMBINE to 0 before the first push (IPX_EINVAL otherwise). */
typedef struct ipx_record {
    uint64_t order;          /* read_index << 30 | ordinal << 10 | read strand << 2 | read category */
    uint32_t target;         /* as in ipx_table.target */
    uint32_t intron_start, intron_end, anchor_start, anchor_end;
    uint32_t read_offset;    /* low 16 bits of intron_start - pos */
} ipx_record;
#define IPX_RECORD_SEQ(order) ((order) >> 10)            /* comparable with ipx_table.creator_order */
#define IPX_RECORD_READ(order) ((order) >> 30)
#define IPX_RECORD_STRAND(order) ((char)(((order) >> 2) & 0xff))
int ipx_records_export(ipx_handle *h, ipx_record *dst, uint64_t capacity, uint64_t *n_records);

/* --- several GPUs of one node behind one object (SURVEY.md 8b "device list", 8e) ---------------------
 * An ipx_group stands where ONE JunctionsExtractor stands (junctions_extractor.hh:73-256) and spreads its
 * alignments over n_devices GPUs of the calling process: every pushed batch is cut into n_devices contiguous
 * ranges of the read index (base_read_index keeps the order global), each device scans its range; finalize
 * exchanges the junction records by key range -- one grouped ncclSend/ncclRecv all-to-all over RCCL/xGMI,
 * librccl loaded at run time -- so that every junction's records meet on one device, reduces there with the
 * single-GPU semantics, and returns ONE table in host memory: rows in key order, ijunc the creation rank over
 * the whole run, the earliest data error of all devices reported as ipx_finalize would.  A device ordinal
 * listed twice makes two members share that GPU (rehearsal on a one-GPU box): the exchange then moves the
 * parts with peer copies instead of RCCL.  devices == NULL means 0 .. n_devices-1. */
typedef struct ipx_group ipx_group;
int ipx_group_create(const ipx_params *params, uint32_t n_devices, const int32_t *devices, ipx_group **out);
int ipx_group_destroy(ipx_group *g);
int ipx_group_set_targets(ipx_group *g, uint32_t n, const char *const *names, const int64_t *lengths);
int ipx_group_set_genome_2bit(ipx_group *g, uint32_t n_seqs, const char *const *names, const int64_t *lengths,
                              const uint8_t *const *twobit, const uint8_t *const *mask);
int ipx_group_set_option(ipx_group *g, int32_t option, int64_t value);
/* batch in host memory (IPX_MEM_HOST), CIGAR words laid out in alignment order */
int ipx_group_push_batch(ipx_group *g, const ipx_batch *batch);
/* one member's share, sharded by the caller; the batch may be in host memory or in THAT member's device memory
 * (IPX_MEM_DEVICE): the group's steps can then be timed on resident input like a single handle's */
int ipx_group_push_member(ipx_group *g, uint32_t member, const ipx_batch *batch);
/* table in host memory, owned by the group, valid until the next finalize / clear / destroy */
int ipx_group_finalize(ipx_group *g, ipx_table *out);
int ipx_group_clear(ipx_group *g);
int ipx_group_last_error(const ipx_group *g, ipx_error *out);
/* 1 if the group's exchange runs over RCCL, 0 if over peer copies (members sharing a device) */
int ipx_group_uses_rccl(const ipx_group *g);

/* Global sjN across key ranges (ABI 5).  The reference numbers junctions in the order it creates them
 * (junctions_extractor.hh:129,202; junctions_extractor.cc:459-460); after a record exchange every rank (or group member)
 * holds a key range, and a junction's number is the rank of its creation order among ALL ranges' junctions.
 * ipx_table_sorted_orders writes the creation orders of `table` (a device table of this handle's last ipx_finalize) in
 * ascending order to out[0 .. capacity) in device memory -- the table's own ijunc is that permutation -- and fills the rest
 * with UINT64_MAX: lists of one agreed length can be all-gathered without exchanging lengths.
 * ipx_table_global_ijunc takes n_lists such lists of `length` words each (device memory, `stride` words apart) and
 * writes, for every junction of `table`, the number of smaller orders in all of them: its global sjN.  Both are queued on the handle's stream and
 * do not wait.  (Rounds 2-4 did this in the Python harness with torch index, searchsorted and sum ops.) */
int ipx_table_sorted_orders(ipx_handle *h, const ipx_table *table, uint64_t *out, uint64_t capacity);
int ipx_table_global_ijunc(ipx_handle *h, const ipx_table *table, const uint64_t *lists, uint32_t n_lists,
                           uint64_t stride, uint64_t length, uint32_t *out_ijunc);

/* copy between any two of {host, device} on the handle's stream, then wait */
int ipx_copy(ipx_handle *h, void *dst, const void *src, uint64_t bytes);
/* the same without the wait (stream-ordered; the source must stay valid until the stream gets there) */
int ipx_copy_async(ipx_handle *h, void *dst, const void *src, uint64_t bytes);

/* device timing of the last ipx_push_batch / ipx_finalize in milliseconds,
 * measured with HIP events on the handle's stream */
int ipx_last_timing(const ipx_handle *h, float *push_ms, float *finalize_ms);

/* Per-kernel device time, measured with HIP event pairs on the handle's stream
 * around every launch while profiling is enabled (bench.py's roofline figures). */
typedef struct ipx_kernel_stat {
    char     name[32];
    uint64_t launches;
    double   total_ms;
} ipx_kernel_stat;
int ipx_profile(ipx_handle *h, int enable);     /* enable/disable, resets the sums */
/* Time one kernel class alone ("k_scan", "k_scatter" or "k_reduce"): an event right before and right
 * after each of its launches and nothing else -- event pairs around EVERY launch (ipx_profile) cost a
 * 10 M-alignment step about 10 % of its time, these do not. */
int ipx_profile_only(ipx_handle *h, const char *kernel);
int ipx_profile_read(const ipx_handle *h, ipx_kernel_stat *out, uint32_t cap, uint32_t *n);

#ifdef __cplusplus
}
#endif
#endif
