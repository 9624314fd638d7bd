// junctions_extractor.hh -- host mirror of the reference's JunctionsExtractor
// (src/junctions_extractor.hh:73-256) on top of the C ABI in include/ipx.h.
//
// Same public surface and call order as the reference's only caller uses
// (src/intronProspector.cc:356-429): construct with the thresholds, open(),
// identify_junctions_for_target() / identify_junctions_for_bam(), get_junctions(),
// clear(), close().  The per-alignment work happens on the GPU; this class keeps
// what stays on the host: the read loops with their one-record look-ahead and
// sortedness check (junctions_extractor.cc:620-683), the target-name vector that
// grows with every open() (:564-568), the running ijunc_ (:129,459-460), the
// missing-sequence warnings (:411-416) and the translation of device error
// reports into the reference's exceptions, and the --debug-trace stream
// (:451-457,471-477,579-584) and the per-read orientation count behind
// --set-XS/TS-strand-tag (:526-528,540-547), both rebuilt from the device's record list,
// and the pass-through output (:589-602,637-652,667-674; host/pass_through.hh).
#ifndef IPX_HOST_JUNCTIONS_EXTRACTOR_HH
#define IPX_HOST_JUNCTIONS_EXTRACTOR_HH
#include <cstdint>
#include <iostream>
#include <atomic>
#include <cstdio>
#include <thread>
#include <memory>
#include <set>
#include <string>
#include <thread>
#include <exception>
#include <vector>

#include "../../include/ipx.h"
#include "aln_reader.hh"
#include "bam_feeder.hh"
#include "genome.hh"
#include "junctions.hh"
#include "pass_through.hh"

namespace ipxhost {

// src/splice_juncs.hh:12-16
enum JunctionFilter { NULL_SJ_FILTER = 0, ALL_SJ_FILTER = 1, CANON_SJ_FILTER = 2 };
// src/junctions_extractor.hh:56-60
enum Strandness { UNSTRANDED = 0, RF_STRANDED = 1, FR_STRANDED = 2 };
// src/junctions_extractor.hh:64-67
enum ExcludeCats { EXCLUDE_NONE = 0x00, EXCLUDE_MULTI = 0x01 };

JunctionFilter junction_filter_parse(const std::string &spec);   // src/splice_juncs.cc:41-49

class JunctionsExtractor {
public:
    JunctionsExtractor(uint32_t min_anchor_length, uint32_t max_anchor_indel_size, uint32_t min_intron_length,
                       uint32_t max_intron_length, JunctionFilter junction_filter, Strandness strandness,
                       unsigned excludes, Genome *genome, bool skip_missing_targets, int device = 0,
                       std::ostream *warn = &std::cerr, std::ostream *trace_fh = nullptr,
                       bool set_XS_strand_tag = false, bool set_TS_strand_tag = false,
                       const std::vector<int> &devices = std::vector<int>());
    ~JunctionsExtractor();
    JunctionsExtractor(const JunctionsExtractor &) = delete;
    JunctionsExtractor &operator=(const JunctionsExtractor &) = delete;

    void open(const std::string &bam);
    void open_pass_through(const std::string &bam_pass_through);
    void copy_unaligned_reads();
    void close();
    int get_num_targets() const { return (int)targets_.size(); }
    void identify_junctions_for_target(int target_index);
    void identify_junctions_for_bam();
    JunctionVector get_junctions();
    void clear();

    // alignments handed to the device so far (statistics)
    uint64_t alignments_processed() const { return next_read_index_; }

private:
    // one GPU (h_) or several behind one ipx_group (g_, --devices): the alignments of every batch are
    // spread over the group's GPUs by read index, the junction records meet by key range over RCCL
    ipx_handle *h_ = nullptr;
    ipx_group *g_ = nullptr;
    // The device is opened on a thread of its own while the caller already opens the input (its BGZF blocks
    // inflate in the meantime): HIP start-up, handle, genome upload take 0.2 s.  Everything that touches
    // h_ / g_ goes through ensure_device() first, which also hands over the target list open() collected.
    ipx_params params_{};
    std::vector<int> devices_;
    std::thread init_thread_;
    std::exception_ptr init_error_;
    bool targets_dirty_ = false;
    void init_device();
    void ensure_device();
    Genome *genome_;
    bool skip_missing_targets_;
    std::ostream *warn_;
    std::string bam_;
    std::vector<std::string> targets_;        // grows with every open()
    std::vector<int32_t> canon_tid_;          // first index carrying the same name
    std::set<std::string> missing_genomic_warned_;
    int previous_target_ = -1;
    bool aln_pending_ = false;
    AlnRecord aln_buf_;
    std::unique_ptr<AlnReader> reader_;
    uint32_t ijunc_ = 0;
    uint64_t next_read_index_ = 0;
    // The host batch (SoA of include/ipx.h) lives in PINNED memory, twice: the read loop fills one set of
    // arrays while the device copies and scans the other (ipx_push_batch_async); a full set is handed over
    // and the loop goes on with the set whose push has completed by then.  Sets start small (tests read a few
    // hundred alignments) and grow to BATCH_ALIGNMENTS.
    struct HostBatch {
        void *mem = nullptr;                      // one ipx_host_alloc block holding all eight arrays
        size_t cap_aln = 0, cap_words = 0, n = 0, nw = 0;
        int32_t *tid = nullptr, *pos = nullptr;
        uint16_t *flag = nullptr;
        uint8_t *xs = nullptr, *nh = nullptr;
        uint32_t *off = nullptr, *nc = nullptr, *cigar = nullptr;
        // device parse (ipx_push_bam): the records as they lie in the inflated BGZF blocks, and where runs of
        // whole records begin; the tables stay untouched until the push that borrowed them has completed
        void *raw = nullptr;
        size_t raw_cap = 0, raw_n = 0;
        std::vector<uint64_t> span_begin, span_first;
        std::vector<int32_t> tid_map;
        bool tables_lent = false;
    };
    HostBatch batch_[2];
    int cur_ = 0;
    // BAM input with nothing that needs the records one by one (pass-through, trace): blocks are split on this
    // thread and parsed straight into the batches by a pool (host/bam_feeder.hh); IPX_PARSE_THREADS=0 keeps
    // the record-by-record loop
    std::unique_ptr<BamFeeder> feeder_;
    // ... or, on one GPU, only split here and decoded on the device (ipx_push_bam): this thread then copies
    // whole runs of records into pinned memory and never looks inside them.  IPX_DEVICE_PARSE=0 switches it off
    bool device_parse_ = false;
    // ... or the whole file goes to the device as it is (ipx_decode_bgzf): blocks inflated and CRC-checked there,
    // records found and decoded there; this thread walks the block headers, and afterwards the refIDs (the reader's
    // target-by-target state machine runs over them) to push each target's slice of the device-resident batch.
    // IPX_DEVICE_INFLATE=1 asks for it, 0 forbids it; unasked it takes BGZF files of 16 MB and more.  A file the
    // device refuses (damage, records that straddle block ends) is read the ordinary way, which reports what is wrong.
    bool device_file_ = false, device_file_tried_ = false;
    ipx_batch dev_batch_{};
    std::vector<int32_t> dev_tid_;
    uint64_t dev_cursor_ = 0;
    bool try_device_file();
    bool feed_target_device(int32_t target_index, int32_t *other_tid);
    // the file's bytes, read by a few threads from open() on -- while the device is still being opened
    struct FileLoad {
        uint8_t *buf = nullptr;
        size_t size = 0;
        std::vector<std::thread> readers;
        std::atomic<bool> failed{false};
        void start(const std::string &path, size_t n_bytes, unsigned n_threads);
        bool wait();                       // true: buf holds the whole file
        void drop();
        ~FileLoad() { drop(); }
    } file_load_;
    void size_raw(HostBatch &b, size_t cap_bytes);
    bool use_feeder();
    // records of target_index (BamFeeder::ANY_TARGET: all) from the feeder into the batches;
    // returns false at the end of the input, true at a record of another target (*other_tid)
    bool feed_target(int32_t target_index, int32_t *other_tid);
    void size_batch(HostBatch &b, size_t cap_aln, size_t cap_words);
    // IPX_TIMELINE=1: where the wall time of the command goes (stderr, one JSON line at destruction)
    bool timeline_ = false;
    double t_read_ = 0, t_queue_ = 0, t_push_ = 0, t_finalize_ = 0, t_table_ = 0, t_create_ = 0, t_alloc_ = 0, t_born_ = 0, t_identify_ = 0, t_close_ = 0, t_getj_ = 0, t_open_ = 0;
    uint64_t n_push_ = 0, n_device_files_ = 0;
    uint32_t tick_ = 0;
    // finalized table of the records pushed since the last clear()
    bool table_valid_ = false;
    JunctionVector table_;
    // --debug-trace: names and flags of the alignments queued since the last clear()
    std::ostream *trace_fh_;
    uint64_t trace_base_ = 0;                 // read index of trace_qname_[0]
    uint64_t trace_done_ = 0;                 // records with a smaller read index are already written
    std::vector<std::string> trace_qname_;
    std::vector<uint16_t> trace_flag_;
    void write_trace(const ipx_table &t, const std::vector<ipx_record> &recs);
    // --pass-through: alignments of the current target wait here for their orientation count when a
    // strand tag is to be set (the count needs the target's table); otherwise they are written at once
    bool set_xs_, set_ts_;
    std::unique_ptr<PassThroughWriter> pass_through_;
    // Alignments that wait for their target's table before they can be written with a strand tag.  The reference
    // streams (junctions_extractor.cc:647-650); here a read's orientation count comes from the finished table, so the
    // raw records wait -- in memory up to IPX_PASS_THROUGH_BUFFER_MB (default 256), beyond that in an unlinked
    // temporary file, read back once, in order.  What stays in memory per waiting alignment is one byte.
    struct PendingSpool {
        std::string buf;                    // [u32 length][u32 aux_off][raw bytes] ...
        FILE *spill = nullptr;
        uint64_t n = 0, spilled_bytes = 0, first_read_index = 0;
        size_t cap_bytes = (size_t)256 << 20;
        bool empty() const { return n == 0; }
        void add(const AlnRecord &rec, uint64_t read_index);
        // fn(raw, aux_off, k) for the k-th waiting alignment, in input order; empties the spool
        template <typename F> void drain(F fn);
        ~PendingSpool() { if (spill) fclose(spill); }
    } pt_pending_;
    void write_pending_pass_through(const ipx_table &t, const std::vector<ipx_record> &recs);
    std::vector<ipx_record> export_records();

    bool read_align();
    void process_alignment(const AlnRecord &aln);
    void flush_batch();
    void finalize_table();
    [[noreturn]] void throw_last_error(int status);
};

}  // namespace ipxhost
#endif
