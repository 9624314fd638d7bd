// bam_feeder.hh -- BAM records parsed block-parallel into the SoA batch of include/ipx.h.
//
// Stands where the per-record sam_read1 loop stands (src/junctions_extractor.cc:620-664) when nothing needs the
// records one at a time (no pass-through, no --debug-trace).  One thread parsing record after record is the
// limit of the whole command line once the BGZF blocks inflate on a thread pool (a 10 M-record BAM: 0.9 s of
// parsing + queueing against 0.5 s of inflating on twelve threads and 2 ms on the GPU).  So the caller's thread
// only SPLITS: it walks the record lengths of each inflated block (block_size, refID and n_cigar_op of every
// record: a dozen bytes per record), cuts runs of records that belong to the wanted target and fit the batch,
// and hands each run -- a byte range of a block, or one record assembled from the blocks it straddles -- to a
// pool of workers that parse the records (parse_bam_record, the same code AlnReader::read uses) and write
// them to their places in the batch: record index and CIGAR offset of a run are known when it is cut.
#ifndef IPX_HOST_BAM_FEEDER_HH
#define IPX_HOST_BAM_FEEDER_HH
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <exception>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "bgzf_reader.hh"

namespace ipxhost {

class BamFeeder {
public:
    static const int32_t ANY_TARGET = INT32_MIN;
    static const unsigned SPLIT_ONLY = ~0u;        // n_workers: no parsers, the caller takes the spans as they are
    struct Dest {                                  // the batch the records go to
        int32_t *tid, *pos;
        uint16_t *flag;
        uint8_t *xs, *nh;
        uint32_t *off, *nc, *cigar;
    };
    struct Span {                                  // a run of whole records in one buffer
        std::shared_ptr<std::vector<uint8_t>> data;
        size_t begin = 0, end = 0;
        uint64_t n_records = 0, n_words = 0;
    };
    enum Status { SPAN, BOUNDARY, FULL, END };

    // src: positioned behind the BAM header; canon_tid: refID -> the tid the device is given (first index of
    // the chromosome name); n_workers 0: IPX_PARSE_THREADS or a quarter of the usable CPUs (1..4)
    BamFeeder(BgzfReader &src, const std::string &path, const std::vector<int32_t> &canon_tid, unsigned n_workers = 0);
    ~BamFeeder();
    BamFeeder(const BamFeeder &) = delete;
    BamFeeder &operator=(const BamFeeder &) = delete;

    // The next run of records with refID == want_tid (ANY_TARGET: whatever comes) of at most max_records records
    // and max_words CIGAR words.  SPAN: sp holds it; BOUNDARY: the next record has another refID (*other_tid),
    // nothing was taken; FULL: the next record does not fit the limits; END: no more records.
    Status next_span(int32_t want_tid, uint64_t max_records, uint64_t max_words, Span &sp, int32_t *other_tid);
    // parse sp's records into dst: records [rec_base, rec_base + n_records), CIGAR words from word_base on
    void dispatch(Span &&sp, const Dest &dst, uint64_t rec_base, uint64_t word_base);
    // until every dispatched span is written; rethrows what a worker threw
    void wait_all();
    unsigned workers() const { return (unsigned)threads_.size(); }

private:
    struct Task {
        Span sp;
        Dest dst;
        uint64_t rec_base, word_base;
    };
    BgzfReader &src_;
    std::string path_;
    const std::vector<int32_t> &canon_tid_;
    // splitter state (caller's thread)
    std::shared_ptr<std::vector<uint8_t>> blk_, strad_;
    size_t off_ = 0;
    bool eof_ = false;
    bool fetch_block();
    // inflated blocks are recycled: a fresh 64 KiB vector per block (20 000 of them for a 1 GB BAM, allocated on
    // one thread, filled on a second, freed on a third) cost more system time than the parsing itself
    std::mutex pool_m_;
    std::vector<std::vector<uint8_t> *> pool_;
    std::shared_ptr<std::vector<uint8_t>> new_buffer();
    uint32_t words_of(const uint8_t *rec, uint32_t block) const;
    // workers
    std::vector<std::thread> threads_;
    std::mutex m_;
    std::condition_variable have_task_, all_done_;
    std::deque<Task> queue_;
    uint64_t outstanding_ = 0;
    bool stop_ = false;
    std::exception_ptr error_;
    void worker();
    void run(const Task &t);
    [[noreturn]] void bad() const;
};

}  // namespace ipxhost
#endif
