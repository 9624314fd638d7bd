// bgzf_reader.hh -- BGZF (blocked gzip, SAM spec 4.1) input with the blocks inflated by a pool of
// threads, delivered in file order.
//
// Stands where htslib's bgzf_read (single-threaded unless hts_set_threads is called; the
// reference never calls it, src/junctions_extractor.cc:571-586) stands under sam_read1.  BGZF blocks
// are independent raw-deflate streams of <= 64 KiB, so a BAM file inflates in parallel: workers take
// the next block off the file under a lock (a sequential fread), inflate and CRC-check it outside the
// lock, and park the result in a ring slot; the consumer takes slots in order.  SURVEY.md 8(f)-1.
#ifndef IPX_HOST_BGZF_READER_HH
#define IPX_HOST_BGZF_READER_HH
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace ipxhost {

class BgzfReader {
public:
    // does the file start with a BGZF block header?
    static bool is_bgzf(const std::string &path);
    // n_threads 0: IPX_DECODE_THREADS from the environment, else default_threads()
    explicit BgzfReader(const std::string &path, unsigned n_threads = 0);
    // three quarters of the CPUs this process may use (cgroup quota respected), at most 12 (measured)
    static unsigned default_threads();
    // CPUs this process may use (hardware threads, cut to the cgroup's quota)
    static unsigned usable_cpus();
    ~BgzfReader();
    BgzfReader(const BgzfReader &) = delete;
    BgzfReader &operator=(const BgzfReader &) = delete;

    // up to n bytes; 0 at end of file; throws std::runtime_error on a damaged block
    size_t read(void *dst, size_t n);
    // Zero-copy access for records that lie inside one block (all but one in a few hundred): peek() hands out
    // a pointer to the next n bytes if the current block holds them all (false: use read(), the bytes straddle
    // blocks or the input has ended); the pointer stays valid until advance(), which consumes the bytes.
    bool peek(size_t n, const uint8_t **p);
    void advance(size_t n);
    // Whole blocks for a consumer that parses them elsewhere (host/bam_feeder.cc): the rest of the current block
    // (after read / advance consumed its head) or the whole next one is swapped into `out`, whose old storage the
    // ring reuses.  false at the end of the input.
    bool next_block(std::vector<uint8_t> &out);
    unsigned threads() const { return (unsigned)workers_.size(); }
    uint64_t compressed_bytes() const { return compressed_bytes_; }

private:
    struct Slot {
        std::vector<uint8_t> in, out;
        bool ready = false;
        std::string error;
    };
    std::string path_;
    FILE *fp_ = nullptr;
    std::vector<Slot> ring_;
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable can_produce_, can_consume_;
    uint64_t next_claim_ = 0;        // sequence number of the next block to read off the file
    uint64_t next_consume_ = 0;      // sequence number the consumer waits for
    uint64_t end_seq_ = UINT64_MAX;  // first sequence number past the last block, once known
    bool stop_ = false;
    uint64_t compressed_bytes_ = 0;
    // consumer side
    size_t cur_off_ = 0;
    bool cur_valid_ = false;

    void worker();
    // reads one whole block into s.in; false at a clean end of file; sets s.error on a damaged header
    bool read_block(Slot &s);
};

}  // namespace ipxhost
#endif
