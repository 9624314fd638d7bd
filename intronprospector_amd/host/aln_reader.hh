// aln_reader.hh -- alignment decoding: SAM text, gzip'd SAM, and BAM (BGZF).
//
// Stands where htslib's sam_open / sam_hdr_read / sam_read1 / bam_aux_get /
// bam_aux2i / bam_aux2A stand under the reference's read loops
// (src/junctions_extractor.cc:49-56,322-330,571-586,620-634).  htslib is not part
// of this image; the decoder is written against the SAM/BAM specification and
// yields only what the hot path consumes, already in the encoding of
// include/ipx.h (xs / nh classes, BAM-packed CIGAR words).
#ifndef IPX_HOST_ALN_READER_HH
#define IPX_HOST_ALN_READER_HH
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>
#include <memory>
#include <zlib.h>

#include "bgzf_reader.hh"

namespace ipxhost {

struct AlnRecord {
    int32_t tid = -1;
    int64_t pos = 0;
    uint16_t flag = 0;
    uint8_t xs = 0;        // 0 = no XS tag, else strand byte ('.' if the tag is not of type A)
    uint8_t nh = 0;        // 0 = no NH / <= 0, 1 = NH == 1, 2 = NH > 1
    // CIGAR words in BAM packing: n_cig words starting at cig (unaligned; valid until the next read()).
    // SAM text fills `cigar` and points cig at it; BAM points straight into the inflated block.
    const uint8_t *cig = nullptr;
    uint32_t n_cig = 0;
    uint32_t cigar_word(size_t k) const { uint32_t w; memcpy(&w, cig + 4 * k, 4); return w; }
    std::vector<uint32_t> cigar;
    std::string qname;
    // kept only when the reader was asked to (pass-through): the SAM line without its newline, or
    // the BAM record without its block_size word; aux_off = offset of the BAM record's tag section
    std::string raw;
    uint32_t aux_off = 0;
};

// What the hot path reads of one BAM record (the record body b[0 .. block), i.e. without its block_size word).
struct BamFields {
    int32_t tid = -1, pos = 0;
    uint16_t flag = 0;
    uint8_t xs = 0, nh = 0;
    const uint8_t *cig = nullptr;        // CIGAR words (the CG tag's for a long CIGAR), unaligned
    uint32_t n_cig = 0;
    uint32_t l_read_name = 0, aux_off = 0;
};
// throws std::runtime_error("Error reader BAM record: <path>") on a malformed record
void parse_bam_record(const uint8_t *b, uint32_t block, const std::string &path, BamFields &f);

class AlnReader {
public:
    explicit AlnReader(const std::string &path);     // throws std::runtime_error
    ~AlnReader();
    AlnReader(const AlnReader &) = delete;
    AlnReader &operator=(const AlnReader &) = delete;

    const std::vector<std::string> &target_names() const { return names_; }
    const std::vector<int64_t> &target_lengths() const { return lengths_; }
    bool is_bam() const { return bam_; }
    // pass-through needs the records and the header as they came
    void keep_raw(bool on) { keep_raw_ = on; }
    // read names are only wanted by --debug-trace; leaving them out saves a string assignment per record
    void keep_qname(bool on) { keep_qname_ = on; }
    // SAM: the header lines; BAM: everything between the magic and the first record (l_text ... refs)
    const std::string &raw_header() const { return raw_header_; }
    unsigned decode_threads() const { return bgzf_ ? bgzf_->threads() : 1; }
    // the BGZF block source of a BAM file (null for SAM text, pipes, plain-gzip BAM): host/bam_feeder.cc parses
    // whole blocks on several threads instead of calling read() per record
    BgzfReader *bgzf() { return bam_ ? bgzf_.get() : nullptr; }
    const std::string &path() const { return path_; }
    // false at end of input; throws std::runtime_error on a malformed record
    bool read(AlnRecord &rec);

private:
    std::string path_;
    gzFile fh_ = nullptr;                    // SAM text, gzip'd SAM
    std::unique_ptr<BgzfReader> bgzf_;       // BAM: blocks inflated by a thread pool
    bool bam_ = false;
    bool keep_raw_ = false;
    bool keep_qname_ = true;
    std::string raw_header_;
    std::vector<std::string> names_;
    std::vector<int64_t> lengths_;
    // SAM text state
    std::string line_, pushback_;            // pushback_: bytes read while probing the format
    bool have_line_ = false;
    bool read_line();
    void parse_sam_header();
    bool read_sam(AlnRecord &rec);
    // BAM state
    std::vector<uint8_t> buf_;
    size_t pending_advance_ = 0;             // bytes of the block the previous record was parsed from in place
    size_t read_some(void *dst, size_t n);
    void read_exact(void *dst, size_t n, const char *what);
    void parse_bam_header();
    bool read_bam(AlnRecord &rec);
};

}  // namespace ipxhost
#endif
