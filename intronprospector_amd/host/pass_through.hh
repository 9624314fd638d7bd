// pass_through.hh -- the --pass-through output and the --set-XS/TS-strand-tag rewrite.
//
// Stands where hts_open_format / sam_hdr_write / sam_write1 and bam_aux_get / bam_aux_del /
// bam_aux_append stand in the reference (src/junctions_extractor.cc:98-111,552-560,589-602):
// every alignment is written out in the input's format (SAM text or BAM), header first; a read
// whose canonical junctions lean to one strand (orientCnt != 0, :526-528,540-547) gets its XS
// and/or TS tag deleted where it was and appended, as type A, at the end of the record.
// Records are not re-encoded: the SAM line / BAM record is kept as read and only the tag bytes
// change, which for files htslib round-trips unchanged (tests/Makefile.in:123,136 diff the
// pass-through against the input) is the same output.
#ifndef IPX_HOST_PASS_THROUGH_HH
#define IPX_HOST_PASS_THROUGH_HH
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "aln_reader.hh"

namespace ipxhost {

// tag edits on a kept record (exposed for the CPU tests through ipx_host_check)
void sam_set_strand_tag(std::string &line, const char *tag, char strand);
void bam_set_strand_tag(std::string &record, uint32_t aux_off, const char *tag, char strand);

class PassThroughWriter {
public:
    // same container as the input: BAM (BGZF) if the reader decodes BAM, else SAM text
    PassThroughWriter(const std::string &path, const AlnReader &reader);
    ~PassThroughWriter();
    PassThroughWriter(const PassThroughWriter &) = delete;
    PassThroughWriter &operator=(const PassThroughWriter &) = delete;

    // orient_cnt != 0 and set_xs / set_ts: update_strand_tag (junctions_extractor.cc:98-111)
    void write(const AlnRecord &rec, int orient_cnt, bool set_xs, bool set_ts);
    void close();

private:
    std::string path_;
    FILE *fp_ = nullptr;
    bool bam_ = false;
    std::vector<uint8_t> block_;          // BGZF: uncompressed bytes waiting for a block
    void bgzf_append(const void *data, size_t n);
    void bgzf_flush_block(size_t n);
};

}  // namespace ipxhost
#endif
