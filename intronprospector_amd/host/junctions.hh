// junctions.hh -- host data model of a junction call and the output writers.
//
// Host side of the drop-in: what the reference keeps in class Junction /
// JunctionVector (src/junctions.hh:103-222) and prints with
// print_anchor_bed / print_intron_bed / print_junction_call_header /
// print_intron_call_tsv (src/junctions.hh:249-265, src/junctions.cc:161-268).
// The aggregate fields come from the device table (include/ipx.h); this file
// only orders and formats them, and computes the Shannon confidence from the
// offset histogram with the reference's float operation order.
#ifndef IPX_HOST_JUNCTIONS_HH
#define IPX_HOST_JUNCTIONS_HH
#include <cstdint>
#include <ostream>
#include <string>
#include <vector>

namespace ipxhost {

// read categories, src/junctions.hh:49-53
enum ReadCategory { SINGLE_MAPPED_READ = 0, MULTI_MAPPED_READ = 1, UNSURE_READ = 2 };

struct Junction {
    std::string chrom;
    uint32_t intron_start = 0, intron_end = 0;
    uint32_t anchor_start = 0, anchor_end = 0;
    char strand = '.';
    std::string splice_sites;          // "" or e.g. "GT/AG"; lower case when not canonical
    uint32_t ijunc = 0;
    uint64_t read_counts[3] = {0, 0, 0};
    float confidence = 0.0f;           // Junction::get_confidence(), src/junctions.cc:44-70

    uint64_t total_read_count() const { return read_counts[0] + read_counts[1] + read_counts[2]; }
};

// Shannon-Wiener index over the read offsets (src/junctions.cc:54-70), from the
// histogram of distinct offsets in ascending order: counts[0..n_bins), total reads n.
float confidence_from_histogram(const uint32_t *counts, uint64_t n_bins, uint64_t n_reads);

class JunctionVector : public std::vector<Junction> {
public:
    void sort_by_anchors();            // src/junctions.hh:218-221 with junctions_lt, junctions.cc:108-132
    void sort_by_introns();            // src/junctions.hh:213-216 with introns_lt, junctions.cc:135-159
};

void print_anchor_bed(const JunctionVector &juncs, float min_confidence_score, std::ostream &out);
void print_intron_bed(const JunctionVector &juncs, float min_confidence_score, int columns, std::ostream &out);
void print_junction_call_header(std::ostream &out);
void print_intron_call_tsv(const JunctionVector &juncs, float min_confidence_score, std::ostream &out);

}  // namespace ipxhost
#endif
