// junctions.cc -- ordering and text formats of junction calls (see junctions.hh).
#include "junctions.hh"

#include <algorithm>
#include <cmath>

namespace ipxhost {

// src/junctions.cc:54-70.  The reference sorts the uint16 offsets and walks runs of
// equal values in ascending order; the device hands over exactly those runs.  Float
// arithmetic, one rounding per operation (this file is built with -ffp-contract=off
// and without -march flags, like the reference's plain -O2 build).
float confidence_from_histogram(const uint32_t *counts, uint64_t n_bins, uint64_t n_reads) {
    if (n_reads == 0) return 0.0f;     // lazy_get_confidence, junctions.cc:44-51
    float summation = 0.0f;
    const float n = (float)n_reads;    // float(cnt) / read_offsets.size(): size_t -> float
    for (uint64_t i = 0; i < n_bins; ++i) {
        const float p = (float)counts[i] / n;
        if (p != 0.0f) {
            const float t = p * log2f(p);
            summation += t;
        }
    }
    return (summation == 0.0f) ? 0.0f : -summation;
}

// std::string operator< / operator>
static bool junctions_lt(const Junction &j1, const Junction &j2) {
    if (j1.chrom < j2.chrom) return true;
    if (j1.chrom > j2.chrom) return false;
    if (j1.anchor_start < j2.anchor_start) return true;
    if (j1.anchor_start > j2.anchor_start) return false;
    if (j1.anchor_end < j2.anchor_end) return true;
    if (j1.anchor_end > j2.anchor_end) return false;
    return j1.strand < j2.strand;
}

static bool introns_lt(const Junction &j1, const Junction &j2) {
    if (j1.chrom < j2.chrom) return true;
    if (j1.chrom > j2.chrom) return false;
    if (j1.intron_start < j2.intron_start) return true;
    if (j1.intron_start > j2.intron_start) return false;
    if (j1.intron_end < j2.intron_end) return true;
    if (j1.intron_end > j2.intron_end) return false;
    return j1.strand < j2.strand;
}

// The reference sorts a vector of pointers with std::sort (unstable; ties under
// junctions_lt are possible).  Sorting an index permutation applies the identical
// sequence of comparisons and element moves, so ties land where the reference's do.
static void sort_like_reference(JunctionVector &v, bool (*lt)(const Junction &, const Junction &)) {
    std::vector<const Junction *> ptrs(v.size());
    for (size_t i = 0; i < v.size(); ++i) ptrs[i] = &v[i];
    std::sort(ptrs.begin(), ptrs.end(), [lt](const Junction *a, const Junction *b) { return lt(*a, *b); });
    JunctionVector sorted;
    sorted.reserve(v.size());
    for (const Junction *p : ptrs) sorted.push_back(*p);
    v.swap(sorted);
}

void JunctionVector::sort_by_anchors() { sort_like_reference(*this, junctions_lt); }
void JunctionVector::sort_by_introns() { sort_like_reference(*this, introns_lt); }

// src/junctions.cc:162-164
static unsigned read_count_to_bed_score(uint64_t read_count) {
    return (read_count <= 1000) ? (unsigned)read_count : 1000;
}

// src/junctions.cc:167-185
static const char *bed_color(const std::string &splice_sites) {
    if (splice_sites.empty()) return "64,128,64";
    if (splice_sites == "GT/AG") return "0,128,0";
    if (splice_sites == "AT/AC" || splice_sites == "AT/AG" || splice_sites == "GT/AC") return "0,0,128";
    return "128,0,0";
}

static void print_name(const Junction &j, std::ostream &out) {
    out << "sj" << j.ijunc;
    if (!j.splice_sites.empty()) out << '_' << j.splice_sites;
}

// src/junctions.cc:189-200 (uint32 arithmetic)
static void print_anchor_bed_row(const Junction &j, std::ostream &out) {
    out << j.chrom << "\t" << j.anchor_start << "\t" << j.anchor_end << "\t";
    print_name(j, out);
    out << "\t" << read_count_to_bed_score(j.total_read_count()) << "\t" << j.strand << "\t" << j.anchor_start
        << "\t" << j.anchor_end << "\t" << bed_color(j.splice_sites) << "\t" << 2 << "\t"
        << (uint32_t)(j.intron_start - j.anchor_start) << "," << (uint32_t)(j.anchor_end - j.intron_end) << "\t"
        << "0," << (uint32_t)(j.intron_end - j.anchor_start) << '\n';
}

// src/junctions.cc:204-217
static void print_intron_bed_row(const Junction &j, int columns, std::ostream &out) {
    out << j.chrom << "\t" << j.intron_start << "\t" << j.intron_end << "\t";
    print_name(j, out);
    out << "\t" << read_count_to_bed_score(j.total_read_count()) << "\t" << j.strand;
    if (columns == 9) out << "\t" << j.intron_start << "\t" << j.intron_end << "\t" << bed_color(j.splice_sites);
    out << '\n';
}

// src/junctions.cc:220-225
static void print_call_row(const Junction &j, std::ostream &out) {
    out << j.chrom << "\t" << j.intron_start << "\t" << j.intron_end << "\t" << j.strand << "\t"
        << j.read_counts[SINGLE_MAPPED_READ] << "\t" << j.read_counts[MULTI_MAPPED_READ] << "\t"
        << j.read_counts[UNSURE_READ] << "\t" << (uint32_t)(j.intron_start - j.anchor_start) << "\t"
        << (uint32_t)(j.anchor_end - j.intron_end) << "\t" << j.confidence << "\t" << j.splice_sites << "\tsj"
        << j.ijunc << '\n';
}

// The reference ends every row with std::endl, i.e. a flush and a write() per row (src/junctions.cc:199,216,224): 200 000
// system calls per file, a tenth of a second of the command line once the device does the rest.  The bytes are the same
// with '\n' and ONE flush when a target's rows are out -- what a reader of the file can tell apart is only how much of an
// unfinished target's rows a crash leaves behind.
void print_anchor_bed(const JunctionVector &juncs, float min_confidence_score, std::ostream &out) {
    for (const Junction &j : juncs)
        if (j.confidence >= min_confidence_score) print_anchor_bed_row(j, out);
    out.flush();
}

void print_intron_bed(const JunctionVector &juncs, float min_confidence_score, int columns, std::ostream &out) {
    for (const Junction &j : juncs)
        if (j.confidence >= min_confidence_score) print_intron_bed_row(j, columns, out);
    out.flush();
}

void print_junction_call_header(std::ostream &out) {
    out << "chrom\tintron_start\tintron_end\tstrand\tuniq_mapped_count\tmulti_mapped_count\t"
           "unsure_mapped_count\tmax_left_overhang\tmax_right_overhang\tconfidence\tsplice_sites\tname"
        << std::endl;
}

void print_intron_call_tsv(const JunctionVector &juncs, float min_confidence_score, std::ostream &out) {
    out.precision(3);
    for (const Junction &j : juncs)
        if (j.confidence >= min_confidence_score) print_call_row(j, out);
    out.flush();
}

}  // namespace ipxhost
