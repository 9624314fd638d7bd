// junctions.cc -- ordering and text formats of junction calls (see junctions.hh).
#include "junctions.hh"

#include <algorithm>
#include <charconv>
#include <cmath>
#include <cstdio>
#include <map>

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

// The orders of src/junctions.cc:108-159 (std::string operator< / operator> on the chromosome first).  The chromosome
// is compared through its RANK among the table's distinct names: the same answers as comparing the strings, at the price
// of comparing two integers (a table of 200 000 rows on a handful of chromosomes spent most of its sort in memcmp).
struct SortKey {
    uint32_t chrom_rank, a, b;
    char strand;
};
static bool key_lt(const SortKey &x, const SortKey &y) {
    if (x.chrom_rank < y.chrom_rank) return true;
    if (x.chrom_rank > y.chrom_rank) return false;
    if (x.a < y.a) return true;
    if (x.a > y.a) return false;
    if (x.b < y.b) return true;
    if (x.b > y.b) return false;
    return x.strand < y.strand;
}

// The reference sorts a vector of pointers with std::sort (unstable; ties under
// junctions_lt are possible).  Sorting an index permutation applies the identical
// sequence of comparisons and element moves, so ties land where the reference's do.
static void sort_like_reference(JunctionVector &v, bool by_anchors) {
    std::map<std::string, uint32_t> rank_of;
    for (const Junction &j : v) rank_of.emplace(j.chrom, 0u);
    uint32_t next = 0;
    for (auto &kv : rank_of) kv.second = next++;             // std::map iterates in std::string order
    std::vector<SortKey> key(v.size());
    const std::string *last_name = nullptr;
    uint32_t last_rank = 0;
    for (size_t i = 0; i < v.size(); ++i) {
        const Junction &j = v[i];
        if (!last_name || *last_name != j.chrom) { last_name = &j.chrom; last_rank = rank_of.find(j.chrom)->second; }
        key[i] = by_anchors ? SortKey{last_rank, j.anchor_start, j.anchor_end, j.strand}
                            : SortKey{last_rank, j.intron_start, j.intron_end, j.strand};
    }
    std::vector<uint32_t> idx(v.size());
    for (size_t i = 0; i < v.size(); ++i) idx[i] = (uint32_t)i;
    std::sort(idx.begin(), idx.end(), [&key](uint32_t x, uint32_t y) { return key_lt(key[x], key[y]); });
    JunctionVector sorted;
    sorted.reserve(v.size());
    for (uint32_t i : idx) sorted.push_back(std::move(v[i]));
    v.swap(sorted);
}

void JunctionVector::sort_by_anchors() { sort_like_reference(*this, true); }
void JunctionVector::sort_by_introns() { sort_like_reference(*this, false); }

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

// Rows are formatted into a buffer of a megabyte and written in one piece: the reference streams every field through
// operator<< and ends every row with std::endl, a flush and a write() per row (src/junctions.cc:199,216,224) -- 200 000
// system calls per file and most of a tenth of a second of formatting, a sixth of the command line once the device does the
// rest.  The BYTES are the same: integers by std::to_chars, the confidence as "%.3g" -- what operator<<(float) prints
// with the precision of 3 that print_intron_call_tsv sets and the default float field -- one flush when a target's rows
// are out; what a reader of the file can tell apart is only how much of an unfinished target's rows a crash leaves behind.
namespace {
struct RowBuf {
    std::string s;
    std::ostream &out;
    explicit RowBuf(std::ostream &o) : out(o) { s.reserve((size_t)1 << 20); }
    RowBuf &str(const std::string &x) { s.append(x); return *this; }
    RowBuf &lit(const char *x) { s.append(x); return *this; }
    RowBuf &ch(char c) { s.push_back(c); return *this; }
    RowBuf &tab() { s.push_back('\t'); return *this; }
    RowBuf &u(uint64_t v) {
        char b[24];
        const auto r = std::to_chars(b, b + sizeof b, v);
        s.append(b, r.ptr);
        return *this;
    }
    RowBuf &g3(float f) {
        char b[48];
        const int n = snprintf(b, sizeof b, "%.3g", (double)f);
        s.append(b, (size_t)(n > 0 ? n : 0));
        return *this;
    }
    void end_row() {
        s.push_back('\n');
        if (s.size() + 1024 > ((size_t)1 << 20)) write();
    }
    void write() {
        out.write(s.data(), (std::streamsize)s.size());
        s.clear();
    }
    ~RowBuf() {
        write();
        out.flush();
    }
};

void put_name(const Junction &j, RowBuf &o) {
    o.lit("sj").u(j.ijunc);
    if (!j.splice_sites.empty()) o.ch('_').str(j.splice_sites);
}
}  // namespace

// src/junctions.cc:189-200 (uint32 arithmetic)
static void print_anchor_bed_row(const Junction &j, RowBuf &o) {
    o.str(j.chrom).tab().u(j.anchor_start).tab().u(j.anchor_end).tab();
    put_name(j, o);
    o.tab().u(read_count_to_bed_score(j.total_read_count())).tab().ch(j.strand).tab().u(j.anchor_start).tab().u(j.anchor_end)
        .tab().lit(bed_color(j.splice_sites)).tab().u(2).tab().u((uint32_t)(j.intron_start - j.anchor_start)).ch(',')
        .u((uint32_t)(j.anchor_end - j.intron_end)).tab().lit("0,").u((uint32_t)(j.intron_end - j.anchor_start));
    o.end_row();
}

// src/junctions.cc:204-217
static void print_intron_bed_row(const Junction &j, int columns, RowBuf &o) {
    o.str(j.chrom).tab().u(j.intron_start).tab().u(j.intron_end).tab();
    put_name(j, o);
    o.tab().u(read_count_to_bed_score(j.total_read_count())).tab().ch(j.strand);
    if (columns == 9) o.tab().u(j.intron_start).tab().u(j.intron_end).tab().lit(bed_color(j.splice_sites));
    o.end_row();
}

// src/junctions.cc:220-225
static void print_call_row(const Junction &j, RowBuf &o) {
    o.str(j.chrom).tab().u(j.intron_start).tab().u(j.intron_end).tab().ch(j.strand).tab().u(j.read_counts[SINGLE_MAPPED_READ]).tab()
        .u(j.read_counts[MULTI_MAPPED_READ]).tab().u(j.read_counts[UNSURE_READ]).tab().u((uint32_t)(j.intron_start - j.anchor_start))
        .tab().u((uint32_t)(j.anchor_end - j.intron_end)).tab().g3(j.confidence).tab().str(j.splice_sites).lit("\tsj").u(j.ijunc);
    o.end_row();
}

void print_anchor_bed(const JunctionVector &juncs, float min_confidence_score, std::ostream &out) {
    RowBuf o(out);
    for (const Junction &j : juncs)
        if (j.confidence >= min_confidence_score) print_anchor_bed_row(j, o);
}

void print_intron_bed(const JunctionVector &juncs, float min_confidence_score, int columns, std::ostream &out) {
    RowBuf o(out);
    for (const Junction &j : juncs)
        if (j.confidence >= min_confidence_score) print_intron_bed_row(j, columns, o);
}

void print_junction_call_header(std::ostream &out) {
    out << "chrom\tintron_start\tintron_end\tstrand\tuniq_mapped_count\tmulti_mapped_count\t"
           "unsure_mapped_count\tmax_left_overhang\tmax_right_overhang\tconfidence\tsplice_sites\tname"
        << std::endl;
}

void print_intron_call_tsv(const JunctionVector &juncs, float min_confidence_score, std::ostream &out) {
    out.precision(3);
    RowBuf o(out);
    for (const Junction &j : juncs)
        if (j.confidence >= min_confidence_score) print_call_row(j, o);
}

}  // namespace ipxhost
