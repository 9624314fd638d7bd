/* ipxo_sort.cc -- oracle: output orderings.
 *
 * TEST INFRASTRUCTURE ONLY (see ipxo.h).  Restates the comparators
 * JunctionVector::junctions_lt / introns_lt (src/junctions.cc:108-159) and
 * applies them with std::sort exactly as JunctionVector::sort_by_anchors /
 * sort_by_introns do (src/junctions.hh:213-221).  std::sort is unstable and
 * ties under junctions_lt are possible, so this one file is C++ in order to
 * use the same libstdc++ introsort the reference is built with.
 */
#include "ipxo.h"
#include <algorithm>
#include <cstring>

static int name_cmp(const char *a, const char *b) {
    size_t la = strlen(a), lb = strlen(b);
    int c = memcmp(a, b, la < lb ? la : lb);
    if (c) return c;
    return (la > lb) - (la < lb);
}

static bool junctions_lt(const ipxo_junction &j1, const ipxo_junction &j2) {
    int c = name_cmp(j1.chrom, j2.chrom);
    if (c < 0) return true;
    if (c > 0) return false;
    if (j1.anchor_start < j2.anchor_start) return true;
    if (j1.anchor_start > j2.anchor_start) return false;
    if (j1.anchor_end < j2.anchor_end) return true;
    if (j1.anchor_end > j2.anchor_end) return false;
    return j1.strand < j2.strand;
}

static bool introns_lt(const ipxo_junction &j1, const ipxo_junction &j2) {
    int c = name_cmp(j1.chrom, j2.chrom);
    if (c < 0) return true;
    if (c > 0) return false;
    if (j1.intron_start < j2.intron_start) return true;
    if (j1.intron_start > j2.intron_start) return false;
    if (j1.intron_end < j2.intron_end) return true;
    if (j1.intron_end > j2.intron_end) return false;
    return j1.strand < j2.strand;
}

/* The reference sorts a vector<Junction*>; sorting the views by value applies
 * the same sequence of comparisons and moves. */
extern "C" void ipxo_sort_by_anchors(ipxo_junction *v, size_t n) { std::sort(v, v + n, junctions_lt); }
extern "C" void ipxo_sort_by_introns(ipxo_junction *v, size_t n) { std::sort(v, v + n, introns_lt); }
