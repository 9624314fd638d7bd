/* ipxo_out.c -- oracle: TSV / BED writers.
 *
 * TEST INFRASTRUCTURE ONLY (see ipxo.h).  Restates src/junctions.cc:161-268
 * (row formats, BED score clamp, colours, confidence filter at print time) and
 * src/AutoGzip.hh:9-11,40-65 (".gz" suffix switches to gzip).
 * C++ stream formatting restated: `out << float` with out.precision(3) in the
 * default float format is printf("%.3g", (double)f).
 */
#include "ipxo.h"
#include <stdlib.h>
#include <string.h>
#include <stdarg.h>
#include <zlib.h>

struct ipxo_out {
    FILE *fh;
    gzFile gz;
    int owned;
};

ipxo_out *ipxo_out_open(const char *path) {
    ipxo_out *o = (ipxo_out *)calloc(1, sizeof(*o));
    size_t n = strlen(path);
    o->owned = 1;
    if (n >= 3 && strcmp(path + n - 3, ".gz") == 0) {
        o->gz = gzopen(path, "wb");
        if (!o->gz) { free(o); return NULL; }
    } else {
        o->fh = fopen(path, "w");
        if (!o->fh) { free(o); return NULL; }
    }
    return o;
}

ipxo_out *ipxo_out_wrap(FILE *fh) {
    ipxo_out *o = (ipxo_out *)calloc(1, sizeof(*o));
    o->fh = fh;
    return o;
}

void ipxo_out_close(ipxo_out *o) {
    if (!o) return;
    if (o->gz) gzclose(o->gz);
    else if (o->owned) fclose(o->fh);
    else fflush(o->fh);
    free(o);
}

void ipxo_out_printf(ipxo_out *o, const char *fmt, ...) {
    char buf[4096];
    va_list ap;
    va_start(ap, fmt);
    int n = vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (n < 0) return;
    if ((size_t)n >= sizeof buf) n = (int)sizeof buf - 1;
    if (o->gz) gzwrite(o->gz, buf, (unsigned)n);
    else fwrite(buf, 1, (size_t)n, o->fh);
}

/* junctions.cc:162-164 */
static unsigned bed_score(uint64_t read_count) {
    return (read_count <= 1000) ? (unsigned)read_count : 1000;
}

/* junctions.cc:167-185.  Only "GT/AG" and "AT/AC" of the listed strings can
 * occur as upper-case splice sites; anything else with a motif is red. */
static const char *bed_color(const char *ss) {
    if (ss[0] == 0) return "64,128,64";
    if (strcmp(ss, "GT/AG") == 0) return "0,128,0";
    if (strcmp(ss, "AT/AC") == 0 || strcmp(ss, "AT/AG") == 0 || strcmp(ss, "GT/AC") == 0)
        return "0,0,128";
    return "128,0,0";
}

static uint64_t total_reads(const ipxo_junction *j) {
    return j->read_counts[0] + j->read_counts[1] + j->read_counts[2];
}

static void print_name(const ipxo_junction *j, ipxo_out *o) {
    ipxo_out_printf(o, "sj%u", j->ijunc);
    if (j->splice_sites[0]) ipxo_out_printf(o, "_%s", j->splice_sites);
}

/* junctions.cc:251-256 */
void ipxo_print_junction_call_header(ipxo_out *o) {
    ipxo_out_printf(o, "chrom\tintron_start\tintron_end\tstrand\tuniq_mapped_count\tmulti_mapped_count\t"
                       "unsure_mapped_count\tmax_left_overhang\tmax_right_overhang\tconfidence\t"
                       "splice_sites\tname\n");
}

/* junctions.cc:189-200, :228-236.  All subtractions are uint32 (wrapping). */
void ipxo_print_anchor_bed(const ipxo_junction *v, size_t n, float min_conf, ipxo_out *o) {
    for (size_t i = 0; i < n; i++) {
        const ipxo_junction *j = &v[i];
        if (!(j->confidence >= min_conf)) continue;
        ipxo_out_printf(o, "%s\t%u\t%u\t", j->chrom, j->anchor_start, j->anchor_end);
        print_name(j, o);
        ipxo_out_printf(o, "\t%u\t%c\t%u\t%u\t%s\t2\t%u,%u\t0,%u\n", bed_score(total_reads(j)),
                        j->strand, j->anchor_start, j->anchor_end, bed_color(j->splice_sites),
                        (uint32_t)(j->intron_start - j->anchor_start),
                        (uint32_t)(j->anchor_end - j->intron_end),
                        (uint32_t)(j->intron_end - j->anchor_start));
    }
}

/* junctions.cc:204-217, :239-248 */
void ipxo_print_intron_bed(const ipxo_junction *v, size_t n, float min_conf, int columns, ipxo_out *o) {
    for (size_t i = 0; i < n; i++) {
        const ipxo_junction *j = &v[i];
        if (!(j->confidence >= min_conf)) continue;
        ipxo_out_printf(o, "%s\t%u\t%u\t", j->chrom, j->intron_start, j->intron_end);
        print_name(j, o);
        ipxo_out_printf(o, "\t%u\t%c", bed_score(total_reads(j)), j->strand);
        if (columns == 9)
            ipxo_out_printf(o, "\t%u\t%u\t%s", j->intron_start, j->intron_end, bed_color(j->splice_sites));
        ipxo_out_printf(o, "\n");
    }
}

/* junctions.cc:220-225, :259-268 */
void ipxo_print_intron_call_tsv(const ipxo_junction *v, size_t n, float min_conf, ipxo_out *o) {
    for (size_t i = 0; i < n; i++) {
        const ipxo_junction *j = &v[i];
        if (!(j->confidence >= min_conf)) continue;
        ipxo_out_printf(o, "%s\t%u\t%u\t%c\t%llu\t%llu\t%llu\t%u\t%u\t%.3g\t%s\tsj%u\n", j->chrom,
                        j->intron_start, j->intron_end, j->strand,
                        (unsigned long long)j->read_counts[0], (unsigned long long)j->read_counts[1],
                        (unsigned long long)j->read_counts[2],
                        (uint32_t)(j->intron_start - j->anchor_start),
                        (uint32_t)(j->anchor_end - j->intron_end),
                        (double)j->confidence, j->splice_sites, j->ijunc);
    }
}
