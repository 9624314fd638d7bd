/* ipxo_genome.c -- oracle: indexed-FASTA access.
 *
 * TEST INFRASTRUCTURE ONLY (see ipxo.h).  Stands where htslib's faidx stands
 * under src/genome.cc:44-74 of the reference.  htslib is a third-party,
 * un-vendored dependency of the reference (configure.ac:67, >= 0.20.0) and is
 * absent from this image; the three calls the reference makes are restated
 * from htslib's published behaviour (faidx.c, htslib 1.10+):
 *   fai_load3(fa, NULL, NULL, 0)  : read "<fa>.fai" (5 tab-separated columns:
 *        name, length, offset, linebases, linewidth); no index => failure
 *   faidx_has_seq(fai, name)      : name present in the index
 *   faidx_fetch_seq(fai, name, beg, end_incl, &len):
 *        unknown name => NULL, len = -2; otherwise positions are clamped
 *        (end < beg => beg = end; beg < 0 => 0; beg >= L => beg = L;
 *         end < 0 => 0; end >= L => end = L-1) and the bytes of
 *        [beg, end] are returned, skipping non-graphic characters.
 */
#include "ipxo.h"
#include <stdlib.h>
#include <string.h>
#include <ctype.h>

typedef struct {
    char *name;
    int64_t len, offset;
    int line_blen, line_len;
} fai_entry;

struct ipxo_genome {
    fai_entry *e; int n;
    unsigned char *data; size_t size;     /* whole FASTA file */
};

static const fai_entry *find_seq(const ipxo_genome *g, const char *chrom) {
    for (int i = 0; i < g->n; i++)
        if (strcmp(g->e[i].name, chrom) == 0) return &g->e[i];
    return NULL;
}

ipxo_genome *ipxo_genome_load(const char *fasta_path) {
    size_t pl = strlen(fasta_path);
    char *fai_path = (char *)malloc(pl + 5);
    if (!fai_path) return NULL;
    memcpy(fai_path, fasta_path, pl);
    memcpy(fai_path + pl, ".fai", 5);
    FILE *fi = fopen(fai_path, "r");
    free(fai_path);
    if (!fi) return NULL;
    ipxo_genome *g = (ipxo_genome *)calloc(1, sizeof(*g));
    char line[4096];
    int cap = 0;
    while (fgets(line, sizeof line, fi)) {
        char name[2048];
        long long len, off; int lb, ll;
        if (sscanf(line, "%2047[^\t]\t%lld\t%lld\t%d\t%d", name, &len, &off, &lb, &ll) != 5) continue;
        if (g->n == cap) { cap = cap ? 2 * cap : 16; g->e = (fai_entry *)realloc(g->e, (size_t)cap * sizeof(fai_entry)); }
        fai_entry *e = &g->e[g->n++];
        e->name = (char *)malloc(strlen(name) + 1);
        strcpy(e->name, name);
        e->len = len; e->offset = off; e->line_blen = lb; e->line_len = ll;
    }
    fclose(fi);
    FILE *fa = fopen(fasta_path, "rb");
    if (!fa) { ipxo_genome_free(g); return NULL; }
    fseek(fa, 0, SEEK_END);
    long sz = ftell(fa);
    fseek(fa, 0, SEEK_SET);
    g->data = (unsigned char *)malloc(sz > 0 ? (size_t)sz : 1);
    g->size = fread(g->data, 1, (size_t)sz, fa);
    fclose(fa);
    return g;
}

void ipxo_genome_free(ipxo_genome *g) {
    if (!g) return;
    for (int i = 0; i < g->n; i++) free(g->e[i].name);
    free(g->e); free(g->data); free(g);
}

int ipxo_genome_has_seq(const ipxo_genome *g, const char *chrom) {
    return find_seq(g, chrom) != NULL;
}

int64_t ipxo_genome_seq_len(const ipxo_genome *g, const char *chrom) {
    const fai_entry *e = find_seq(g, chrom);
    return e ? e->len : -1;
}

int ipxo_genome_fetch(const ipxo_genome *g, const char *chrom, int beg_i, int end_i,
                      char *out, int cap) {
    const fai_entry *e = find_seq(g, chrom);
    if (!e) return -2;
    int64_t beg = beg_i, end = end_i;
    if (end < beg) beg = end;
    if (beg < 0) beg = 0; else if (e->len <= beg) beg = e->len;
    if (end < 0) end = 0; else if (e->len <= end) end = e->len - 1;
    int64_t want = end + 1 - beg;
    if (want <= 0 || e->line_blen <= 0) return 0;
    int64_t p = e->offset + beg / e->line_blen * e->line_len + beg % e->line_blen;
    int l = 0;
    while (l < want && l < cap && p < (int64_t)g->size) {
        unsigned char c = g->data[p++];
        if (isgraph(c)) out[l++] = (char)c;
    }
    return l;
}
