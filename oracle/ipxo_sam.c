/* ipxo_sam.c -- oracle: minimal SAM-text reader.
 *
 * TEST INFRASTRUCTURE ONLY (see ipxo.h).  Stands where htslib's
 * sam_open/sam_hdr_read/sam_read1/bam_aux_get/bam_aux2i/bam_aux2A stand under
 * src/junctions_extractor.cc:49-56,322-330,571-586,620-634.  htslib is absent
 * from this image; only what the hot path consumes is decoded:
 *   tid (index of RNAME among @SQ lines, '*' or unknown => -1), pos (POS-1),
 *   flag, CIGAR in BAM packing (len<<4|op, ops "MIDNSHP=XB"), first XS and NH
 *   aux fields (bam_aux2A yields 0 for a non-'A' type; bam_aux2i yields 0 for a
 *   non-integer type), qname (for --debug-trace).
 */
#define _GNU_SOURCE
#include "ipxo.h"
#include <stdlib.h>
#include <string.h>
#include <errno.h>
#include <zlib.h>

struct ipxo_sam {
    gzFile fh;                 /* gzopen reads plain text transparently */
    char **tnames; int n_targets, cap_targets;
    char *line; size_t cap_line;
    int have_line;             /* first alignment line read while scanning the header */
    uint32_t *cigar; size_t cap_cigar;
};

static int read_line(ipxo_sam *s) {
    size_t n = 0;
    for (;;) {
        if (s->cap_line - n < 2) {
            s->cap_line = s->cap_line ? 2 * s->cap_line : 65536;
            s->line = (char *)realloc(s->line, s->cap_line);
        }
        if (!gzgets(s->fh, s->line + n, (int)(s->cap_line - n))) {
            if (n == 0) return 0;
            break;
        }
        n += strlen(s->line + n);
        if (n && s->line[n - 1] == '\n') { s->line[--n] = 0; break; }
    }
    if (n && s->line[n - 1] == '\r') s->line[--n] = 0;
    return 1;
}

static void add_target(ipxo_sam *s, const char *name, size_t len) {
    if (s->n_targets == s->cap_targets) {
        s->cap_targets = s->cap_targets ? 2 * s->cap_targets : 64;
        s->tnames = (char **)realloc(s->tnames, (size_t)s->cap_targets * sizeof(char *));
    }
    char *d = (char *)malloc(len + 1);
    memcpy(d, name, len); d[len] = 0;
    s->tnames[s->n_targets++] = d;
}

ipxo_sam *ipxo_sam_open(const char *path) {
    gzFile fh = gzopen(path, "rb");
    if (!fh) return NULL;
    ipxo_sam *s = (ipxo_sam *)calloc(1, sizeof(*s));
    s->fh = fh;
    /* sam_hdr_read: consume '@' lines */
    while (read_line(s)) {
        if (s->line[0] != '@') { s->have_line = 1; break; }
        if (strncmp(s->line, "@SQ\t", 4) == 0) {
            char *p = s->line + 4;
            while (p && *p) {
                char *e = strchr(p, '\t');
                size_t fl = e ? (size_t)(e - p) : strlen(p);
                if (fl > 3 && p[0] == 'S' && p[1] == 'N' && p[2] == ':') { add_target(s, p + 3, fl - 3); break; }
                p = e ? e + 1 : NULL;
            }
        }
    }
    return s;
}

void ipxo_sam_close(ipxo_sam *s) {
    if (!s) return;
    gzclose(s->fh);
    for (int i = 0; i < s->n_targets; i++) free(s->tnames[i]);
    free(s->tnames); free(s->line); free(s->cigar); free(s);
}

int ipxo_sam_n_targets(const ipxo_sam *s) { return s->n_targets; }
const char *const *ipxo_sam_target_names(const ipxo_sam *s) { return (const char *const *)s->tnames; }

static int op_code(char c) {
    static const char ops[] = "MIDNSHP=XB";
    const char *p = strchr(ops, c);
    return (p && c) ? (int)(p - ops) : -1;
}

int ipxo_sam_read(ipxo_sam *s, ipxo_aln *a) {
    if (s->have_line) s->have_line = 0;
    else if (!read_line(s)) return 0;
    char *f[12];
    int nf = 0;
    char *p = s->line;
    char *aux = NULL;
    while (nf < 11) {
        f[nf++] = p;
        char *e = strchr(p, '\t');
        if (!e) { p = NULL; break; }
        *e = 0; p = e + 1;
    }
    if (nf < 11) return -2;
    aux = p;                       /* remainder: optional fields (may be NULL) */
    a->qname = f[0];
    a->flag = (uint32_t)strtoul(f[1], NULL, 0);
    a->tid = -1;
    if (strcmp(f[2], "*") != 0) {
        for (int i = 0; i < s->n_targets; i++)
            if (strcmp(s->tnames[i], f[2]) == 0) { a->tid = i; break; }
    }
    a->pos = strtoll(f[3], NULL, 10) - 1;
    /* CIGAR */
    size_t nc = 0;
    if (strcmp(f[5], "*") != 0) {
        const char *c = f[5];
        while (*c) {
            char *e;
            unsigned long len = strtoul(c, &e, 10);
            if (e == c) return -2;
            int op = op_code(*e);
            if (op < 0) return -2;            /* htslib: unrecognized CIGAR operator */
            if (nc == s->cap_cigar) {
                s->cap_cigar = s->cap_cigar ? 2 * s->cap_cigar : 256;
                s->cigar = (uint32_t *)realloc(s->cigar, s->cap_cigar * sizeof(uint32_t));
            }
            s->cigar[nc++] = ((uint32_t)len << 4) | (uint32_t)op;
            c = e + 1;
        }
    }
    a->n_cigar = (uint32_t)nc;
    a->cigar = s->cigar;
    /* aux: first XS and first NH, as bam_aux_get would find them */
    a->xs_present = 0; a->xs_char = 0; a->nh = -1;
    int have_nh = 0;
    while (aux && *aux) {
        char *e = strchr(aux, '\t');
        if (e) *e = 0;
        if (strlen(aux) >= 5 && aux[2] == ':' && aux[4] == ':') {
            if (aux[0] == 'X' && aux[1] == 'S' && !a->xs_present) {
                a->xs_present = 1;
                a->xs_char = (aux[3] == 'A') ? aux[5] : 0;
            } else if (aux[0] == 'N' && aux[1] == 'H' && !have_nh) {
                have_nh = 1;
                a->nh = (aux[3] == 'i') ? (int)strtoll(aux + 5, NULL, 10) : 0;
            }
        }
        aux = e ? e + 1 : NULL;
    }
    return 1;
}
