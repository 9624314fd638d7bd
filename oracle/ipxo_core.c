/* ipxo_core.c -- oracle: per-read CIGAR state machine, junction QC, strand,
 * read category, splice-site lookup, table create/update, confidence.
 *
 * TEST INFRASTRUCTURE ONLY (see ipxo.h).  Sequential restatement of
 *   src/junctions_extractor.cc, src/junctions.{hh,cc}, src/splice_juncs.cc,
 *   src/genome.cc (dna_reverse_complement)
 * of diekhans/intronProspector v1.4.0; each function cites the lines it follows.
 */
#include "ipxo.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <ctype.h>
#include <stdarg.h>

/* ------------------------------------------------------------------------ */
/* small helpers                                                            */

static void *xrealloc(void *p, size_t n) {
    void *q = realloc(p, n ? n : 1);
    if (!q) { fprintf(stderr, "ipxo: out of memory\n"); abort(); }
    return q;
}

static char *xstrdup(const char *s) {
    size_t n = strlen(s) + 1;
    char *d = (char *)xrealloc(NULL, n);
    memcpy(d, s, n);
    return d;
}

/* std::string operator< : char_traits<char>::compare is memcmp (unsigned
 * bytes), then length.  Used by JunctionKey::operator< junctions.hh:77-100. */
static int name_cmp(const char *a, const char *b) {
    size_t la = strlen(a), lb = strlen(b);
    size_t m = la < lb ? la : lb;
    int c = memcmp(a, b, m);
    if (c) return c;
    return (la > lb) - (la < lb);
}

/* ------------------------------------------------------------------------ */
/* canonical motifs: splice_juncs.cc:10-38.  The reference builds the set of
 * every upper/lower-case spelling of GT/AG, GC/AG, AT/AC; membership is
 * therefore a case-insensitive compare of a 5-byte string.                 */

int ipxo_is_canonical(const char *s) {
    static const char *canon[3] = { "GT/AG", "GC/AG", "AT/AC" };
    if (strlen(s) != 5) return 0;
    for (int k = 0; k < 3; k++) {
        int ok = 1;
        for (int i = 0; i < 5; i++) {
            /* the set is generated with ::toupper / ::tolower of the canonical
             * letters only, so a match is: same letter in either case */
            char c = s[i], w = canon[k][i];
            if (!(c == w || c == (char)tolower((unsigned char)w))) { ok = 0; break; }
        }
        if (ok) return 1;
    }
    return 0;
}

/* genome.cc:12-42: table maps ACGT/acgt, everything else passes through;
 * the string is reversed. */
void ipxo_revcomp(char *s, int n) {
    char tmp[64];
    if (n > 64) n = 64;
    for (int i = 0; i < n; i++) {
        char c = s[i], r = c;
        switch (c) {
            case 'A': r = 'T'; break;  case 'T': r = 'A'; break;
            case 'G': r = 'C'; break;  case 'C': r = 'G'; break;
            case 'a': r = 't'; break;  case 't': r = 'a'; break;
            case 'g': r = 'c'; break;  case 'c': r = 'g'; break;
            default: break;
        }
        tmp[n - 1 - i] = r;
    }
    memcpy(s, tmp, (size_t)n);
}

static void str_upper(char *s) { for (; *s; s++) *s = (char)toupper((unsigned char)*s); }
static void str_lower(char *s) { for (; *s; s++) *s = (char)tolower((unsigned char)*s); }

/* ------------------------------------------------------------------------ */
/* table                                                                    */

typedef struct {
    int chrom_id;                 /* index into x->names */
    uint32_t intron_start, intron_end, anchor_start, anchor_end;
    char strand;
    char splice_sites[8];
    uint32_t ijunc;
    uint64_t read_counts[3];
    uint16_t *offsets;            /* junctions.hh:121 vector<uint16_t> */
    uint64_t n_offsets, cap_offsets;
    float confidence;             /* NaN until computed, junctions.hh:122 */
} junc_t;

struct ipxo {
    ipxo_params p;
    /* interned chromosome names */
    char **names; int n_names, cap_names;
    /* targets_: vector<string>, appended by every open() */
    int *targets; int n_targets, cap_targets;
    /* junction storage + open-addressing index on (chrom_id,start,end) */
    junc_t *juncs; size_t n_juncs, cap_juncs;
    int64_t *slots; size_t n_slots;      /* -1 empty else index into juncs */
    uint32_t ijunc;                      /* junctions_extractor.hh:129 */
    const ipxo_genome *genome;
    /* missing_genomic_warned_ (set<string>) as a flag per interned name */
    unsigned char *warned; int cap_warned;
    ipxo_out *trace;
    FILE *warn;
    /* reader state, junctions_extractor.hh:121-134 */
    ipxo_sam *sam;
    int previous_target;
    int aln_pending;
    ipxo_aln pending;
    char *bam_name;
    /* scratch for map-order listing */
    size_t *order; size_t cap_order;
    char errmsg[512];
};

void ipxo_default_params(ipxo_params *p) {
    /* intronProspector.cc:44-50 and :289-292 (indels not allowed => size 0) */
    p->min_anchor_length = 8;
    p->max_anchor_indel_size = 0;
    p->min_intron_length = 70;
    p->max_intron_length = 500000;
    p->junction_filter = IPXO_SJ_NULL;
    p->strandness = IPXO_UNSTRANDED;
    p->excludes = IPXO_EXCLUDE_NONE;
    p->skip_missing_targets = 0;
}

ipxo_t *ipxo_create(const ipxo_params *p) {
    ipxo_t *x = (ipxo_t *)calloc(1, sizeof(*x));
    if (!x) return NULL;
    x->p = *p;
    x->previous_target = -1;
    x->warn = stderr;
    x->n_slots = 1024;
    x->slots = (int64_t *)xrealloc(NULL, x->n_slots * sizeof(int64_t));
    for (size_t i = 0; i < x->n_slots; i++) x->slots[i] = -1;
    return x;
}

void ipxo_clear(ipxo_t *x) {
    for (size_t i = 0; i < x->n_juncs; i++) free(x->juncs[i].offsets);
    x->n_juncs = 0;
    for (size_t i = 0; i < x->n_slots; i++) x->slots[i] = -1;
}

void ipxo_destroy(ipxo_t *x) {
    if (!x) return;
    ipxo_clear(x);
    if (x->sam) ipxo_close(x);
    for (int i = 0; i < x->n_names; i++) free(x->names[i]);
    free(x->names); free(x->targets); free(x->juncs); free(x->slots);
    free(x->warned); free(x->order); free(x->bam_name);
    free(x);
}

const char *ipxo_errmsg(const ipxo_t *x) { return x->errmsg; }
void ipxo_set_genome(ipxo_t *x, const ipxo_genome *g) { x->genome = g; }
void ipxo_set_trace(ipxo_t *x, void *fh) { x->trace = (ipxo_out *)fh; }
void ipxo_set_warn_stream(ipxo_t *x, FILE *fh) { x->warn = fh; }
int ipxo_num_targets(const ipxo_t *x) { return x->n_targets; }

static int intern_name(ipxo_t *x, const char *name) {
    for (int i = 0; i < x->n_names; i++)
        if (strcmp(x->names[i], name) == 0) return i;
    if (x->n_names == x->cap_names) {
        x->cap_names = x->cap_names ? 2 * x->cap_names : 64;
        x->names = (char **)xrealloc(x->names, (size_t)x->cap_names * sizeof(char *));
    }
    x->names[x->n_names] = xstrdup(name);
    return x->n_names++;
}

/* junctions_extractor.cc:564-568 */
void ipxo_add_targets(ipxo_t *x, int n, const char *const *names) {
    for (int i = 0; i < n; i++) {
        if (x->n_targets == x->cap_targets) {
            x->cap_targets = x->cap_targets ? 2 * x->cap_targets : 64;
            x->targets = (int *)xrealloc(x->targets, (size_t)x->cap_targets * sizeof(int));
        }
        x->targets[x->n_targets++] = intern_name(x, names[i]);
    }
}

static uint64_t key_hash(int chrom_id, uint32_t s, uint32_t e) {
    uint64_t h = ((uint64_t)s << 32) | e;
    h ^= (uint64_t)(uint32_t)chrom_id * 0x9E3779B97F4A7C15ull;
    h ^= h >> 33; h *= 0xff51afd7ed558ccdull; h ^= h >> 33;
    h *= 0xc4ceb9fe1a85ec53ull; h ^= h >> 33;
    return h;
}

static void index_insert(ipxo_t *x, size_t ji) {
    const junc_t *j = &x->juncs[ji];
    size_t m = x->n_slots - 1;
    size_t s = (size_t)key_hash(j->chrom_id, j->intron_start, j->intron_end) & m;
    while (x->slots[s] >= 0) s = (s + 1) & m;
    x->slots[s] = (int64_t)ji;
}

static void index_grow(ipxo_t *x) {
    x->n_slots *= 2;
    x->slots = (int64_t *)xrealloc(x->slots, x->n_slots * sizeof(int64_t));
    for (size_t i = 0; i < x->n_slots; i++) x->slots[i] = -1;
    for (size_t i = 0; i < x->n_juncs; i++) index_insert(x, i);
}

/* junctions_.count(key): JunctionKey truncates positions to uint32
 * (junctions.hh:57-67) */
static junc_t *table_find(ipxo_t *x, int chrom_id, uint32_t s, uint32_t e) {
    size_t m = x->n_slots - 1;
    size_t h = (size_t)key_hash(chrom_id, s, e) & m;
    while (x->slots[h] >= 0) {
        junc_t *j = &x->juncs[x->slots[h]];
        if (j->chrom_id == chrom_id && j->intron_start == s && j->intron_end == e) return j;
        h = (h + 1) & m;
    }
    return NULL;
}

static junc_t *table_add(ipxo_t *x) {
    if (x->n_juncs == x->cap_juncs) {
        x->cap_juncs = x->cap_juncs ? 2 * x->cap_juncs : 256;
        x->juncs = (junc_t *)xrealloc(x->juncs, x->cap_juncs * sizeof(junc_t));
    }
    junc_t *j = &x->juncs[x->n_juncs];
    memset(j, 0, sizeof(*j));
    return j;
}

/* Junction::count_read junctions.hh:163-167: offset stored as uint16_t */
static void count_read(junc_t *j, int category, uint32_t read_offset) {
    j->read_counts[category] += 1;
    if (j->n_offsets == j->cap_offsets) {
        j->cap_offsets = j->cap_offsets ? 2 * j->cap_offsets : 4;
        j->offsets = (uint16_t *)xrealloc(j->offsets, j->cap_offsets * sizeof(uint16_t));
    }
    j->offsets[j->n_offsets++] = (uint16_t)read_offset;
}

/* ------------------------------------------------------------------------ */
/* per-alignment attributes                                                 */

/* junctions_extractor.cc:59-68 over get_num_aligns :49-56 */
static int category_from_nh(int nh) {
    if (nh <= 0) return IPXO_UNSURE;
    if (nh == 1) return IPXO_SINGLE;
    return IPXO_MULTI;
}

/* junctions_extractor.cc:71-95; returns -1 for the logic_error at :73-74 */
static int read_category(uint32_t flag, int nh) {
    if (flag & (IPXO_FQCFAIL | IPXO_FUNMAP | IPXO_FDUP)) return -1;
    if (flag & IPXO_FSECONDARY) return IPXO_MULTI;
    if (flag & IPXO_FSUPPLEMENTARY) return IPXO_MULTI;
    if (flag & IPXO_FPAIRED) {
        if ((flag & IPXO_FPROPER_PAIR) && !(flag & IPXO_FUNMAP)) return category_from_nh(nh);
        return IPXO_UNSURE;   /* mate unmapped / other chrom / not proper: all UNSURE (:84-90) */
    }
    return category_from_nh(nh);
}

/* junctions_extractor.cc:333-359 */
static char strand_from_flag(uint32_t flag, int strandness) {
    int reversed = (flag & IPXO_FREVERSE) != 0;
    int mate_reversed = (flag & IPXO_FMREVERSE) != 0;
    int first_in_pair = (flag & IPXO_FREAD1) != 0;
    int second_in_pair = (flag & IPXO_FREAD2) != 0;
    int first_strand = first_in_pair ^ reversed;
    int second_strand = second_in_pair ^ mate_reversed;
    if (strandness != IPXO_UNSTRANDED) {
        int bool_strandness = strandness - 1;          /* RF -> 0, FR -> 1 */
        first_strand ^= !bool_strandness;
        second_strand ^= !bool_strandness;
    }
    char strand = first_strand ? '+' : '-';
    return (first_strand == second_strand) ? strand : '.';
}

/* junctions_extractor.cc:362-371 with :322-330 */
static char junction_strand(const ipxo_t *x, uint32_t flag, int xs_present, char xs_char) {
    if (x->p.strandness != IPXO_UNSTRANDED) return strand_from_flag(flag, x->p.strandness);
    if (xs_present) return xs_char ? xs_char : '.';
    return '.';
}

/* ------------------------------------------------------------------------ */
/* record_junction and below                                                */

typedef struct {
    int32_t tid; int64_t pos; uint32_t flag; int nh; const char *qname;
} aln_ctx;

/* junctions_extractor.cc:384-401.  hts_pos_t is int64; the counters are
 * uint32; "(a - b) - (u32 + u32) < u32" is therefore an int64 comparison of
 * an int64 minus a (wrapping) uint32 sum. */
static int junction_qc(const ipxo_t *x, const aln_ctx *a, int64_t anchor_start, int64_t intron_start,
                       int64_t intron_end, int64_t anchor_end,
                       uint32_t lmm, uint32_t rmm, uint32_t lind, uint32_t rind) {
    if (a->flag & IPXO_FSECONDARY) return 0;
    if ((x->p.excludes & IPXO_EXCLUDE_MULTI) && (a->nh > 1)) return 0;
    if ((intron_end - intron_start < (int64_t)x->p.min_intron_length) ||
        (intron_end - intron_start > (int64_t)x->p.max_intron_length)) return 0;
    if ((((intron_start - anchor_start) - (int64_t)(uint32_t)(lmm + lind)) < (int64_t)x->p.min_anchor_length) ||
        (((anchor_end - intron_end) - (int64_t)(uint32_t)(rmm + rind)) < (int64_t)x->p.min_anchor_length)) return 0;
    return 1;
}

static void coords_str(char *buf, size_t n, const char *chrom, int start, int end) {
    snprintf(buf, n, "%s:%d-%d", chrom, start, end);     /* type_ops.hh:57-59 */
}

/* Genome::fetch genome.cc:62-74: (int) coordinates, end exclusive. */
static int genome_fetch(ipxo_t *x, const char *chrom, int start, int end, char *out /* >= 8 */) {
    char tmp[16];
    int retlen = ipxo_genome_fetch(x->genome, chrom, start, end - 1, tmp, (int)sizeof(tmp) - 1);
    char cs[400];
    if (retlen == -2) {
        coords_str(cs, sizeof cs, chrom, start, end);
        snprintf(x->errmsg, sizeof x->errmsg, "can't load genome sequence for: %s", cs);
        return IPXO_ERR_GENOME;
    }
    if (retlen != (end - start)) {
        coords_str(cs, sizeof cs, chrom, start, end);
        snprintf(x->errmsg, sizeof x->errmsg, "unexpected length returned for: %s: %d", cs, retlen);
        return IPXO_ERR_GENOME;
    }
    memcpy(out, tmp, (size_t)retlen);
    out[retlen] = 0;
    return IPXO_OK;
}

/* junctions_extractor.cc:405-428.  With g++ the operands of
 *   fetch(start,start+2) + "/" + fetch(end-2,end)
 * are evaluated right to left (verified with g++ 11.4 -O0/-O2 on a stand-alone
 * probe), so the acceptor-side fetch runs, and fails, first. */
static int get_splice_sites(ipxo_t *x, int chrom_id, char strand, int64_t intron_start,
                            int64_t intron_end, char *ss /* >= 8 */) {
    const char *chrom = x->names[chrom_id];
    ss[0] = 0;
    if (x->genome == NULL) return IPXO_OK;
    if (x->p.skip_missing_targets && !ipxo_genome_has_seq(x->genome, chrom)) {
        if (chrom_id >= x->cap_warned) {
            int nc = x->cap_names > chrom_id ? x->cap_names : chrom_id + 1;
            x->warned = (unsigned char *)xrealloc(x->warned, (size_t)nc);
            memset(x->warned + x->cap_warned, 0, (size_t)(nc - x->cap_warned));
            x->cap_warned = nc;
        }
        if (!x->warned[chrom_id]) {
            x->warned[chrom_id] = 1;
            fprintf(x->warn, "Warning: genomic sequence not found for %s splice junctions not available\n", chrom);
            fflush(x->warn);
        }
        return IPXO_OK;
    }
    char acc[8], don[8];
    int rc = genome_fetch(x, chrom, (int)(intron_end - 2), (int)intron_end, acc);
    if (rc) return rc;
    rc = genome_fetch(x, chrom, (int)intron_start, (int)(intron_start + 2), don);
    if (rc) return rc;
    ss[0] = don[0]; ss[1] = don[1]; ss[2] = '/'; ss[3] = acc[0]; ss[4] = acc[1]; ss[5] = 0;
    str_upper(ss);
    if (strand == '-') ipxo_revcomp(ss, 5);
    if (!ipxo_is_canonical(ss)) str_lower(ss);
    return IPXO_OK;
}

/* junctions_extractor.cc:431-444 */
static char correct_strand(char strand, char *ss) {
    char corrected = strand;
    if (ss[0] && strand == '.') {
        if (ipxo_is_canonical(ss)) {
            corrected = '+';
        } else {
            char rc[8];
            strcpy(rc, ss);
            ipxo_revcomp(rc, 5);
            if (ipxo_is_canonical(rc)) {
                corrected = '-';
                str_upper(rc);
                strcpy(ss, rc);
            }
        }
    }
    return corrected;
}

/* splice_juncs.cc:52-61 */
static int junction_filter_check(int filter, const char *ss) {
    if (ss[0] == 0) return 1;
    if (filter == IPXO_SJ_CANON) return ipxo_is_canonical(ss);
    return 1;
}

static void trace_row(ipxo_t *x, const aln_ctx *a, const char *chrom, int64_t is, int64_t ie,
                      char strand, int have_jstrand, char jstrand, const char *ss,
                      int64_t as, int64_t ae) {
    /* junctions_extractor.cc:451-457 (create: empty field) / :471-477 (update) */
    if (have_jstrand)
        ipxo_out_printf(x->trace, "%s\t%lld\t%lld\t%s\t%u\t%c\t%c\t%s\t%lld\t%lld\n", chrom,
                        (long long)is, (long long)ie, a->qname ? a->qname : "", a->flag, strand,
                        jstrand, ss, (long long)as, (long long)ae);
    else
        ipxo_out_printf(x->trace, "%s\t%lld\t%lld\t%s\t%u\t%c\t\t%s\t%lld\t%lld\n", chrom,
                        (long long)is, (long long)ie, a->qname ? a->qname : "", a->flag, strand,
                        ss, (long long)as, (long long)ae);
}

static int category_or_error(ipxo_t *x, const aln_ctx *a, int *cat) {
    *cat = read_category(a->flag, a->nh);
    if (*cat < 0) {
        snprintf(x->errmsg, sizeof x->errmsg, "should never look at FQCAIL, FUNMAP, FDUP reads");
        return IPXO_ERR_CATEGORY;
    }
    return IPXO_OK;
}

/* add_junction junctions_extractor.cc:488-506 with create :446-464 and update :466-485.
 * *junc_out = NULL when the filter rejected a new key. */
static int add_junction(ipxo_t *x, const aln_ctx *a, int chrom_id, char strand,
                        int64_t anchor_start, int64_t intron_start, int64_t intron_end,
                        int64_t anchor_end, junc_t **junc_out) {
    uint32_t ks = (uint32_t)intron_start, ke = (uint32_t)intron_end;
    junc_t *j = table_find(x, chrom_id, ks, ke);
    int cat, rc;
    *junc_out = NULL;
    if (j == NULL) {
        char ss[8];
        rc = get_splice_sites(x, chrom_id, strand, intron_start, intron_end, ss);
        if (rc) return rc;
        strand = correct_strand(strand, ss);
        if (!junction_filter_check(x->p.junction_filter, ss)) return IPXO_OK;
        if (x->trace)
            trace_row(x, a, x->names[chrom_id], intron_start, intron_end, strand, 0, 0, ss,
                      anchor_start, anchor_end);
        if ((x->n_juncs + 1) * 2 > x->n_slots) index_grow(x);
        j = table_add(x);
        j->chrom_id = chrom_id;
        j->intron_start = ks; j->intron_end = ke;
        j->anchor_start = (uint32_t)anchor_start; j->anchor_end = (uint32_t)anchor_end;
        j->strand = strand;
        strcpy(j->splice_sites, ss);
        j->ijunc = x->ijunc++;
        j->confidence = NAN;
        x->n_juncs++;
        index_insert(x, x->n_juncs - 1);
        rc = category_or_error(x, a, &cat);
        if (rc) return rc;
        count_read(j, cat, (uint32_t)(intron_start - a->pos));
    } else {
        if (x->trace)
            trace_row(x, a, x->names[chrom_id], intron_start, intron_end, strand, 1, j->strand,
                      j->splice_sites, anchor_start, anchor_end);
        /* int64 vs uint32 compares, :479-482 */
        if (anchor_start < (int64_t)j->anchor_start) j->anchor_start = (uint32_t)anchor_start;
        if (anchor_end > (int64_t)j->anchor_end) j->anchor_end = (uint32_t)anchor_end;
        rc = category_or_error(x, a, &cat);
        if (rc) return rc;
        count_read(j, cat, (uint32_t)(intron_start - a->pos));
    }
    *junc_out = j;
    return IPXO_OK;
}

/* junctions_extractor.cc:509-530 (orientCnt feeds only the tag rewrite, which is
 * outside the hot path; it is tracked so that the restatement is complete) */
static int record_junction(ipxo_t *x, const aln_ctx *a, int chrom_id, char strand,
                           int64_t anchor_start, int64_t intron_start, int64_t intron_end,
                           int64_t anchor_end, uint32_t lmm, uint32_t rmm, uint32_t lind,
                           uint32_t rind, int *orient_cnt) {
    if (junction_qc(x, a, anchor_start, intron_start, intron_end, anchor_end, lmm, rmm, lind, rind)) {
        junc_t *j;
        int rc = add_junction(x, a, chrom_id, strand, anchor_start, intron_start, intron_end,
                              anchor_end, &j);
        if (rc) return rc;
        if (j && j->splice_sites[0] >= 'A' && j->splice_sites[0] <= 'Z')   /* junctions.hh:187-191 */
            *orient_cnt += (j->strand == '+') ? 1 : -1;
    }
    return IPXO_OK;
}

/* ------------------------------------------------------------------------ */
/* ReadJunctionExtractor, junctions_extractor.cc:114-319                    */

enum { ST_LIMBO = 1, ST_LEFT = 2, ST_INTRON = 4, ST_RIGHT = 8 };

typedef struct {
    ipxo_t *x;
    const aln_ctx *a;
    int chrom_id;
    char strand;
    int state;
    int64_t target_pos, anchor_start, intron_start, intron_end, anchor_end;
    uint32_t lmm, rmm, lind, rind, cur_indel;
    int orient_cnt;
} rje_t;

static int rje_record(rje_t *r) {
    return record_junction(r->x, r->a, r->chrom_id, r->strand, r->anchor_start, r->intron_start,
                           r->intron_end, r->anchor_end, r->lmm, r->rmm, r->lind, r->rind,
                           &r->orient_cnt);
}

/* :153-158 */
static int rje_enter_limbo(rje_t *r) {
    int rc = IPXO_OK;
    if (r->state == ST_RIGHT) rc = rje_record(r);
    r->state = ST_LIMBO;
    return rc;
}

/* :175-181 */
static void rje_count_indels(rje_t *r, uint32_t len) {
    if (r->state == ST_LEFT) r->lind += len;
    else if (r->state == ST_RIGHT) r->rind += len;
}

/* :208-229 */
static int rje_enter_intron(rje_t *r, uint32_t len) {
    if (r->state == ST_RIGHT) {
        int rc = rje_record(r);
        if (rc) return rc;
        r->anchor_start = r->intron_end;
        r->lmm = r->rmm; r->rmm = 0;          /* shift_anchor_counts :160-165 */
        r->lind = r->rind; r->rind = 0;
        r->state = ST_LEFT;
    }
    if (r->state == ST_LEFT) {
        r->intron_start = r->target_pos;
        r->intron_end = r->target_pos + len;
        r->state = ST_INTRON;
    }
    r->cur_indel = 0;
    r->target_pos += len;
    return IPXO_OK;
}

/* :232-254 */
static void rje_enter_aligned(rje_t *r, char op, uint32_t len) {
    if (r->state == ST_LIMBO) {
        r->anchor_start = r->target_pos;
        r->anchor_end = r->target_pos + len;
        r->lmm = r->rmm = 0;                  /* NB: indel counters are not reset */
        r->state = ST_LEFT;
    } else if (r->state & (ST_LEFT | ST_RIGHT)) {
        r->anchor_end = r->target_pos + len;
    } else if (r->state == ST_INTRON) {
        r->intron_end = r->target_pos;
        r->anchor_end = r->target_pos + len;
        r->state = ST_RIGHT;
    }
    if (op == 'X') {                          /* count_mismatchs :167-173 */
        if (r->state == ST_LEFT) r->lmm += len;
        else if (r->state == ST_RIGHT) r->rmm += len;
    }
    r->cur_indel = 0;
    r->target_pos += len;
}

/* :257-265 (D) and :268-275 (I) */
static int rje_enter_indel(rje_t *r, uint32_t len, int consumes_ref) {
    int rc = IPXO_OK;
    rje_count_indels(r, len);
    r->cur_indel += len;
    if (r->cur_indel > r->x->p.max_anchor_indel_size) rc = rje_enter_limbo(r);
    if (consumes_ref) r->target_pos += len;
    return rc;
}

/* :285-318; op chars per BAM_CIGAR_STR "MIDNSHP=XB", codes >= 10 print '?' */
static int rje_parse_read(rje_t *r, const uint32_t *cigar, uint32_t n_cigar) {
    static const char opchr[] = "MIDNSHP=XB??????";
    for (uint32_t i = 0; i < n_cigar; i++) {
        char op = opchr[cigar[i] & 0xf];
        uint32_t len = cigar[i] >> 4;
        int rc = IPXO_OK;
        switch (op) {
            case 'N': rc = rje_enter_intron(r, len); break;
            case '=': case 'M': case 'X': rje_enter_aligned(r, op, len); break;
            case 'D': rc = rje_enter_indel(r, len, 1); break;
            case 'I': rc = rje_enter_indel(r, len, 0); break;
            case 'S': case 'H': rc = rje_enter_limbo(r); break;
            default:
                snprintf(r->x->errmsg, sizeof r->x->errmsg, "Unknown cigar operation '%c'", op);
                return IPXO_ERR_CIGAR;
        }
        if (rc) return rc;
    }
    if (r->state == ST_RIGHT) return rje_record(r);    /* finish :278-282 */
    return IPXO_OK;
}

/* process_alignment junctions_extractor.cc:533-549 */
int ipxo_process_alignment(ipxo_t *x, int32_t tid, int64_t pos, uint32_t flag,
                           int xs_present, char xs_char, int nh,
                           const uint32_t *cigar, uint32_t n_cigar, const char *qname) {
    if ((n_cigar <= 1) || (tid < 0) || (flag & (IPXO_FQCFAIL | IPXO_FDUP))) return IPXO_OK;
    if (tid >= x->n_targets) {
        snprintf(x->errmsg, sizeof x->errmsg, "target index %d out of range", tid);
        return IPXO_ERR_IO;
    }
    aln_ctx a = { tid, pos, flag, nh, qname };
    rje_t r;
    memset(&r, 0, sizeof r);
    r.x = x; r.a = &a;
    r.chrom_id = x->targets[tid];
    r.strand = junction_strand(x, flag, xs_present, xs_char);
    r.state = ST_LIMBO;
    r.target_pos = pos;
    r.anchor_start = r.intron_start = r.intron_end = r.anchor_end = -1;
    return rje_parse_read(&r, cigar, n_cigar);
}

int ipxo_process_batch(ipxo_t *x, uint64_t n,
                       const int32_t *tid, const int32_t *pos, const uint16_t *flag,
                       const uint8_t *xs, const uint8_t *nh,
                       const uint32_t *cigar_off, const uint32_t *n_cigar,
                       const uint32_t *cigar, uint64_t *err_index) {
    /* nh class -> a representative get_num_aligns() value: 0 -> -1, 1 -> 1, 2 -> 2 */
    static const int nh_of_class[3] = { -1, 1, 2 };
    for (uint64_t i = 0; i < n; i++) {
        int rc = ipxo_process_alignment(x, tid[i], pos[i], flag[i], xs[i] != 0, (char)xs[i],
                                        nh_of_class[nh[i] > 2 ? 2 : nh[i]],
                                        cigar + cigar_off[i], n_cigar[i], NULL);
        if (rc) { if (err_index) *err_index = i; return rc; }
    }
    return IPXO_OK;
}

/* ------------------------------------------------------------------------ */
/* confidence, junctions.cc:44-70 (float arithmetic, glibc log2f)           */

static int cmp_u16(const void *a, const void *b) {
    uint16_t x = *(const uint16_t *)a, y = *(const uint16_t *)b;
    return (x > y) - (x < y);
}

static float calculate_confidence(junc_t *j) {
    qsort(j->offsets, j->n_offsets, sizeof(uint16_t), cmp_u16);
    float summation = 0.0f;
    for (uint64_t i = 0; i < j->n_offsets; ) {
        uint16_t pos = j->offsets[i];
        unsigned cnt = 0;
        for (; (i < j->n_offsets) && (j->offsets[i] == pos); i++) cnt++;
        float p = (float)cnt / (float)j->n_offsets;    /* float / size_t -> float division */
        if (p != 0.0f) {
            volatile float t = p * log2f(p);           /* one rounding per product, no FMA */
            summation += t;
        }
    }
    return (summation == 0.0f) ? 0.0f : -summation;
}

static float get_confidence(junc_t *j) {
    if (isnan(j->confidence)) {
        uint64_t total = j->read_counts[0] + j->read_counts[1] + j->read_counts[2];
        j->confidence = (total == 0) ? 0.0f : calculate_confidence(j);
    }
    return j->confidence;
}

/* ------------------------------------------------------------------------ */
/* listing                                                                  */

static ipxo_t *g_sort_ctx;
static int cmp_map_order(const void *pa, const void *pb) {
    const junc_t *a = &g_sort_ctx->juncs[*(const size_t *)pa];
    const junc_t *b = &g_sort_ctx->juncs[*(const size_t *)pb];
    int c = name_cmp(g_sort_ctx->names[a->chrom_id], g_sort_ctx->names[b->chrom_id]);
    if (c) return c;
    if (a->intron_start != b->intron_start) return a->intron_start < b->intron_start ? -1 : 1;
    if (a->intron_end != b->intron_end) return a->intron_end < b->intron_end ? -1 : 1;
    return 0;
}

size_t ipxo_num_junctions(const ipxo_t *x) { return x->n_juncs; }

static void map_order(ipxo_t *x) {
    if (x->cap_order < x->n_juncs) {
        x->cap_order = x->n_juncs;
        x->order = (size_t *)xrealloc(x->order, x->cap_order * sizeof(size_t));
    }
    for (size_t i = 0; i < x->n_juncs; i++) x->order[i] = i;
    g_sort_ctx = x;
    qsort(x->order, x->n_juncs, sizeof(size_t), cmp_map_order);   /* keys are unique */
}

static void fill_view(ipxo_t *x, junc_t *j, ipxo_junction *o) {
    o->chrom = x->names[j->chrom_id];
    o->intron_start = j->intron_start; o->intron_end = j->intron_end;
    o->anchor_start = j->anchor_start; o->anchor_end = j->anchor_end;
    o->strand = j->strand;
    memcpy(o->splice_sites, j->splice_sites, 8);
    o->ijunc = j->ijunc;
    for (int c = 0; c < 3; c++) o->read_counts[c] = j->read_counts[c];
    o->confidence = get_confidence(j);
    o->n_offsets = j->n_offsets;
    o->offsets = j->offsets;
}

size_t ipxo_get_junctions(ipxo_t *x, ipxo_junction *out, size_t cap) {
    map_order(x);
    size_t n = x->n_juncs < cap ? x->n_juncs : cap;
    for (size_t i = 0; i < n; i++) fill_view(x, &x->juncs[x->order[i]], &out[i]);
    return n;
}

size_t ipxo_export(ipxo_t *x, int32_t *chrom_target, uint32_t *intron_start, uint32_t *intron_end,
                   uint32_t *anchor_start, uint32_t *anchor_end, uint8_t *strand, uint32_t *ijunc,
                   uint64_t *counts3, float *confidence, char *splice_sites6, uint64_t *n_offsets) {
    map_order(x);
    for (size_t i = 0; i < x->n_juncs; i++) {
        junc_t *j = &x->juncs[x->order[i]];
        int t = -1;
        for (int k = 0; k < x->n_targets; k++) if (x->targets[k] == j->chrom_id) { t = k; break; }
        chrom_target[i] = t;
        intron_start[i] = j->intron_start; intron_end[i] = j->intron_end;
        anchor_start[i] = j->anchor_start; anchor_end[i] = j->anchor_end;
        strand[i] = (uint8_t)j->strand; ijunc[i] = j->ijunc;
        for (int c = 0; c < 3; c++) counts3[3 * i + c] = j->read_counts[c];
        confidence[i] = get_confidence(j);
        memset(splice_sites6 + 6 * i, 0, 6);
        memcpy(splice_sites6 + 6 * i, j->splice_sites, strlen(j->splice_sites));
        n_offsets[i] = j->n_offsets;
    }
    return x->n_juncs;
}

uint64_t ipxo_export_offsets(ipxo_t *x, uint16_t *out, uint64_t cap) {
    map_order(x);
    uint64_t k = 0;
    for (size_t i = 0; i < x->n_juncs; i++) {
        junc_t *j = &x->juncs[x->order[i]];
        (void)get_confidence(j);            /* sorts the offsets */
        for (uint64_t t = 0; t < j->n_offsets && k < cap; t++) out[k++] = j->offsets[t];
    }
    return k;
}

/* ------------------------------------------------------------------------ */
/* read loops, junctions_extractor.cc:571-683                               */

int ipxo_open(ipxo_t *x, const char *path) {
    x->sam = ipxo_sam_open(path);
    free(x->bam_name);
    x->bam_name = xstrdup(path);
    if (!x->sam) {
        snprintf(x->errmsg, sizeof x->errmsg, "Error opening BAM/SAM/CRAM file: %s", path);
        return IPXO_ERR_IO;
    }
    ipxo_add_targets(x, ipxo_sam_n_targets(x->sam), ipxo_sam_target_names(x->sam));
    if (x->trace)   /* :579-584 */
        ipxo_out_printf(x->trace, "chrom\tintron_start\tintron_end\tqname\tflag\tstrand\tsplice_sites\tanchor_start\tanchor_end\n");
    return IPXO_OK;
}

void ipxo_close(ipxo_t *x) {
    if (x->sam) ipxo_sam_close(x->sam);
    x->sam = NULL;
    x->previous_target = -1;     /* :615-616 */
    x->aln_pending = 0;
}

/* read_align :620-634; 1 = have record, 0 = EOF, <0 error */
static int read_align(ipxo_t *x, ipxo_aln *a) {
    if (x->aln_pending) {
        x->aln_pending = 0;
        *a = x->pending;
        return 1;
    }
    int st = ipxo_sam_read(x->sam, a);
    if (st < 0) {
        snprintf(x->errmsg, sizeof x->errmsg, "Error reader BAM record: %s", x->bam_name);
        return -1;
    }
    return st;
}

static int process_aln(ipxo_t *x, const ipxo_aln *a) {
    return ipxo_process_alignment(x, a->tid, a->pos, a->flag, a->xs_present, a->xs_char, a->nh,
                                  a->cigar, a->n_cigar, a->qname);
}

/* identify_junctions_for_target :656-664 with process_target_alignment :637-652 */
int ipxo_identify_junctions_for_target(ipxo_t *x, int target_index) {
    ipxo_aln a;
    int st;
    while ((st = read_align(x, &a)) > 0) {
        if (a.tid != target_index) {
            if ((a.tid >= 0) && (a.tid < x->previous_target)) {
                snprintf(x->errmsg, sizeof x->errmsg, "BAM appears to not be coordinate sorted");
                return IPXO_ERR_UNSORTED;
            }
            x->aln_pending = 1;      /* record stays in the reader's buffer */
            x->pending = a;
            break;
        }
        int rc = process_aln(x, &a);
        if (rc) return rc;
    }
    if (st < 0) return IPXO_ERR_IO;
    x->previous_target = target_index;
    return IPXO_OK;
}

/* identify_junctions_for_bam :678-683 */
int ipxo_identify_junctions_for_bam(ipxo_t *x) {
    ipxo_aln a;
    int st;
    while ((st = read_align(x, &a)) > 0) {
        int rc = process_aln(x, &a);
        if (rc) return rc;
    }
    return st < 0 ? IPXO_ERR_IO : IPXO_OK;
}
