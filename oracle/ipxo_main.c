/* ipxo_main.c -- oracle command line, used by tests to reproduce the reference's
 * golden outputs byte for byte.
 *
 * TEST INFRASTRUCTURE ONLY (see ipxo.h).  Restates the driver of
 * src/intronProspector.cc: option table :159-185, defaults :44-50, filter
 * default :274-282, indel switch :289-292, the three drivers :356-401, output
 * scheduling :330-351, error exit :63-67.  Pass-through and tag rewriting are
 * outside the hot path and not provided.
 */
#define _GNU_SOURCE
#include "ipxo.h"
#include <getopt.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <errno.h>

static void cmd_error(const char *msg) {
    fprintf(stderr, "Error: %s\nUse --help to get usage\n", msg);
    exit(1);
}

typedef struct {
    ipxo_out *junction_bed, *intron_bed, *intron_bed6, *intron_call;
} outputs;

static ipxo_out *open_or_null(const char *path) {
    if (!path || !*path) return NULL;
    ipxo_out *o = ipxo_out_open(path);
    if (!o) { fprintf(stderr, "Failed to open file: %s\n", path); exit(1); }
    return o;
}

/* intronProspector.cc:330-351 */
static void output_junctions(ipxo_t *x, float min_conf, outputs *out) {
    size_t n = ipxo_num_junctions(x);
    ipxo_junction *v = (ipxo_junction *)malloc((n ? n : 1) * sizeof(*v));
    ipxo_get_junctions(x, v, n);
    if (out->junction_bed) {
        ipxo_sort_by_anchors(v, n);
        ipxo_print_anchor_bed(v, n, min_conf, out->junction_bed);
    }
    if (out->intron_bed || out->intron_bed6 || out->intron_call) {
        ipxo_sort_by_introns(v, n);
        if (out->intron_bed) ipxo_print_intron_bed(v, n, min_conf, 9, out->intron_bed);
        if (out->intron_bed6) ipxo_print_intron_bed(v, n, min_conf, 6, out->intron_bed6);
        if (out->intron_call) ipxo_print_intron_call_tsv(v, n, min_conf, out->intron_call);
    }
    free(v);
}

int main(int argc, char **argv) {
    ipxo_params p;
    ipxo_default_params(&p);
    int allow_anchor_indels = 0, unsorted = 0;
    uint32_t max_anchor_indel_size = 36;       /* intronProspector.cc:48 */
    float min_conf = 1.0f;
    const char *genome_fa = "", *junction_bed = "", *intron_bed = "", *intron_bed6 = "",
               *intron_calls = "", *debug_trace = "";
    static const struct option long_options[] = {
        {"help", no_argument, NULL, 'h'}, {"version", no_argument, NULL, 'v'},
        {"unsorted", no_argument, NULL, 'u'},
        {"min-anchor-length", required_argument, NULL, 'a'},
        {"min-intron-length", required_argument, NULL, 'i'},
        {"max-intron-length", required_argument, NULL, 'I'},
        {"allow-anchor-indels", no_argument, NULL, 'd'},
        {"max-anchor-indel-size", required_argument, NULL, 'm'},
        {"min-confidence-score", required_argument, NULL, 'C'},
        {"strandness", required_argument, NULL, 's'},
        {"excludes", required_argument, NULL, 'X'},
        {"genome-fasta", required_argument, NULL, 'g'},
        {"sj-filter", required_argument, NULL, 'f'},
        {"skip-missing-targets", no_argument, NULL, 'S'},
        {"junction-bed", required_argument, NULL, 'j'},
        {"intron-bed", required_argument, NULL, 'n'},
        {"intron-bed6", required_argument, NULL, 'b'},
        {"intron-calls", required_argument, NULL, 'c'},
        {"pass-through", required_argument, NULL, 'p'},
        {"debug-trace", required_argument, NULL, 'D'},
        {NULL, 0, NULL, 0}};
    int c;
    while ((c = getopt_long(argc, argv, "hvua:i:I:C:s:X:g:S:j:n:b:c:p:D:f:", long_options, NULL)) != -1) {
        switch (c) {
            case 'a': p.min_anchor_length = (uint32_t)atoi(optarg); break;
            case 'u': unsorted = 1; break;
            case 'd': allow_anchor_indels = 1; break;
            case 'm': max_anchor_indel_size = (uint32_t)strtoul(optarg, NULL, 10); break;
            case 'i': p.min_intron_length = (uint32_t)strtoul(optarg, NULL, 10); break;
            case 'I': p.max_intron_length = (uint32_t)strtoul(optarg, NULL, 10); break;
            case 'C': min_conf = strtof(optarg, NULL); break;
            case 's':
                if (!strcasecmp(optarg, "UN")) p.strandness = IPXO_UNSTRANDED;
                else if (!strcasecmp(optarg, "RF")) p.strandness = IPXO_RF;
                else if (!strcasecmp(optarg, "FR")) p.strandness = IPXO_FR;
                else {
                    fprintf(stderr, "Error: invalid strandness value '%s', expected one of 'UN', 'RF', or 'FR' (case insensitive)\n", optarg);
                    return 1;
                }
                break;
            case 'X':
                if (!strcmp(optarg, "multi")) p.excludes |= IPXO_EXCLUDE_MULTI;
                else {
                    fprintf(stderr, "Error: invalid exclude category '%s', expected one of: 'multi'\n", optarg);
                    return 1;
                }
                break;
            case 'g': genome_fa = optarg; break;
            case 'f':
                if (!strcmp(optarg, "all")) p.junction_filter = IPXO_SJ_ALL;
                else if (!strcmp(optarg, "canon")) p.junction_filter = IPXO_SJ_CANON;
                else {
                    fprintf(stderr, "Error parsing command line: invalid splice junction filter '%s', expected 'all' or 'canon'\n", optarg);
                    return 1;
                }
                break;
            case 'S': p.skip_missing_targets = 1; break;
            case 'j': junction_bed = optarg; break;
            case 'n': intron_bed = optarg; break;
            case 'b': intron_bed6 = optarg; break;
            case 'c': intron_calls = optarg; break;
            case 'D': debug_trace = optarg; break;
            case 'p': cmd_error("oracle: --pass-through is outside the hot path"); break;
            case 'h': case 'v': fprintf(stderr, "ipxo_cli: oracle for intronProspector 1.4.0 hot path\n"); return 0;
            default: cmd_error("bug: option not handled");
        }
    }
    const char *stdin_name = "/dev/stdin";
    const char **bams = (const char **)&argv[optind];
    int n_bams = argc - optind;
    if (n_bams == 0) { bams = &stdin_name; n_bams = 1; }
    if (p.junction_filter == IPXO_SJ_CANON && !*genome_fa) cmd_error("--sj-filter=canon requires --genome_fa");
    if (p.junction_filter == IPXO_SJ_NULL && *genome_fa) p.junction_filter = IPXO_SJ_CANON;
    p.max_anchor_indel_size = allow_anchor_indels ? max_anchor_indel_size : 0;

    ipxo_genome *genome = NULL;
    if (*genome_fa) {
        genome = ipxo_genome_load(genome_fa);
        if (!genome) {
            char msg[600];
            snprintf(msg, sizeof msg, "can't open genome FASTA file or associated index: %s", genome_fa);
            cmd_error(msg);
        }
    }
    ipxo_out *trace = open_or_null(debug_trace);
    ipxo_t *x = ipxo_create(&p);
    ipxo_set_genome(x, genome);
    ipxo_set_trace(x, trace);
    outputs out = { open_or_null(junction_bed), open_or_null(intron_bed), open_or_null(intron_bed6),
                    open_or_null(intron_calls) };
    if (out.intron_call) ipxo_print_junction_call_header(out.intron_call);   /* :317-319 */

    int rc = 0;
    if (n_bams > 1) {                                   /* :391-401 */
        for (int i = 0; i < n_bams && !rc; i++) {
            rc = ipxo_open(x, bams[i]);
            if (!rc) rc = ipxo_identify_junctions_for_bam(x);
            if (!rc) ipxo_close(x);
        }
        if (!rc) output_junctions(x, min_conf, &out);
    } else if (unsorted) {                              /* :375-387 */
        rc = ipxo_open(x, bams[0]);
        if (!rc) rc = ipxo_identify_junctions_for_bam(x);
        if (!rc) { ipxo_close(x); output_junctions(x, min_conf, &out); }
    } else {                                            /* :356-371 */
        rc = ipxo_open(x, bams[0]);
        for (int t = 0; !rc && t < ipxo_num_targets(x); t++) {
            rc = ipxo_identify_junctions_for_target(x, t);
            if (rc) break;
            output_junctions(x, min_conf, &out);
            ipxo_clear(x);
        }
        if (!rc) ipxo_close(x);
    }
    /* on error the reference exits from cmd_error(); streams already written stay */
    ipxo_out_close(out.junction_bed); ipxo_out_close(out.intron_bed);
    ipxo_out_close(out.intron_bed6); ipxo_out_close(out.intron_call);
    ipxo_out_close(trace);
    if (rc) cmd_error(ipxo_errmsg(x));
    ipxo_destroy(x);
    ipxo_genome_free(genome);
    return 0;
}
