// intronProspector.cc -- command line of the MI355X build; same options, messages
// and output scheduling as the reference's driver (src/intronProspector.cc):
// option table :159-185, defaults :44-50, filter default :274-282, indel switch
// :289-292, the three drivers :356-401, output_junctions :330-351, error exit :63-67.
//
// --pass-through, --set-XS/TS-strand-tag and --debug-trace are host work over the
// device's record list (host/junctions_extractor.cc, host/pass_through.cc).
#include <chrono>
#include <unistd.h>
#include <getopt.h>
#include <future>

#include <algorithm>
#include <cerrno>
#include <cstdlib>
#include <iostream>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "genome.hh"
#include "gz_ostream.hh"
#include "junctions.hh"
#include "junctions_extractor.hh"

using namespace ipxhost;
using std::cerr;
using std::endl;
using std::string;

static const uint32_t DEFAULT_MIN_ANCHOR_LENGTH = 8;
static const uint32_t DEFAULT_MIN_INTRON_LENGTH = 70;
static const uint32_t DEFAULT_MAX_INTRON_LENGTH = 500000;
static const uint32_t DEFAULT_MAX_ANCHOR_INDEL_SIZE = 36;
static const float DEFAULT_MIN_CONFIDENCE_SCORE = 1.0;

static const char *usage_msg =
    "intronProspector [options] [readaligns ...]\n"
    "\n"
    "Find putative intron junctions in RNA-Seq alignments (SAM or BAM; stdin if omitted).\n"
    "MI355X build of the junction-extraction path of intronProspector 1.4.0; options:\n"
    "  -h, --help  -v, --version  -u, --unsorted\n"
    "  -a, --min-anchor-length=8   -i, --min-intron-length=70   -I, --max-intron-length=500000\n"
    "  --allow-anchor-indels   --max-anchor-indel-size=36   -C, --min-confidence-score=1.0\n"
    "  -s, --strandness=UN|RF|FR   -X, --excludes=multi\n"
    "  -g, --genome-fasta=fasta (needs fasta.fai)   -f, --sj-filter=all|canon   -S, --skip-missing-targets\n"
    "  -j, --junction-bed=file   -n, --intron-bed=file   -b, --intron-bed6=file   -c, --intron-calls=file\n"
    "  -D, --debug-trace=file   -p, --pass-through=file   --set-XS-strand-tag   --set-TS-strand-tag\n"
    "Outputs whose name ends in .gz are gzip-compressed.\n";

static void cmd_error(const string &msg) {
    cerr << "Error: " << msg << endl;
    cerr << "Use --help to get usage" << endl;
    exit(1);
}

// src/strconvert.hh:33-46, :14-28
static unsigned toUnsigned(const char *str) {
    char *end;
    errno = 0;
    unsigned long lnum = strtoul(str, &end, 10);
    if (end == str || *end != '\0') throw std::runtime_error("Invalid unsigned integer \"" + string(str) + "\"");
    unsigned num = (unsigned)lnum;
    if (errno != 0 || (unsigned long)num != lnum)
        throw std::runtime_error("Unsigned integer out of range \"" + string(str) + "\"");
    return num;
}

static float toFloat(const char *str) {
    char *end;
    errno = 0;
    float num = strtof(str, &end);
    if (end == str || *end != '\0' || errno != 0) {
        if (errno != 0) throw std::runtime_error("Float out of range \"" + string(str) + "\"");
        throw std::runtime_error("Invalid float \"" + string(str) + "\"");
    }
    return num;
}

struct Options {
    std::vector<string> bam_files;
    uint32_t min_anchor_length = DEFAULT_MIN_ANCHOR_LENGTH;
    uint32_t min_intron_length = DEFAULT_MIN_INTRON_LENGTH;
    uint32_t max_intron_length = DEFAULT_MAX_INTRON_LENGTH;
    bool allow_anchor_indels = false;
    bool unsorted = false;
    uint32_t max_anchor_indel_size = DEFAULT_MAX_ANCHOR_INDEL_SIZE;
    float min_confidence_score = DEFAULT_MIN_CONFIDENCE_SCORE;
    JunctionFilter junction_filter = NULL_SJ_FILTER;
    Strandness strandness = UNSTRANDED;
    unsigned excludes = EXCLUDE_NONE;
    string genome_fa;
    bool skip_missing_targets = false;
    string junction_bed, intron_bed, intron_bed6, intron_call_tsv, debug_trace_tsv, bam_pass_through;
    bool set_XS_strand_tag = false, set_TS_strand_tag = false;
    std::vector<int> devices;          // --devices (this build only): empty = GPU 0
};

// "0-7", "0,2,5" or a mix: the HIP device ordinals --devices names, in the order given
static std::vector<int> parse_devices(const string &spec) {
    std::vector<int> out;
    size_t i = 0;
    while (i < spec.size()) {
        size_t j = spec.find(',', i);
        if (j == string::npos) j = spec.size();
        const string part = spec.substr(i, j - i);
        const size_t dash = part.find('-');
        try {
            if (dash == string::npos) out.push_back(std::stoi(part));
            else
                for (int d = std::stoi(part.substr(0, dash)), e = std::stoi(part.substr(dash + 1)); d <= e; ++d) out.push_back(d);
        } catch (const std::exception &) {
            cmd_error("invalid --devices value '" + spec + "', expected a list like 0-7 or 0,2,5");
        }
        i = j + 1;
    }
    if (out.empty()) cmd_error("invalid --devices value '" + spec + "', expected a list like 0-7 or 0,2,5");
    return out;
}

static void parse_options(int argc, char *argv[], Options &o) {
    enum { OPT_XS = 256, OPT_TS = 257, OPT_DEVICES = 258 };
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
        {"set-XS-strand-tag", no_argument, NULL, OPT_XS},
        {"set-TS-strand-tag", no_argument, NULL, OPT_TS},
        {"devices", required_argument, NULL, OPT_DEVICES},       // not in the reference: GPUs to spread the input over
        {NULL, 0, NULL, 0}};
    // the reference's short-option string (intronProspector.cc:185), quirks included
    const char *short_options = "hvua:i:I:C:s:X:g:S:j:n:b:c:p:D:f:";
    int c;
    while ((c = getopt_long(argc, argv, short_options, long_options, NULL)) != -1) {
        switch (c) {
            case 'a': o.min_anchor_length = (uint32_t)atoi(optarg); break;
            case 'u': o.unsorted = true; break;
            case 'd': o.allow_anchor_indels = true; break;
            case 'm': o.max_anchor_indel_size = toUnsigned(optarg); break;
            case 'i': o.min_intron_length = toUnsigned(optarg); break;
            case 'I': o.max_intron_length = toUnsigned(optarg); break;
            case 'C': o.min_confidence_score = toFloat(optarg); break;
            case 's': {
                string su(optarg);
                std::transform(su.begin(), su.end(), su.begin(), ::toupper);
                if (su == "UN") o.strandness = UNSTRANDED;
                else if (su == "RF") o.strandness = RF_STRANDED;
                else if (su == "FR") o.strandness = FR_STRANDED;
                else {
                    cerr << "Error: invalid strandness value '" << optarg
                         << "', expected one of 'UN', 'RF', or 'FR' (case insensitive)" << endl;
                    exit(1);
                }
                break;
            }
            case 'X':
                if (string(optarg) == "multi") o.excludes |= EXCLUDE_MULTI;
                else {
                    cerr << "Error: invalid exclude category '" << optarg << "', expected one of: 'multi'" << endl;
                    exit(1);
                }
                break;
            case 'g': o.genome_fa = optarg; break;
            case 'f': o.junction_filter = junction_filter_parse(optarg); break;
            case 'S': o.skip_missing_targets = true; break;
            case 'j': o.junction_bed = optarg; break;
            case 'n': o.intron_bed = optarg; break;
            case 'b': o.intron_bed6 = optarg; break;
            case 'c': o.intron_call_tsv = optarg; break;
            case 'D': o.debug_trace_tsv = optarg; break;
            case 'p': o.bam_pass_through = optarg; break;
            case OPT_XS: o.set_XS_strand_tag = true; break;
            case OPT_TS: o.set_TS_strand_tag = true; break;
            case OPT_DEVICES: o.devices = parse_devices(optarg); break;
            case 'h': cerr << usage_msg; exit(0);
            case 'v': cerr << "intronProspector 1.4.0 (MI355X junction-extraction build)" << endl; exit(0);
            case '?':
            default: cmd_error("bug: option not handled");
        }
    }
    for (int i = optind; i < argc; ++i) o.bam_files.push_back(argv[i]);
    if (o.bam_files.empty()) o.bam_files.push_back("/dev/stdin");
    if (o.bam_files.size() > 1 && !o.bam_pass_through.empty())                    // :269-271
        cmd_error("can only have one input BAM with --bam-pass-through");
    if (o.junction_filter == CANON_SJ_FILTER && o.genome_fa.empty()) cmd_error("--sj-filter=canon requires --genome_fa");
    if (o.junction_filter == NULL_SJ_FILTER && !o.genome_fa.empty()) o.junction_filter = CANON_SJ_FILTER;
    if (!o.allow_anchor_indels) o.max_anchor_indel_size = 0;
    if ((o.set_XS_strand_tag || o.set_TS_strand_tag) && (o.bam_pass_through.empty() || o.genome_fa.empty()))   // :294-298
        cmd_error("--set-XS-strand-tag and --set-TS-strand-tag require --pass-through and --genome-fasta");
}

struct OutputFiles {
    std::unique_ptr<std::ostream> junction_bed_fh, intron_bed_fh, intron_bed6_fh, intron_call_fh;
    static std::unique_ptr<std::ostream> open(const string &fname) {
        return fname.empty() ? nullptr : std::unique_ptr<std::ostream>(new AutoGzipOutput(fname));
    }
    explicit OutputFiles(const Options &o)
        : junction_bed_fh(open(o.junction_bed)), intron_bed_fh(open(o.intron_bed)),
          intron_bed6_fh(open(o.intron_bed6)), intron_call_fh(open(o.intron_call_tsv)) {
        if (intron_call_fh) print_junction_call_header(*intron_call_fh);       // :317-319
    }
};

// src/intronProspector.cc:330-351
static void write_junctions(JunctionVector &juncs, float min_confidence_score, OutputFiles &output) {
    const bool introns = output.intron_bed_fh || output.intron_bed6_fh || output.intron_call_fh;
    // The anchor file and the intron files are sorted and formatted side by side when both are asked for (0.15 s
    // each for 200 000 junctions): the first sort's input is the table's order either way, and the intron order has
    // no ties (junctions.cc:108-159), so it does not depend on what the anchor sort left behind.
    std::future<void> anchors;
    if (output.junction_bed_fh) {
        if (introns && juncs.size() >= 10000) {
            std::shared_ptr<JunctionVector> copy(new JunctionVector(juncs));
            std::ostream *fh = output.junction_bed_fh.get();
            anchors = std::async(std::launch::async, [copy, min_confidence_score, fh] {
                copy->sort_by_anchors();
                print_anchor_bed(*copy, min_confidence_score, *fh);
            });
        } else {
            juncs.sort_by_anchors();
            print_anchor_bed(juncs, min_confidence_score, *output.junction_bed_fh);
        }
    }
    struct Join { std::future<void> &f; ~Join() { if (f.valid()) f.wait(); } } join{anchors};
    if (introns) {
        juncs.sort_by_introns();
        if (output.intron_bed_fh) print_intron_bed(juncs, min_confidence_score, 9, *output.intron_bed_fh);
        if (output.intron_bed6_fh) print_intron_bed(juncs, min_confidence_score, 6, *output.intron_bed6_fh);
        if (output.intron_call_fh) print_intron_call_tsv(juncs, min_confidence_score, *output.intron_call_fh);
    }
    if (anchors.valid()) anchors.get();              // (an exception of the anchor writer surfaces here)
}

static void output_junctions(JunctionsExtractor &extractor, float min_confidence_score, OutputFiles &output) {
    JunctionVector juncs = extractor.get_junctions();
    write_junctions(juncs, min_confidence_score, output);
}

static void extract_junctions(const Options &o) {
    std::unique_ptr<Genome> genome;
    if (!o.genome_fa.empty()) genome.reset(new Genome(o.genome_fa));
    // src/intronProspector.cc:408-415,427
    std::unique_ptr<std::ostream> trace_fh;
    if (!o.debug_trace_tsv.empty()) trace_fh.reset(new AutoGzipOutput(o.debug_trace_tsv));
    JunctionsExtractor extractor(o.min_anchor_length, o.max_anchor_indel_size, o.min_intron_length,
                                 o.max_intron_length, o.junction_filter, o.strandness, o.excludes, genome.get(),
                                 o.skip_missing_targets, 0, &std::cerr, trace_fh.get(), o.set_XS_strand_tag,
                                 o.set_TS_strand_tag, o.devices);
    OutputFiles output(o);
    if (o.bam_files.size() > 1) {                       // :391-401
        for (const string &bam : o.bam_files) {
            extractor.open(bam);
            extractor.identify_junctions_for_bam();
            extractor.close();
        }
        output_junctions(extractor, o.min_confidence_score, output);
    } else if (o.unsorted) {                            // :375-387
        extractor.open(o.bam_files[0]);
        if (!o.bam_pass_through.empty()) extractor.open_pass_through(o.bam_pass_through);
        extractor.identify_junctions_for_bam();
        extractor.copy_unaligned_reads();
        extractor.close();
        output_junctions(extractor, o.min_confidence_score, output);
    } else {                                            // :356-371
        extractor.open(o.bam_files[0]);
        if (!o.bam_pass_through.empty()) extractor.open_pass_through(o.bam_pass_through);
        // a target's rows are written (sorted, formatted: 0.3 s for 200 000 junctions) on a thread of their own
        // while the next target's alignments are read; rows leave in target order, one writer at a time, and a
        // writer still running when a later target throws finishes first (the future's destructor waits): what the
        // reference had flushed for the earlier targets stays written
        std::future<void> writer;
        for (int target_index = 0; target_index < extractor.get_num_targets(); target_index++) {
            extractor.identify_junctions_for_target(target_index);
            std::shared_ptr<JunctionVector> juncs(new JunctionVector(extractor.get_junctions()));
            extractor.clear();
            if (writer.valid()) writer.get();
            const float conf = o.min_confidence_score;
            OutputFiles *out = &output;
            writer = std::async(std::launch::async, [juncs, conf, out] { write_junctions(*juncs, conf, *out); });
        }
        if (writer.valid()) writer.get();
        extractor.copy_unaligned_reads();
        extractor.close();
    }
}

int main(int argc, char *argv[]) {
    const std::chrono::steady_clock::time_point t_main = std::chrono::steady_clock::now();
    Options opts;
    try {
        parse_options(argc, argv, opts);
    } catch (const std::exception &ex) {
        cerr << "Error parsing command line: " << ex.what() << endl;
        return 1;
    }
    try {
        extract_junctions(opts);
    } catch (const std::exception &e) {
        cmd_error(e.what());
    }
    {
        const char *e = getenv("IPX_TIMELINE");        // main()'s own span: what the wall clock adds to it is loading the program
        if (e && e[0] == '1')
            cerr << "ipx_timeline: main() took " << std::chrono::duration<double>(std::chrono::steady_clock::now() - t_main).count() << " s" << endl;
    }
    // Every output is written and closed (extract_junctions' objects are gone).  What a plain return would still do is
    // unload the HIP runtime -- measured: the command's wall time is 0.10 s longer than main() -- which no caller waits for
    // anything from: leave at once.
    // (IPX_FULL_EXIT=1, a sanitizer or a coverage build: the ordinary way out, with atexit handlers and static destructors --
    // LeakSanitizer, gcov and a profiler's tool library report from there.  An output stream that failed is an error either
    // way.)
    std::cout.flush();
    std::cerr.flush();
    const bool flushed = fflush(nullptr) == 0 && std::cout.good();
    if (!flushed) {
        fprintf(stderr, "Error: writing to standard output failed\n");
        return 1;
    }
#if defined(__SANITIZE_ADDRESS__) || defined(IPX_COVERAGE)
    return 0;
#else
#if defined(__has_feature)
#if __has_feature(address_sanitizer)
    return 0;
#endif
#endif
    {
        const char *e = getenv("IPX_FULL_EXIT");
        if (e && e[0] == '1') return 0;
    }
    _exit(0);
#endif
}
