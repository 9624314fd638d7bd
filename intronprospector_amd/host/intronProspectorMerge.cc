// intronProspectorMerge.cc -- merge intron-call TSVs of several intronProspector runs
// (SURVEY.md 8f-4).  CPU only, like the reference's tool: same options
// (src/intronProspectorMerge.cc:92-135), same row reading (:164-178), the merge rules of
// Junction::merge (src/junctions.cc:76-105), renumbering in anchor order (:198-202) and the
// output scheduling of :205-238 over this build's writers (host/junctions.cc).
#include <getopt.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>
#include <zlib.h>

#include "gz_ostream.hh"
#include "junctions.hh"

using namespace ipxhost;
using std::cerr;
using std::endl;
using std::string;

static const char *usage_msg =
    "intronProspectorMerge [options] [intron_calls_tsv ...]\n"
    "\n"
    "Merge intron calls (--intron-calls output) of several intronProspector runs; options:\n"
    "  -h, --help  -v, --version\n"
    "  -i, --intron-calls-files=file    file with one input TSV per line\n"
    "  -j, --junction-bed=file   -n, --intron-bed=file   -b, --intron-bed6=file   -c, --intron-calls=file\n"
    "  -f, --sj-filter=all|canon\n"
    "Inputs and outputs whose name ends in .gz are gzip-compressed.\n";

static void cmd_error(const string &msg) {
    cerr << "Error: " << msg << endl;
    cerr << "Use --help to get usage" << endl;
    exit(1);
}

enum JunctionFilter { NULL_SJ_FILTER = 0, ALL_SJ_FILTER = 1, CANON_SJ_FILTER = 2 };

// src/splice_juncs.cc:41-49
static JunctionFilter junction_filter_parse(const string &spec) {
    if (spec == "all") return ALL_SJ_FILTER;
    if (spec == "canon") return CANON_SJ_FILTER;
    throw std::invalid_argument("invalid splice junction filter '" + spec + "', expected 'all' or 'canon'");
}

// src/splice_juncs.cc:10-38: the three motifs in every mix of upper and lower case
static bool is_canonical(const string &s) {
    if (s.size() != 5) return false;
    string u(s);
    for (char &c : u) c = (char)toupper((unsigned char)c);
    return u == "GT/AG" || u == "GC/AG" || u == "AT/AC";
}

// src/splice_juncs.cc:52-61
static bool junction_filter_check(JunctionFilter f, const string &splice_sites) {
    if (splice_sites.empty()) return true;
    return f == CANON_SJ_FILTER ? is_canonical(splice_sites) : true;
}

// lines of a plain or gzip'd text file (src/AutoGzip.hh: by the .gz suffix; zlib reads both)
class LineReader {
public:
    explicit LineReader(const string &path) : fh_(gzopen(path.c_str(), "rb")) {}
    ~LineReader() { if (fh_) gzclose(fh_); }
    bool ok() const { return fh_ != nullptr; }
    bool getline(string &line) {
        line.clear();
        char chunk[65536];
        bool any = false;
        while (gzgets(fh_, chunk, sizeof chunk)) {
            any = true;
            const size_t n = strlen(chunk);
            line.append(chunk, n);
            if (n && chunk[n - 1] == '\n') { line.pop_back(); break; }
        }
        return any;
    }
private:
    gzFile fh_;
};

// src/type_ops.hh:18-43
static uint32_t string_to_uint(const string &s) {
    char *end;
    uint32_t n = (uint32_t)strtoul(s.c_str(), &end, 10);
    if (end == s.c_str() || *end != '\0') throw std::runtime_error("invalid integer \"" + s + "\"");
    return n;
}
static float string_to_float(const string &s) {
    char *end;
    float n = strtof(s.c_str(), &end);
    if (end == s.c_str() || *end != '\0') throw std::runtime_error("invalid float \"" + s + "\"");
    return n;
}

// src/tsv.hh: rows by column name
class Tsv {
public:
    explicit Tsv(const string &file) : file_(file), in_(file) {
        // a missing file reads as empty there too (gzopen failure -> fail bit -> "can't open")
        if (!in_.ok()) throw std::runtime_error("can't open " + file);
        string line;
        if (!in_.getline(line)) throw std::runtime_error("empty TSV file: " + file);
        const std::vector<string> words = split(line);
        n_cols_ = words.size();
        for (size_t i = 0; i < words.size(); ++i) cols_[words[i]] = i;
    }
    bool next_row() {
        string line;
        if (!in_.getline(line)) return false;
        row_ = split(line);
        if (row_.size() != n_cols_) {            // (the reference prints the expected count twice, tsv.hh:70)
            std::ostringstream m;
            m << "expected " << n_cols_ << " columns, got " << n_cols_ << ":" << file_;
            throw std::runtime_error(m.str());
        }
        return true;
    }
    const string &get_col(const string &name) const {
        auto it = cols_.find(name);
        if (it == cols_.end()) throw std::runtime_error("column '" + name + "' not found: " + file_);
        return row_[it->second];
    }
    uint32_t get_col_int(const string &name) const { return string_to_uint(get_col(name)); }
    float get_col_float(const string &name) const { return string_to_float(get_col(name)); }

private:
    string file_;
    LineReader in_;
    size_t n_cols_ = 0;
    std::map<string, size_t> cols_;
    std::vector<string> row_;
    static std::vector<string> split(const string &line) {
        std::vector<string> words;
        size_t prev = 0, cur;
        while ((cur = line.find('\t', prev)) != string::npos) {
            words.push_back(line.substr(prev, cur - prev));
            prev = cur + 1;
        }
        words.push_back(line.substr(prev));
        return words;
    }
};

struct Options {
    std::vector<string> input_calls_tsvs;
    string intron_calls_files, junction_bed, intron_bed, intron_bed6, intron_call_tsv;
    JunctionFilter junction_filter = NULL_SJ_FILTER;
};

static void parse_options(int argc, char *argv[], Options &o) {
    static const struct option long_options[] = {
        {"help", no_argument, NULL, 'h'}, {"version", no_argument, NULL, 'v'},
        {"intron-calls-files", required_argument, NULL, 'i'}, {"junction-bed", required_argument, NULL, 'j'},
        {"intron-bed", required_argument, NULL, 'n'}, {"intron-bed6", required_argument, NULL, 'b'},
        {"intron-calls", required_argument, NULL, 'c'}, {"sj-filter", required_argument, NULL, 'f'},
        {NULL, 0, NULL, 0}};
    int c;
    while ((c = getopt_long(argc, argv, "hvi:j:n:b:c:f:", long_options, NULL)) != -1) {
        switch (c) {
            case 'i': o.intron_calls_files = optarg; break;
            case 'j': o.junction_bed = optarg; break;
            case 'n': o.intron_bed = optarg; break;
            case 'b': o.intron_bed6 = optarg; break;
            case 'c': o.intron_call_tsv = optarg; break;
            case 'f': o.junction_filter = junction_filter_parse(optarg); break;
            case 'h': cerr << usage_msg; exit(0);
            case 'v': cerr << "intronProspector 1.4.0 (MI355X junction-extraction build)" << endl; exit(0);
            case '?':
            default: cmd_error("bug: option not handled");
        }
    }
    for (int i = optind; i < argc; ++i) o.input_calls_tsvs.push_back(argv[i]);
    if (!o.intron_calls_files.empty()) {                    // :137-143
        LineReader in(o.intron_calls_files);
        string line;
        while (in.ok() && in.getline(line)) o.input_calls_tsvs.push_back(line);
    }
    if (o.input_calls_tsvs.empty())
        cmd_error("a least one input call TSV required on the command list or with the --intron_call_files (-i) option");
}

// src/type_ops.hh:57-59 through int; src/junctions.cc:72-74
static string description(const Junction &j) {
    std::ostringstream s;
    s << j.chrom << ":" << (int)j.intron_start << "-" << (int)j.intron_end << "[" << j.strand << "/" << j.splice_sites << "]";
    return s.str();
}

// Junction::merge, src/junctions.cc:76-105
static void merge(Junction &self, const Junction &j1) {
    bool update_strand = false;
    if (j1.strand != '.' && self.strand == '.') {
        update_strand = true;
    } else if (j1.strand == '.' && self.strand != '.') {
        // leave as-is
    } else if (j1.strand != self.strand) {
        cerr << "intron records have different strand:" + description(j1) + " with " + description(self) << endl;
        return;
    } else if (j1.splice_sites != self.splice_sites) {
        throw std::runtime_error("intron records have different splice sites:" + description(j1) + " with " + description(self));
    }
    self.anchor_start = std::min(j1.anchor_start, self.anchor_start);
    self.anchor_end = std::max(j1.anchor_end, self.anchor_end);
    for (int c = 0; c < 3; ++c) self.read_counts[c] += j1.read_counts[c];
    self.confidence = std::max(j1.confidence, self.confidence);
    if (update_strand) {
        self.strand = j1.strand;
        self.splice_sites = j1.splice_sites;
    }
}

// src/junctions.hh:57-100: chromosome name, start, end
struct Key {
    string chrom;
    uint32_t start, end;
    bool operator<(const Key &o) const {
        if (chrom != o.chrom) return chrom < o.chrom;
        if (start != o.start) return start < o.start;
        return end < o.end;
    }
};

static void run(const Options &o) {
    std::map<Key, Junction> table;
    for (const string &file : o.input_calls_tsvs) {
        Tsv tsv(file);
        while (tsv.next_row()) {
            if (!junction_filter_check(o.junction_filter, tsv.get_col("splice_sites"))) continue;
            Junction j;                                      // read_junction, :164-178; ijunc is ignored
            j.intron_start = tsv.get_col_int("intron_start");
            j.intron_end = tsv.get_col_int("intron_end");
            j.chrom = tsv.get_col("chrom");
            j.anchor_start = j.intron_start - tsv.get_col_int("max_left_overhang");
            j.anchor_end = j.intron_end + tsv.get_col_int("max_right_overhang");
            j.strand = tsv.get_col("strand")[0];
            j.splice_sites = tsv.get_col("splice_sites");
            j.read_counts[SINGLE_MAPPED_READ] = tsv.get_col_int("uniq_mapped_count");
            j.read_counts[MULTI_MAPPED_READ] = tsv.get_col_int("multi_mapped_count");
            j.read_counts[UNSURE_READ] = tsv.get_col_int("unsure_mapped_count");
            j.confidence = tsv.get_col_float("confidence");
            const Key key{j.chrom, j.intron_start, j.intron_end};
            auto it = table.find(key);
            if (it == table.end()) table.emplace(key, j);
            else merge(it->second, j);
        }
    }
    JunctionVector juncs;
    for (auto &kv : table) juncs.push_back(kv.second);
    juncs.sort_by_anchors();
    for (size_t i = 0; i < juncs.size(); ++i) juncs[i].ijunc = (uint32_t)i;      // renumber_junctions
    if (!o.junction_bed.empty()) {
        AutoGzipOutput out(o.junction_bed);
        print_anchor_bed(juncs, 0.0, out);
    }
    if (!o.intron_bed.empty() || !o.intron_bed6.empty() || !o.intron_call_tsv.empty()) {
        juncs.sort_by_introns();
        if (!o.intron_bed.empty()) {
            AutoGzipOutput out(o.intron_bed);
            print_intron_bed(juncs, 0.0, 9, out);
        }
        if (!o.intron_bed6.empty()) {
            AutoGzipOutput out(o.intron_bed6);
            print_intron_bed(juncs, 0.0, 6, out);
        }
        if (!o.intron_call_tsv.empty()) {
            AutoGzipOutput out(o.intron_call_tsv);
            print_junction_call_header(out);
            print_intron_call_tsv(juncs, 0.0, out);
        }
    }
}

int main(int argc, char *argv[]) {
    Options opts;
    try {
        parse_options(argc, argv, opts);
    } catch (const std::exception &ex) {
        cerr << "Error parsing command line: " << ex.what() << endl;
        return 1;
    }
    try {
        run(opts);
    } catch (const std::exception &e) {
        cmd_error(e.what());
    }
    return 0;
}
