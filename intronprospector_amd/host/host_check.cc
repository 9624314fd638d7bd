// host_check.cc -- small driver that exposes the host-side pieces (alignment decoding,
// ordering, confidence, writers) to the CPU-only tests; it never touches the GPU.
//   ipx_host_check dump-aln FILE
//       one line per alignment: tid pos flag xs nh n_cigar w0,w1,...
//   ipx_host_check print-table PREFIX MIN_CONF < table.txt
//       table.txt: one junction per line in std::map order,
//         chrom start end anchor_start anchor_end strand splice_sites(- = none) ijunc c0 c1 c2 nbins cnt...
//       writes PREFIX.juncs.bed, PREFIX.introns.bed, PREFIX.introns.bed6, PREFIX.introns.tsv with
//       the output scheduling of src/intronProspector.cc:330-351
#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>

#include "aln_reader.hh"
#include "gz_ostream.hh"
#include "junctions.hh"

using namespace ipxhost;

static int dump_aln(const char *path) {
    AlnReader rd(path);
    for (size_t i = 0; i < rd.target_names().size(); ++i)
        printf("@SQ\t%s\t%lld\n", rd.target_names()[i].c_str(), (long long)rd.target_lengths()[i]);
    AlnRecord r;
    while (rd.read(r)) {
        printf("%d\t%lld\t%u\t%u\t%u\t%zu\t", r.tid, (long long)r.pos, r.flag, r.xs, r.nh, r.cigar.size());
        for (size_t k = 0; k < r.cigar.size(); ++k) printf(k ? ",%u" : "%u", r.cigar[k]);
        printf("\n");
    }
    return 0;
}

static int print_table(const std::string &prefix, float min_conf) {
    JunctionVector juncs;
    std::string line;
    while (std::getline(std::cin, line)) {
        if (line.empty()) continue;
        std::istringstream is(line);
        Junction j;
        std::string strand, ss;
        uint64_t nbins;
        is >> j.chrom >> j.intron_start >> j.intron_end >> j.anchor_start >> j.anchor_end >> strand >> ss >> j.ijunc >>
            j.read_counts[0] >> j.read_counts[1] >> j.read_counts[2] >> nbins;
        j.strand = strand[0];
        if (ss != "-") j.splice_sites = ss;
        std::vector<uint32_t> counts(nbins);
        for (auto &c : counts) is >> c;
        j.confidence = confidence_from_histogram(counts.data(), nbins, j.total_read_count());
        juncs.push_back(j);
    }
    AutoGzipOutput jb(prefix + ".juncs.bed"), ib(prefix + ".introns.bed"), ib6(prefix + ".introns.bed6"),
        tsv(prefix + ".introns.tsv");
    print_junction_call_header(tsv);
    juncs.sort_by_anchors();
    print_anchor_bed(juncs, min_conf, jb);
    juncs.sort_by_introns();
    print_intron_bed(juncs, min_conf, 9, ib);
    print_intron_bed(juncs, min_conf, 6, ib6);
    print_intron_call_tsv(juncs, min_conf, tsv);
    return 0;
}

int main(int argc, char **argv) {
    try {
        if (argc == 3 && std::string(argv[1]) == "dump-aln") return dump_aln(argv[2]);
        if (argc == 4 && std::string(argv[1]) == "print-table") return print_table(argv[2], strtof(argv[3], nullptr));
    } catch (const std::exception &e) {
        fprintf(stderr, "Error: %s\n", e.what());
        return 1;
    }
    fprintf(stderr, "usage: ipx_host_check dump-aln FILE | print-table PREFIX MIN_CONF < table.txt\n");
    return 2;
}
