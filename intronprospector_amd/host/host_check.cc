// host_check.cc -- small driver that exposes the host-side pieces (alignment decoding,
// ordering, confidence, writers) to the CPU-only tests; it never touches the GPU.
//   ipx_host_check dump-aln FILE
//   ipx_host_check dump-feed FILE [by-target|all [CAP_ALN [CAP_WORDS]]]   the same through host/bam_feeder.cc
//       one line per alignment: tid pos flag xs nh n_cigar w0,w1,...
//   ipx_host_check print-table PREFIX MIN_CONF < table.txt
//       table.txt: one junction per line in std::map order,
//         chrom start end anchor_start anchor_end strand splice_sites(- = none) ijunc c0 c1 c2 nbins cnt...
//       writes PREFIX.juncs.bed, PREFIX.introns.bed, PREFIX.introns.bed6, PREFIX.introns.tsv with
//       the output scheduling of src/intronProspector.cc:330-351
//   ipx_host_check make-bam N OUT.bam
//       N synthetic 100 bp spliced alignments (aM nN bM over 200 000 junction sites in coordinate order,
//       sequence, qualities, NH/XS tags) as a BAM file, to time the decoder and the command line on
//   ipx_host_check copy-aln IN OUT [XS|TS +|-]...
//       pass-through writer: every record of IN to OUT in IN's container (SAM text / BAM), the given
//       strand tags set on every record the way update_strand_tag does
//   ipx_host_check bench-decode FILE [THREADS]
//       decodes every record; prints records, seconds, records/s and compressed MB/s
//   ipx_host_check bench-inflate FILE [THREADS]
//       BGZF inflate only (no record parsing)
#include <chrono>
#include <cstdio>
#include "fast_inflate.hh"
#include <zlib.h>
#include <memory>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "aln_reader.hh"
#include "bam_feeder.hh"
#include "gz_ostream.hh"
#include "junctions.hh"
#include "pass_through.hh"

using namespace ipxhost;

static int dump_aln(const char *path) {
    AlnReader rd(path);
    for (size_t i = 0; i < rd.target_names().size(); ++i)
        printf("@SQ\t%s\t%lld\n", rd.target_names()[i].c_str(), (long long)rd.target_lengths()[i]);
    AlnRecord r;
    while (rd.read(r)) {
        printf("%d\t%lld\t%u\t%u\t%u\t%zu\t", r.tid, (long long)r.pos, r.flag, r.xs, r.nh, (size_t)r.n_cig);
        for (size_t k = 0; k < r.n_cig; ++k) printf(k ? ",%u" : "%u", r.cigar_word(k));
        printf("\n");
    }
    return 0;
}

// the same listing through the block-parallel feeder (host/bam_feeder.cc): target by target like the sorted-mode
// loop when by_target, else everything at once; small batches so that spans get cut in every possible place
static int dump_feed(const char *path, bool by_target, size_t cap_aln, size_t cap_words) {
    AlnReader rd(path);
    for (size_t i = 0; i < rd.target_names().size(); ++i)
        printf("@SQ\t%s\t%lld\n", rd.target_names()[i].c_str(), (long long)rd.target_lengths()[i]);
    if (!rd.bgzf()) { fprintf(stderr, "not a BGZF-compressed BAM file\n"); return 1; }
    std::vector<int32_t> canon(rd.target_names().size());
    for (size_t i = 0; i < canon.size(); ++i) canon[i] = (int32_t)i;
    // (the arrays before the feeder: its workers write into them until its destructor has joined them)
    std::vector<int32_t> tid(cap_aln), pos(cap_aln);
    std::vector<uint16_t> flag(cap_aln);
    std::vector<uint8_t> xs(cap_aln), nh(cap_aln);
    std::vector<uint32_t> off(cap_aln), nc(cap_aln), cigar(cap_words);
    BamFeeder feeder(*rd.bgzf(), path, canon);
    size_t n = 0, nw = 0;
    auto flush = [&]() {
        feeder.wait_all();
        for (size_t i = 0; i < n; ++i) {
            printf("%d\t%lld\t%u\t%u\t%u\t%zu\t", tid[i], (long long)pos[i], flag[i], xs[i], nh[i], (size_t)nc[i]);
            for (size_t k = 0; k < nc[i]; ++k) printf(k ? ",%u" : "%u", cigar[off[i] + k]);
            printf("\n");
        }
        n = nw = 0;
    };
    const int n_targets = (int)canon.size();
    for (int t = by_target ? 0 : -1; t < (by_target ? n_targets : 0); ++t) {
        for (;;) {
            BamFeeder::Span sp;
            int32_t other = 0;
            const BamFeeder::Status st = feeder.next_span(by_target ? t : BamFeeder::ANY_TARGET, cap_aln - n, cap_words - nw, sp, &other);
            if (st == BamFeeder::SPAN) {
                const BamFeeder::Dest d{tid.data(), pos.data(), flag.data(), xs.data(), nh.data(), off.data(), nc.data(), cigar.data()};
                const size_t rn = sp.n_records, rw = sp.n_words;
                feeder.dispatch(std::move(sp), d, n, nw);
                n += rn;
                nw += rw;
            } else if (st == BamFeeder::FULL) {
                if (n == 0) { cap_words *= 4; cigar.resize(cap_words); }
                flush();
            } else {
                break;
            }
        }
        flush();
        if (by_target) printf("#target %d done\n", t);
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

// ---- synthetic BAM (SAM spec 4.2), BGZF blocks of <= 60000 bytes at zlib level 6
static void put32(std::vector<uint8_t> &v, uint32_t x) { for (int i = 0; i < 4; ++i) v.push_back((uint8_t)(x >> (8 * i))); }
static void put16(std::vector<uint8_t> &v, uint32_t x) { v.push_back((uint8_t)x); v.push_back((uint8_t)(x >> 8)); }

static void write_bgzf_block(FILE *f, const uint8_t *data, size_t n) {
    uint8_t out[70000];
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    deflateInit2(&zs, 6, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
    zs.next_in = const_cast<uint8_t *>(data);
    zs.avail_in = (uInt)n;
    zs.next_out = out + 18;
    zs.avail_out = sizeof out - 18 - 8;
    deflate(&zs, Z_FINISH);
    const size_t clen = zs.total_out;
    deflateEnd(&zs);
    const uint8_t hdr[16] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0};
    memcpy(out, hdr, 16);
    const size_t bsize = clen + 18 + 8 - 1;
    out[16] = (uint8_t)bsize;
    out[17] = (uint8_t)(bsize >> 8);
    const uint32_t crc = (uint32_t)crc32(crc32(0, nullptr, 0), data, (uInt)n);
    uint8_t *ft = out + 18 + clen;
    for (int i = 0; i < 4; ++i) { ft[i] = (uint8_t)(crc >> (8 * i)); ft[4 + i] = (uint8_t)((uint32_t)n >> (8 * i)); }
    fwrite(out, 1, 18 + clen + 8, f);
}

static int make_bam(uint64_t n, const char *path) {
    FILE *f = fopen(path, "wb");
    if (!f) { perror(path); return 1; }
    std::vector<uint8_t> buf;
    auto flush = [&](bool all) {
        size_t off = 0;
        while (buf.size() - off >= 60000 || (all && off < buf.size())) {
            const size_t k = std::min<size_t>(60000, buf.size() - off);
            write_bgzf_block(f, buf.data() + off, k);
            off += k;
        }
        buf.erase(buf.begin(), buf.begin() + off);
    };
    const char *text = "@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:chr1\tLN:200000000\n";
    buf.insert(buf.end(), {'B', 'A', 'M', 1});
    put32(buf, (uint32_t)strlen(text));
    buf.insert(buf.end(), text, text + strlen(text));
    put32(buf, 1);
    put32(buf, 5);
    buf.insert(buf.end(), {'c', 'h', 'r', '1', 0});
    put32(buf, 200000000);
    uint64_t rng = 88172645463325252ull;
    auto next = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
    for (uint64_t i = 0; i < n; ++i) {
        char name[32];
        const int ln = snprintf(name, sizeof name, "read%llu", (unsigned long long)i) + 1;
        // 200 000 junction sites, visited in coordinate order; a site's intron length is a function of the site
        const uint64_t n_sites = 200000, site = i * n_sites / (n ? n : 1);
        const uint32_t a = 10 + (uint32_t)(next() % 80), ilen = 70 + (uint32_t)((site * 2654435761ull) % 20000), b = 100 - a;
        const uint32_t pos = (uint32_t)(site * 900 + 1000 - a);
        std::vector<uint8_t> r;
        put32(r, 0); put32(r, pos);
        r.push_back((uint8_t)ln); r.push_back(60); put16(r, 4681);
        put16(r, 3); put16(r, (next() & 1) ? 16 : 0); put32(r, 100);
        put32(r, 0xffffffffu); put32(r, 0xffffffffu); put32(r, 0);
        r.insert(r.end(), name, name + ln);
        put32(r, a << 4); put32(r, (ilen << 4) | 3); put32(r, b << 4);
        for (int k = 0; k < 50; ++k) { const uint64_t x = next(); r.push_back((uint8_t)(((1u << (x & 3)) << 4) | (1u << ((x >> 2) & 3)))); }
        for (int k = 0; k < 100; ++k) r.push_back((uint8_t)(20 + next() % 21));
        r.insert(r.end(), {'N', 'H', 'C', 1, 'X', 'S', 'A', (uint8_t)((next() & 1) ? '+' : '-')});
        put32(buf, (uint32_t)r.size());
        buf.insert(buf.end(), r.begin(), r.end());
        if (buf.size() >= (1u << 20)) flush(false);
    }
    flush(true);
    write_bgzf_block(f, nullptr, 0);
    fclose(f);
    return 0;
}

static int copy_aln(int argc, char **argv) {
    AlnReader rd(argv[2]);
    PassThroughWriter out(argv[3], rd);
    rd.keep_raw(true);
    AlnRecord r;
    while (rd.read(r)) {
        if (argc == 4) { out.write(r, 0, false, false); continue; }
        // one writer call per requested tag, so that "XS + TS -" can differ; the writer sets both from one count
        bool xs = false, ts = false;
        int orient = 0;
        for (int i = 4; i + 1 < argc; i += 2) {
            if (std::string(argv[i]) == "XS") xs = true; else ts = true;
            orient = argv[i + 1][0] == '+' ? 1 : -1;
        }
        out.write(r, orient, xs, ts);
    }
    out.close();
    return 0;
}

static int bench_decode(const char *path, const char *threads) {
    if (threads) setenv("IPX_DECODE_THREADS", threads, 1);
    const auto t0 = std::chrono::steady_clock::now();
    AlnReader rd(path);
    AlnRecord r;
    uint64_t n = 0, words = 0, sum = 0;
    while (rd.read(r)) {
        ++n;
        words += r.n_cig;
        sum += (uint64_t)r.pos + r.flag + r.nh + r.xs;
    }
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::ifstream in(path, std::ios::binary | std::ios::ate);
    const double mb = (double)in.tellg() / 1e6;
    printf("{\"records\": %llu, \"cigar_words\": %llu, \"seconds\": %.4f, \"records_per_s\": %.0f, \"compressed_MB_per_s\": %.1f, "
           "\"threads\": %u, \"checksum\": %llu}\n",
           (unsigned long long)n, (unsigned long long)words, s, n / s, mb / s, rd.decode_threads(), (unsigned long long)sum);
    return 0;
}

// inflate only: what the BGZF stage delivers when nobody parses the records
static int bench_inflate(const char *path, const char *threads) {
    const unsigned nt = threads ? (unsigned)atoi(threads) : 0;
    const auto t0 = std::chrono::steady_clock::now();
    BgzfReader rd(path, nt);
    std::vector<uint8_t> buf(1 << 20);
    uint64_t bytes = 0, sum = 0;
    for (size_t got; (got = rd.read(buf.data(), buf.size())) != 0;) {
        bytes += got;
        sum += buf[0] + buf[got - 1];
    }
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("{\"inflated_MB\": %.1f, \"seconds\": %.4f, \"inflated_MB_per_s\": %.1f, \"compressed_MB_per_s\": %.1f, \"threads\": %u, "
           "\"checksum\": %llu}\n", bytes / 1e6, s, bytes / 1e6 / s, rd.compressed_bytes() / 1e6 / s, rd.threads(), (unsigned long long)sum);
    return 0;
}

// check-inflate FILE [DAMAGE]: FILE holds raw DEFLATE streams as [u32 compressed size][u32 inflated size][bytes];
// every stream through host/fast_inflate.cc and through zlib: same bytes, same CRC (fast_crc32 against crc32).
// DAMAGE > 0: each stream also DAMAGE times with a few bytes changed -- whatever the fast decoder makes of those,
// it must stay inside its buffers (the buffers are exact-size heap blocks: run under AddressSanitizer) and it
// must not produce anything where zlib reports an error without reporting one itself... except that a damaged
// stream may still be a valid one: then both must agree on the bytes.
static int check_inflate(const char *path, int damage) {
    FILE *f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); return 2; }
    uint64_t n_streams = 0, n_bytes = 0, n_damaged = 0, n_damaged_valid = 0;
    double t_fast = 0, t_zlib = 0;
    uint32_t hdr[2];
    uint64_t rng = 88172645463325252ull;
    auto rnd = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
    while (fread(hdr, 4, 2, f) == 2) {
        std::vector<uint8_t> in(hdr[0]);
        if (hdr[0] && fread(in.data(), 1, hdr[0], f) != hdr[0]) { fprintf(stderr, "truncated\n"); return 2; }
        auto zlib_inflate = [&](const std::vector<uint8_t> &src, std::vector<uint8_t> &dst) -> long {
            z_stream zs;
            memset(&zs, 0, sizeof zs);
            if (inflateInit2(&zs, -15) != Z_OK) return -2;
            zs.next_in = const_cast<uint8_t *>(src.data());
            zs.avail_in = (uInt)src.size();
            uint8_t spare;                                   // (zlib wants somewhere to write even when nothing is to come)
            zs.next_out = dst.empty() ? &spare : dst.data();
            zs.avail_out = dst.empty() ? 1 : (uInt)dst.size();
            const int rc = inflate(&zs, Z_FINISH);
            const long got = (long)((dst.empty() ? 1 : dst.size()) - zs.avail_out);
            inflateEnd(&zs);
            return rc == Z_STREAM_END ? got : -1;
        };
        // exact-size heap blocks, so that a sanitizer sees any step outside
        std::unique_ptr<uint8_t[]> in_exact(new uint8_t[hdr[0] ? hdr[0] : 1]);
        if (hdr[0]) memcpy(in_exact.get(), in.data(), hdr[0]);
        std::unique_ptr<uint8_t[]> out_fast(new uint8_t[hdr[1] ? hdr[1] : 1]);
        std::vector<uint8_t> out_z(hdr[1]);
        auto t0 = std::chrono::steady_clock::now();
        const long a = ipxhost::fast_inflate(in_exact.get(), hdr[0], out_fast.get(), hdr[1]);
        auto t1 = std::chrono::steady_clock::now();
        const long b = zlib_inflate(in, out_z);
        auto t2 = std::chrono::steady_clock::now();
        t_fast += std::chrono::duration<double>(t1 - t0).count();
        t_zlib += std::chrono::duration<double>(t2 - t1).count();
        if (a != (long)hdr[1] || b != (long)hdr[1] || memcmp(out_fast.get(), out_z.data(), hdr[1]) != 0) {
            fprintf(stderr, "stream %llu: fast_inflate %ld, zlib %ld, expected %u\n", (unsigned long long)n_streams, a, b, hdr[1]);
            return 1;
        }
        if (ipxhost::fast_crc32(0, out_z.data(), hdr[1]) != (uint32_t)crc32(crc32(0, nullptr, 0), out_z.data(), hdr[1])) {
            fprintf(stderr, "stream %llu: CRC differs\n", (unsigned long long)n_streams);
            return 1;
        }
        for (int d = 0; d < damage && hdr[0]; ++d) {
            std::vector<uint8_t> bad(in);
            const int n_hits = 1 + (int)(rnd() % 3);
            for (int k = 0; k < n_hits; ++k) bad[rnd() % bad.size()] ^= (uint8_t)(1u << (rnd() % 8));
            if (rnd() % 8 == 0) bad.resize(rnd() % bad.size() + 1);            // cut short
            std::unique_ptr<uint8_t[]> bad_exact(new uint8_t[bad.size()]);
            if (!bad.empty()) memcpy(bad_exact.get(), bad.data(), bad.size());
            const long fa = ipxhost::fast_inflate(bad_exact.get(), bad.size(), out_fast.get(), hdr[1]);
            const long zb = zlib_inflate(bad, out_z);
            ++n_damaged;
            if (fa >= 0) {
                // accepted: zlib must have accepted the same bytes (zlib also rejects a few things this decoder
                // lets pass -- incomplete code sets -- so only the converse is an error)
                if (zb >= 0 && (fa != zb || memcmp(out_fast.get(), out_z.data(), (size_t)fa) != 0)) {
                    fprintf(stderr, "stream %llu damaged: both accept, different output\n", (unsigned long long)n_streams);
                    return 1;
                }
                ++n_damaged_valid;
            } else if (zb >= 0 && zb == (long)hdr[1]) {
                // zlib inflated it to the full size and we refused: only allowed if zlib stopped before the input's end
                // is not knowable here; count it as a disagreement
                fprintf(stderr, "stream %llu damaged: zlib accepts (%ld bytes), fast_inflate refuses\n", (unsigned long long)n_streams, zb);
                return 1;
            }
        }
        ++n_streams;
        n_bytes += hdr[1];
    }
    fclose(f);
    printf("{\"streams\": %llu, \"bytes\": %llu, \"fast_MB_per_s\": %.1f, \"zlib_MB_per_s\": %.1f, \"damaged\": %llu, \"damaged_still_valid\": %llu}\n",
           (unsigned long long)n_streams, (unsigned long long)n_bytes, n_bytes / 1e6 / (t_fast > 0 ? t_fast : 1e-9),
           n_bytes / 1e6 / (t_zlib > 0 ? t_zlib : 1e-9), (unsigned long long)n_damaged, (unsigned long long)n_damaged_valid);
    return 0;
}

int main(int argc, char **argv) {
    try {
        if (argc == 3 && std::string(argv[1]) == "dump-aln") return dump_aln(argv[2]);
        if (argc >= 3 && std::string(argv[1]) == "dump-feed")
            return dump_feed(argv[2], argc > 3 && std::string(argv[3]) == "by-target", argc > 4 ? strtoull(argv[4], nullptr, 10) : 1000,
                             argc > 5 ? strtoull(argv[5], nullptr, 10) : 6000);
        if (argc == 4 && std::string(argv[1]) == "print-table") return print_table(argv[2], strtof(argv[3], nullptr));
        if (argc >= 4 && std::string(argv[1]) == "copy-aln") return copy_aln(argc, argv);
        if (argc == 4 && std::string(argv[1]) == "make-bam") return make_bam(strtoull(argv[2], nullptr, 10), argv[3]);
        if ((argc == 3 || argc == 4) && std::string(argv[1]) == "bench-decode") return bench_decode(argv[2], argc == 4 ? argv[3] : nullptr);
        if (argc >= 3 && std::string(argv[1]) == "check-inflate") return check_inflate(argv[2], argc >= 4 ? atoi(argv[3]) : 0);
        if ((argc == 3 || argc == 4) && std::string(argv[1]) == "bench-inflate") return bench_inflate(argv[2], argc == 4 ? argv[3] : nullptr);
    } catch (const std::exception &e) {
        fprintf(stderr, "Error: %s\n", e.what());
        return 1;
    }
    fprintf(stderr, "usage: ipx_host_check dump-aln FILE | print-table PREFIX MIN_CONF < table.txt | make-bam N OUT | bench-decode FILE [THREADS] | bench-inflate FILE [THREADS]\n");
    return 2;
}
