// genome.cc -- see genome.hh.
#include "genome.hh"

#include <cctype>
#include <cstdio>
#include <cstring>
#include <fcntl.h>
#include <stdexcept>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace ipxhost {

Genome::Genome(const std::string &genome_fa) {
    const std::string err = "can't open genome FASTA file or associated index: " + genome_fa;
    FILE *fi = fopen((genome_fa + ".fai").c_str(), "r");
    if (!fi) throw std::runtime_error(err);
    char line[8192];
    while (fgets(line, sizeof line, fi)) {
        char name[4096];
        long long len, off;
        int lb, ll;
        if (sscanf(line, "%4095[^\t]\t%lld\t%lld\t%d\t%d", name, &len, &off, &lb, &ll) != 5) continue;
        Seq s;
        s.name = name;
        s.len = len; s.offset = off; s.line_blen = lb; s.line_len = ll;
        seqs_.push_back(std::move(s));
    }
    fclose(fi);
    fd_ = open(genome_fa.c_str(), O_RDONLY);
    if (fd_ < 0) throw std::runtime_error(err);
    struct stat st;
    if (fstat(fd_, &st) != 0) throw std::runtime_error(err);
    size_ = (size_t)st.st_size;
    if (size_) {
        void *m = mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd_, 0);
        if (m == MAP_FAILED) throw std::runtime_error(err);
        data_ = (const uint8_t *)m;
    }
    // pack: A0 C1 G2 T3 (either case), anything else sets the mask bit
    static uint8_t code[256];
    static bool init = false;
    if (!init) {
        memset(code, 4, sizeof code);
        code['A'] = code['a'] = 0; code['C'] = code['c'] = 1;
        code['G'] = code['g'] = 2; code['T'] = code['t'] = 3;
        init = true;
    }
    for (Seq &s : seqs_) {
        s.two.assign((size_t)(s.len + 3) / 4 + 4, 0);
        s.mask.assign((size_t)(s.len + 7) / 8 + 4, 0);
        if (s.line_blen <= 0) continue;
        int64_t p = s.offset, i = 0;
        while (i < s.len && p < (int64_t)size_) {
            const uint8_t c = data_[p++];
            if (!isgraph(c)) continue;           // faidx skips non-graphic bytes (line ends)
            const uint8_t v = code[c];
            if (v < 4) s.two[(size_t)(i >> 2)] |= (uint8_t)(v << ((i & 3) * 2));
            else s.mask[(size_t)(i >> 3)] |= (uint8_t)(1u << (i & 7));
            ++i;
        }
        // bases the file does not hold (truncated FASTA) count as non-ACGT
        for (; i < s.len; ++i) s.mask[(size_t)(i >> 3)] |= (uint8_t)(1u << (i & 7));
    }
}

Genome::~Genome() {
    if (data_) munmap((void *)data_, size_);
    if (fd_ >= 0) close(fd_);
}

const Genome::Seq *Genome::find(const std::string &chrom) const {
    for (const Seq &s : seqs_)
        if (s.name == chrom) return &s;
    return nullptr;
}

bool Genome::has_seq(const std::string &chrom) const { return find(chrom) != nullptr; }

char Genome::raw(const std::string &chrom, int64_t pos) const {
    const Seq *s = find(chrom);
    if (!s || pos < 0 || pos >= s->len || s->line_blen <= 0) return 'N';
    int64_t p = s->offset + pos / s->line_blen * s->line_len + pos % s->line_blen;
    while (p < (int64_t)size_) {
        const uint8_t c = data_[p++];
        if (isgraph(c)) return (char)c;
    }
    return 'N';
}

}  // namespace ipxhost
