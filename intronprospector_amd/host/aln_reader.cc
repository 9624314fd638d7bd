// aln_reader.cc -- see aln_reader.hh.
#include "aln_reader.hh"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

namespace ipxhost {

static uint8_t nh_class(long long v) { return v <= 0 ? 0 : (v == 1 ? 1 : 2); }

AlnReader::AlnReader(const std::string &path) : path_(path) {
    if (BgzfReader::is_bgzf(path)) {
        bgzf_.reset(new BgzfReader(path));
        unsigned char magic[4];
        if (bgzf_->read(magic, 4) == 4 && memcmp(magic, "BAM\1", 4) == 0) {
            bam_ = true;
            parse_bam_header();
            return;
        }
        bgzf_.reset();                      // bgzip'd SAM text: a multi-member gzip file to zlib
    }
    fh_ = gzopen(path.c_str(), "rb");       // transparent for plain text
    if (!fh_) throw std::runtime_error("Error opening BAM/SAM/CRAM file: " + path);
    gzbuffer(fh_, 1 << 20);
    unsigned char magic[4];
    int n = gzread(fh_, magic, 4);
    if (n == 4 && memcmp(magic, "BAM\1", 4) == 0) {
        bam_ = true;                        // BAM in plain gzip or uncompressed: single-threaded
        parse_bam_header();
    } else {
        if (n == 4 && memcmp(magic, "CRAM", 4) == 0)
            throw std::runtime_error("Error opening BAM/SAM/CRAM file: " + path + " (CRAM is not supported)");
        if (n > 0) pushback_.assign((const char *)magic, (size_t)n);     // a pipe cannot be rewound
        parse_sam_header();
    }
}

AlnReader::~AlnReader() {
    if (fh_) gzclose(fh_);
}

bool AlnReader::read(AlnRecord &rec) { return bam_ ? read_bam(rec) : read_sam(rec); }

// ---- SAM text ----------------------------------------------------------------
bool AlnReader::read_line() {
    line_.clear();
    if (!pushback_.empty()) {
        const size_t nl = pushback_.find('\n');
        if (nl != std::string::npos) {               // the probed bytes hold a whole (tiny) line
            line_.assign(pushback_, 0, nl);
            pushback_.erase(0, nl + 1);
            while (!line_.empty() && line_.back() == '\r') line_.pop_back();
            return true;
        }
        line_.swap(pushback_);
        pushback_.clear();
    }
    char chunk[65536];
    for (;;) {
        if (!gzgets(fh_, chunk, sizeof chunk)) return !line_.empty();
        size_t n = strlen(chunk);
        line_.append(chunk, n);
        if (n && chunk[n - 1] == '\n') break;
    }
    while (!line_.empty() && (line_.back() == '\n' || line_.back() == '\r')) line_.pop_back();
    return true;
}

void AlnReader::parse_sam_header() {
    while (read_line()) {
        if (line_.empty()) continue;
        if (line_[0] != '@') { have_line_ = true; break; }
        raw_header_ += line_;
        raw_header_ += '\n';
        if (line_.compare(0, 4, "@SQ\t") == 0) {
            std::string sn;
            int64_t ln = -1;
            size_t p = 4;
            while (p <= line_.size()) {
                size_t e = line_.find('\t', p);
                if (e == std::string::npos) e = line_.size();
                if (e - p > 3 && line_.compare(p, 3, "SN:") == 0) sn = line_.substr(p + 3, e - p - 3);
                if (e - p > 3 && line_.compare(p, 3, "LN:") == 0) ln = strtoll(line_.c_str() + p + 3, nullptr, 10);
                p = e + 1;
            }
            names_.push_back(sn);
            lengths_.push_back(ln);
        }
    }
}

bool AlnReader::read_sam(AlnRecord &rec) {
    do {
        if (have_line_) have_line_ = false;
        else if (!read_line()) return false;
    } while (line_.empty());
    const char *f[11];
    size_t flen[11];
    size_t p = 0;
    int nf = 0;
    while (nf < 11) {
        size_t e = line_.find('\t', p);
        if (e == std::string::npos) e = line_.size();
        f[nf] = line_.c_str() + p;
        flen[nf] = e - p;
        ++nf;
        if (e == line_.size()) { p = e; break; }
        p = e + 1;
    }
    if (nf < 11) throw std::runtime_error("Error reader BAM record: " + path_);
    if (keep_raw_) rec.raw = line_;
    if (keep_qname_) rec.qname.assign(f[0], flen[0]);
    rec.flag = (uint16_t)strtol(f[1], nullptr, 0);
    rec.tid = -1;
    if (!(flen[2] == 1 && f[2][0] == '*')) {
        for (size_t i = 0; i < names_.size(); ++i)
            if (names_[i].size() == flen[2] && memcmp(names_[i].data(), f[2], flen[2]) == 0) { rec.tid = (int32_t)i; break; }
    }
    rec.pos = strtoll(f[3], nullptr, 10) - 1;
    rec.cigar.clear();
    if (!(flen[5] == 1 && f[5][0] == '*')) {
        const char *c = f[5], *cend = f[5] + flen[5];
        while (c < cend) {
            char *e;
            unsigned long len = strtoul(c, &e, 10);
            static const char ops[] = "MIDNSHP=XB";
            const char *o = (e < cend && *e) ? strchr(ops, *e) : nullptr;
            if (e == c || !o) throw std::runtime_error("Error reader BAM record: " + path_);
            rec.cigar.push_back(((uint32_t)len << 4) | (uint32_t)(o - ops));
            c = e + 1;
        }
    }
    rec.cig = reinterpret_cast<const uint8_t *>(rec.cigar.data());
    rec.n_cig = (uint32_t)rec.cigar.size();
    rec.xs = 0;
    rec.nh = 0;
    bool have_xs = false, have_nh = false;
    while (p < line_.size()) {
        size_t e = line_.find('\t', p);
        if (e == std::string::npos) e = line_.size();
        const char *t = line_.c_str() + p;
        if (e - p >= 5 && t[2] == ':' && t[4] == ':') {
            if (t[0] == 'X' && t[1] == 'S' && !have_xs) {
                have_xs = true;
                char c = (t[3] == 'A' && e - p > 5) ? t[5] : 0;      // bam_aux2A: 0 unless type 'A'
                rec.xs = c ? (uint8_t)c : (uint8_t)'.';
            } else if (t[0] == 'N' && t[1] == 'H' && !have_nh) {
                have_nh = true;
                rec.nh = (t[3] == 'i') ? nh_class((int)strtoll(t + 5, nullptr, 10)) : 0;   // bam_aux2i: 0 unless integer
            }
        }
        p = e + 1;
    }
    return true;
}

// ---- BAM -----------------------------------------------------------------------
size_t AlnReader::read_some(void *dst, size_t n) {
    if (bgzf_) return bgzf_->read(dst, n);
    size_t got = 0;
    while (got < n) {
        int r = gzread(fh_, (char *)dst + got, (unsigned)std::min<size_t>(n - got, 1u << 30));
        if (r < 0) throw std::runtime_error("Error reader BAM record: " + path_);
        if (r == 0) break;
        got += (size_t)r;
    }
    return got;
}

void AlnReader::read_exact(void *dst, size_t n, const char *what) {
    if (read_some(dst, n) != n) throw std::runtime_error(std::string("Error reader BAM record: ") + path_ + " (" + what + ")");
}

static uint32_t le32(const uint8_t *p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }

void AlnReader::parse_bam_header() {
    uint8_t w[4];
    read_exact(w, 4, "l_text");
    raw_header_.append((const char *)w, 4);
    uint32_t l_text = le32(w);
    std::vector<char> text(l_text);
    if (l_text) read_exact(text.data(), l_text, "header text");
    raw_header_.append(text.data(), l_text);
    read_exact(w, 4, "n_ref");
    raw_header_.append((const char *)w, 4);
    uint32_t n_ref = le32(w);
    for (uint32_t i = 0; i < n_ref; ++i) {
        read_exact(w, 4, "l_name");
        raw_header_.append((const char *)w, 4);
        uint32_t l_name = le32(w);
        std::string nm(l_name, '\0');
        read_exact(&nm[0], l_name, "ref name");
        raw_header_.append(nm);
        if (!nm.empty() && nm.back() == '\0') nm.pop_back();
        read_exact(w, 4, "l_ref");
        raw_header_.append((const char *)w, 4);
        names_.push_back(nm);
        lengths_.push_back((int32_t)le32(w));
    }
}

void parse_bam_record(const uint8_t *b, uint32_t block, const std::string &path_, BamFields &rec) {
    if (block < 32) throw std::runtime_error("Error reader BAM record: " + path_);
    rec.tid = (int32_t)le32(b);
    rec.pos = (int32_t)le32(b + 4);
    const uint32_t l_read_name = b[8];
    const uint32_t n_cigar = b[12] | (b[13] << 8);
    rec.flag = (uint16_t)(b[14] | (b[15] << 8));
    const uint32_t l_seq = le32(b + 16);
    size_t off = 32;
    if (off + l_read_name > block) throw std::runtime_error("Error reader BAM record: " + path_);
    rec.l_read_name = l_read_name;
    off += l_read_name;
    if (off + 4ull * n_cigar > block) throw std::runtime_error("Error reader BAM record: " + path_);
    rec.cig = b + off;                       // little-endian words, as the device wants them
    rec.n_cig = n_cigar;
    off += 4ull * n_cigar;
    off += (l_seq + 1) / 2 + l_seq;
    if (off > block) throw std::runtime_error("Error reader BAM record: " + path_);
    rec.aux_off = (uint32_t)off;
    // aux fields
    rec.xs = 0;
    rec.nh = 0;
    bool have_xs = false, have_nh = false;
    const uint8_t *cg = nullptr;
    uint32_t cg_n = 0;
    while (off + 3 <= block) {
        const uint8_t *t = b + off;
        const char type = (char)t[2];
        size_t p = off + 3, sz = 0;
        long long ival = 0;
        bool is_int = false;
        // the value's size first, checked against the record, and only then the value (a record whose last tag is
        // cut short must not make this read past the block)
        switch (type) {
            case 'A': case 'c': case 'C': sz = 1; break;
            case 's': case 'S': sz = 2; break;
            case 'i': case 'I': case 'f': sz = 4; break;
            case 'd': sz = 8; break;
            case 'Z': case 'H': {
                size_t q = p;
                while (q < block && b[q]) ++q;
                sz = q - p + 1;
                break;
            }
            case 'B': {
                if (p + 5 > block) throw std::runtime_error("Error reader BAM record: " + path_);
                const char sub = (char)b[p];
                const uint32_t cnt = le32(b + p + 1);
                const size_t es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
                sz = 5 + es * (size_t)cnt;
                break;
            }
            default: throw std::runtime_error("Error reader BAM record: " + path_);
        }
        if (p + sz > block) throw std::runtime_error("Error reader BAM record: " + path_);
        switch (type) {
            case 'c': ival = (int8_t)b[p]; is_int = true; break;
            case 'C': ival = b[p]; is_int = true; break;
            case 's': ival = (int16_t)(b[p] | (b[p + 1] << 8)); is_int = true; break;
            case 'S': ival = (uint16_t)(b[p] | (b[p + 1] << 8)); is_int = true; break;
            case 'i': ival = (int32_t)le32(b + p); is_int = true; break;
            case 'I': ival = le32(b + p); is_int = true; break;
            case 'B':
                // bam_aux_get returns the FIRST tag of a name: a second CG array does not replace the first
                if (t[0] == 'C' && t[1] == 'G' && (char)b[p] == 'I' && !cg) { cg = b + p + 5; cg_n = le32(b + p + 1); }
                break;
            default: break;
        }
        if (t[0] == 'X' && t[1] == 'S' && !have_xs) {
            have_xs = true;
            char c = (type == 'A') ? (char)b[p] : 0;
            rec.xs = c ? (uint8_t)c : (uint8_t)'.';
        } else if (t[0] == 'N' && t[1] == 'H' && !have_nh) {
            have_nh = true;
            rec.nh = is_int ? nh_class((int)ival) : 0;
        }
        off = p + sz;
    }
    // long CIGARs live in the CG tag behind a "<l_seq>S<ref_len>N" placeholder (SAM spec 4.2.2)
    if (cg && n_cigar == 2 && (le32(rec.cig) & 0xf) == 4 && (le32(rec.cig) >> 4) == l_seq && (le32(rec.cig + 4) & 0xf) == 3) {
        rec.cig = cg;
        rec.n_cig = cg_n;
    }
}

bool AlnReader::read_bam(AlnRecord &rec) {
    if (pending_advance_) {                  // the previous record was parsed where it lay: release it now
        bgzf_->advance(pending_advance_);
        pending_advance_ = 0;
    }
    const uint8_t *b = nullptr;
    uint32_t block = 0;
    const uint8_t *pk = nullptr;
    if (bgzf_ && bgzf_->peek(4, &pk)) {
        block = le32(pk);
        if (block < 32) throw std::runtime_error("Error reader BAM record: " + path_);
        if (bgzf_->peek(4 + (size_t)block, &pk)) {       // the whole record lies in this block: no copy
            b = pk + 4;
            pending_advance_ = 4 + (size_t)block;
        }
    }
    if (!b) {
        uint8_t w[4];
        const size_t r = read_some(w, 4);
        if (r == 0) return false;
        if (r != 4) throw std::runtime_error("Error reader BAM record: " + path_);
        block = le32(w);
        if (block < 32) throw std::runtime_error("Error reader BAM record: " + path_);
        buf_.resize(block);
        read_exact(buf_.data(), block, "record");
        b = buf_.data();
    }
    BamFields f;
    parse_bam_record(b, block, path_, f);
    rec.tid = f.tid;
    rec.pos = f.pos;
    rec.flag = f.flag;
    rec.xs = f.xs;
    rec.nh = f.nh;
    rec.cig = f.cig;
    rec.n_cig = f.n_cig;
    if (keep_qname_) rec.qname.assign((const char *)b + 32, f.l_read_name ? f.l_read_name - 1 : 0);
    if (keep_raw_) {
        rec.raw.assign((const char *)b, block);
        rec.aux_off = f.aux_off;
    }
    return true;
}

}  // namespace ipxhost
