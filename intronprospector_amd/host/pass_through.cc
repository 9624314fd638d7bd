// pass_through.cc -- see pass_through.hh.
#include "pass_through.hh"

#include <cstring>
#include <stdexcept>
#include <zlib.h>

namespace ipxhost {

static const size_t BGZF_BLOCK = 0xff00;      // htslib's BGZF_BLOCK_SIZE: uncompressed bytes per block

// bam_aux_get + bam_aux_del + bam_aux_append on the text form: the first field "TG:..." goes,
// "TG:A:<strand>" is added at the end
void sam_set_strand_tag(std::string &line, const char *tag, char strand) {
    size_t p = 0;
    for (int f = 0; f < 11 && p != std::string::npos; ++f) {
        p = line.find('\t', p);
        if (p != std::string::npos) ++p;
    }
    while (p != std::string::npos && p < line.size()) {
        size_t e = line.find('\t', p);
        const size_t end = e == std::string::npos ? line.size() : e;
        if (end - p >= 3 && line[p] == tag[0] && line[p + 1] == tag[1] && line[p + 2] == ':') {
            if (e == std::string::npos) line.erase(p ? p - 1 : 0);       // last field: drop it and its tab
            else line.erase(p, e + 1 - p);
            break;
        }
        p = e == std::string::npos ? e : e + 1;
    }
    line += '\t';
    line += tag[0];
    line += tag[1];
    line += ":A:";
    line += strand;
}

// size of the value of an aux field of the given type starting at b[p]; 0 if it runs past the end
static size_t aux_value_size(const uint8_t *b, size_t p, size_t n, char type) {
    switch (type) {
        case 'A': case 'c': case 'C': return 1;
        case 's': case 'S': return 2;
        case 'i': case 'I': case 'f': return 4;
        case 'd': return 8;
        case 'Z': case 'H': {
            size_t q = p;
            while (q < n && b[q]) ++q;
            return q < n ? q - p + 1 : 0;
        }
        case 'B': {
            if (p + 5 > n) return 0;
            const char sub = (char)b[p];
            const uint32_t cnt = b[p + 1] | (b[p + 2] << 8) | (b[p + 3] << 16) | ((uint32_t)b[p + 4] << 24);
            const size_t es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
            return 5 + es * (size_t)cnt;
        }
        default: return 0;
    }
}

void bam_set_strand_tag(std::string &record, uint32_t aux_off, const char *tag, char strand) {
    const uint8_t *b = (const uint8_t *)record.data();
    size_t off = aux_off;
    while (off + 3 <= record.size()) {
        const size_t sz = aux_value_size(b, off + 3, record.size(), (char)b[off + 2]);
        if (sz == 0 || off + 3 + sz > record.size()) throw std::runtime_error("Error deleting read tag");
        if (b[off] == (uint8_t)tag[0] && b[off + 1] == (uint8_t)tag[1]) {
            record.erase(off, 3 + sz);
            break;
        }
        off += 3 + sz;
    }
    const char add[4] = {tag[0], tag[1], 'A', strand};
    record.append(add, 4);
}

PassThroughWriter::PassThroughWriter(const std::string &path, const AlnReader &reader) : path_(path), bam_(reader.is_bam()) {
    fp_ = fopen(path.c_str(), "wb");
    if (!fp_) throw std::runtime_error("Error opening BAM/SAM/CRAM file: " + path);
    setvbuf(fp_, nullptr, _IOFBF, 1 << 22);
    const std::string &hdr = reader.raw_header();
    if (bam_) {
        bgzf_append("BAM\1", 4);
        bgzf_append(hdr.data(), hdr.size());
        bgzf_flush_block(block_.size());        // htslib starts the records in a block of their own
    } else if (fwrite(hdr.data(), 1, hdr.size(), fp_) != hdr.size()) {
        throw std::runtime_error("Error writing SAM header: " + path);
    }
}

PassThroughWriter::~PassThroughWriter() {
    try {
        close();
    } catch (...) {
    }
}

void PassThroughWriter::bgzf_flush_block(size_t n) {
    uint8_t out[70000];
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (deflateInit2(&zs, Z_DEFAULT_COMPRESSION, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK)
        throw std::runtime_error("Error writing BAM record to pass-through file");
    zs.next_in = block_.data();
    zs.avail_in = (uInt)n;
    zs.next_out = out + 18;
    zs.avail_out = sizeof out - 18 - 8;
    const int rc = deflate(&zs, Z_FINISH);
    const size_t clen = zs.total_out;
    deflateEnd(&zs);
    if (rc != Z_STREAM_END) throw std::runtime_error("Error writing BAM record to pass-through file");
    static const uint8_t hdr[16] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0};
    memcpy(out, hdr, 16);
    const size_t bsize = clen + 18 + 8 - 1;
    out[16] = (uint8_t)bsize;
    out[17] = (uint8_t)(bsize >> 8);
    const uint32_t crc = (uint32_t)crc32(crc32(0, nullptr, 0), block_.data(), (uInt)n);
    uint8_t *ft = out + 18 + clen;
    for (int i = 0; i < 4; ++i) {
        ft[i] = (uint8_t)(crc >> (8 * i));
        ft[4 + i] = (uint8_t)((uint32_t)n >> (8 * i));
    }
    if (fwrite(out, 1, 18 + clen + 8, fp_) != 18 + clen + 8)
        throw std::runtime_error("Error writing BAM record to pass-through file");
    block_.erase(block_.begin(), block_.begin() + n);
}

void PassThroughWriter::bgzf_append(const void *data, size_t n) {
    const uint8_t *p = (const uint8_t *)data;
    block_.insert(block_.end(), p, p + n);
    while (block_.size() >= BGZF_BLOCK) bgzf_flush_block(BGZF_BLOCK);
}

void PassThroughWriter::write(const AlnRecord &rec, int orient_cnt, bool set_xs, bool set_ts) {
    if (!fp_) return;
    const bool edit = orient_cnt != 0 && (set_xs || set_ts);
    const char strand = orient_cnt > 0 ? '+' : '-';
    if (bam_) {
        const std::string *r = &rec.raw;
        std::string edited;
        if (edit) {
            edited = rec.raw;
            if (set_xs) bam_set_strand_tag(edited, rec.aux_off, "XS", strand);
            if (set_ts) bam_set_strand_tag(edited, rec.aux_off, "TS", strand);
            r = &edited;
        }
        const uint32_t n = (uint32_t)r->size();
        const uint8_t w[4] = {(uint8_t)n, (uint8_t)(n >> 8), (uint8_t)(n >> 16), (uint8_t)(n >> 24)};
        // htslib does not let a record straddle two blocks when it fits one
        if (block_.size() + 4 + n > BGZF_BLOCK && !block_.empty() && 4 + (size_t)n <= BGZF_BLOCK) bgzf_flush_block(block_.size());
        bgzf_append(w, 4);
        bgzf_append(r->data(), n);
    } else {
        bool ok;
        if (edit) {
            std::string line = rec.raw;
            if (set_xs) sam_set_strand_tag(line, "XS", strand);
            if (set_ts) sam_set_strand_tag(line, "TS", strand);
            ok = fwrite(line.data(), 1, line.size(), fp_) == line.size();
        } else {
            ok = fwrite(rec.raw.data(), 1, rec.raw.size(), fp_) == rec.raw.size();
        }
        if (!ok || fputc('\n', fp_) == EOF) throw std::runtime_error("Error writing BAM record to pass-through file");
    }
}

void PassThroughWriter::close() {
    if (!fp_) return;
    if (bam_) {
        if (!block_.empty()) bgzf_flush_block(block_.size());
        bgzf_flush_block(0);                    // the end-of-file marker: an empty block
    }
    const int rc = fclose(fp_);
    fp_ = nullptr;
    if (rc != 0) throw std::runtime_error("Error writing BAM record to pass-through file");
}

}  // namespace ipxhost
