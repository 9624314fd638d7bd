// bgzf_reader.cc -- see bgzf_reader.hh.
#include "bgzf_reader.hh"
#include "fast_inflate.hh"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <sys/stat.h>
#include <zlib.h>

namespace ipxhost {

static const size_t kHeader = 18, kFooter = 8;

static bool bgzf_header_ok(const uint8_t *h) {
    // gzip magic, deflate, FEXTRA; XLEN >= 6 with the BC subfield first (every writer puts it there)
    return h[0] == 31 && h[1] == 139 && h[2] == 8 && (h[3] & 4) && h[12] == 'B' && h[13] == 'C' && h[14] == 2 && h[15] == 0;
}

bool BgzfReader::is_bgzf(const std::string &path) {
    // only files can be probed and reopened; a pipe (/dev/stdin) goes through zlib's gzip reader, which
    // takes BGZF as the multi-member gzip file it is (single-threaded)
    struct stat st;
    if (stat(path.c_str(), &st) != 0 || !S_ISREG(st.st_mode)) return false;
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    uint8_t h[kHeader];
    const bool ok = fread(h, 1, kHeader, f) == kHeader && bgzf_header_ok(h);
    fclose(f);
    return ok;
}

// CPUs this process may use: the hardware threads, cut to the cgroup's CPU quota when there is one (a
// container on a 256-thread host may own 16)
unsigned BgzfReader::usable_cpus() {
    unsigned n = std::max(1u, std::thread::hardware_concurrency());
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {                  // cgroup v2: "<quota|max> <period>"
        char q[32];
        long period = 0;
        if (fscanf(f, "%31s %ld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
            const long quota = atol(q);
            if (quota > 0) n = std::min<unsigned>(n, (unsigned)std::max(1l, (quota + period - 1) / period));
        }
        fclose(f);
    } else if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {   // cgroup v1
        long quota = -1, period = 0;
        if (fscanf(g, "%ld", &quota) != 1) quota = -1;
        fclose(g);
        if (FILE *h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
            if (fscanf(h, "%ld", &period) != 1) period = 0;
            fclose(h);
        }
        if (quota > 0 && period > 0) n = std::min<unsigned>(n, (unsigned)std::max(1l, (quota + period - 1) / period));
    }
    return n;
}

// Three quarters of the usable CPUs, at most 12: on the GPU box's 16-CPU share the inflate stage delivers 1.3 / 2.5 /
// 3.8 / 2.3 GB/s with 4 / 8 / 12 / 16 threads and the whole decode 11.8 M records/s with 8, 16.7 M with 12 -- at 8 the
// inflate is the limit, at 12 the single parsing consumer nearly is, at 16 the threads fight for the quota.
unsigned BgzfReader::default_threads() {
    const unsigned cpus = usable_cpus();
    return std::min(12u, std::max(1u, cpus * 3 / 4));
}

BgzfReader::BgzfReader(const std::string &path, unsigned n_threads) : path_(path) {
    fp_ = fopen(path.c_str(), "rb");
    if (!fp_) throw std::runtime_error("Error opening BAM/SAM/CRAM file: " + path);
    setvbuf(fp_, nullptr, _IOFBF, 1 << 22);
    if (n_threads == 0) {
        const char *e = getenv("IPX_DECODE_THREADS");
        const long v = e ? atol(e) : 0;
        n_threads = v > 0 ? (unsigned)v : default_threads();
    }
    n_threads = std::min(n_threads, 64u);
    // four blocks per thread: 3 MB of inflated data for twelve threads, which stays in the CPU's caches between the
    // worker that fills a block and the thread that reads it (4096 slots, tried so that the pool could run ahead of
    // a consumer busy opening its GPU, made the consumer a third slower: every block came from DRAM)
    ring_.resize((size_t)n_threads * 4);
    for (unsigned i = 0; i < n_threads; ++i) workers_.emplace_back(&BgzfReader::worker, this);
}

BgzfReader::~BgzfReader() {
    {
        std::lock_guard<std::mutex> lk(m_);
        stop_ = true;
    }
    can_produce_.notify_all();
    can_consume_.notify_all();
    for (auto &t : workers_) t.join();
    if (fp_) fclose(fp_);
}

bool BgzfReader::read_block(Slot &s) {
    uint8_t h[kHeader];
    const size_t got = fread(h, 1, kHeader, fp_);
    if (got == 0) return false;
    if (got != kHeader || !bgzf_header_ok(h)) {
        s.error = "not a BGZF block";
        return true;
    }
    const size_t xlen = h[10] | (h[11] << 8);
    const size_t bsize = (size_t)(h[16] | (h[17] << 8)) + 1;        // whole block
    if (xlen < 6 || bsize < 12 + xlen + kFooter) {
        s.error = "bad BGZF block size";
        return true;
    }
    s.in.resize(bsize);
    memcpy(s.in.data(), h, kHeader);
    if (fread(s.in.data() + kHeader, 1, bsize - kHeader, fp_) != bsize - kHeader) s.error = "truncated BGZF block";
    compressed_bytes_ += bsize;
    return true;
}

void BgzfReader::worker() {
    const char *inf_e = getenv("IPX_INFLATE");
    const bool use_fast = !(inf_e && strcmp(inf_e, "zlib") == 0);
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, -15) != Z_OK) {           // no inflater: fail the next block instead of leaving the consumer waiting
        std::unique_lock<std::mutex> lk(m_);
        if (next_claim_ < end_seq_ && !stop_) {
            Slot &s = ring_[next_claim_ % ring_.size()];
            s.error = "cannot initialise zlib";
            s.out.clear();
            s.ready = true;
            ++next_claim_;
            end_seq_ = next_claim_;
        }
        lk.unlock();
        can_consume_.notify_all();
        can_produce_.notify_all();
        return;
    }
    for (;;) {
        Slot *s;
        {
            std::unique_lock<std::mutex> lk(m_);
            can_produce_.wait(lk, [&] { return stop_ || (next_claim_ < end_seq_ && next_claim_ < next_consume_ + ring_.size()); });
            if (stop_ || next_claim_ >= end_seq_) break;
            s = &ring_[next_claim_ % ring_.size()];
            s->error.clear();
            s->out.clear();
            if (!read_block(*s)) {                  // end of file: nothing claimed
                end_seq_ = next_claim_;
                lk.unlock();
                can_consume_.notify_all();
                can_produce_.notify_all();
                break;
            }
            ++next_claim_;
        }
        if (s->error.empty()) {
            const size_t xlen = s->in[10] | (s->in[11] << 8);
            const uint8_t *foot = s->in.data() + s->in.size() - kFooter;
            const uint32_t crc = foot[0] | (foot[1] << 8) | (foot[2] << 16) | ((uint32_t)foot[3] << 24);
            const uint32_t isize = foot[4] | (foot[5] << 8) | (foot[6] << 16) | ((uint32_t)foot[7] << 24);
            if (isize > 65536) {
                s->error = "BGZF block larger than 64 KiB";
            } else {
                s->out.resize(isize);
                inflateReset(&zs);
                zs.next_in = s->in.data() + 12 + xlen;
                zs.avail_in = (uInt)(s->in.size() - 12 - xlen - kFooter);
                zs.next_out = s->out.data();
                zs.avail_out = isize;
                // (an empty block, like the end-of-file marker, has nothing to inflate into)
                // The whole-block decoder first (host/fast_inflate.hh); whatever it does not accept zlib looks at,
                // so a damaged file is reported in zlib's terms.  IPX_INFLATE=zlib takes zlib for everything
                bool done = !isize;
                if (!done && use_fast) done = fast_inflate(zs.next_in, zs.avail_in, s->out.data(), isize) == (long)isize;
                if (!done) {
                    const int rc = inflate(&zs, Z_FINISH);
                    if (rc != Z_STREAM_END || zs.avail_out != 0) s->error = "inflate failed";
                }
                if (s->error.empty() && fast_crc32(0, s->out.data(), isize) != crc) s->error = "CRC mismatch";
            }
        }
        {
            std::lock_guard<std::mutex> lk(m_);
            s->ready = true;
        }
        can_consume_.notify_all();
    }
    inflateEnd(&zs);
}

bool BgzfReader::peek(size_t n, const uint8_t **p) {
    Slot &s = ring_[next_consume_ % ring_.size()];
    if (!cur_valid_) {
        std::unique_lock<std::mutex> lk(m_);
        can_consume_.wait(lk, [&] { return s.ready || next_consume_ >= end_seq_ || stop_; });
        if (!s.ready) return false;                 // end of file
        if (!s.error.empty()) throw std::runtime_error("Error reader BAM record: " + path_ + " (" + s.error + ")");
        cur_valid_ = true;
        cur_off_ = 0;
    }
    if (s.out.size() - cur_off_ < n) return false;
    *p = s.out.data() + cur_off_;
    return true;
}

void BgzfReader::advance(size_t n) {
    Slot &s = ring_[next_consume_ % ring_.size()];
    cur_off_ += n;
    if (cur_off_ >= s.out.size()) {
        {
            std::lock_guard<std::mutex> lk(m_);
            s.ready = false;
            ++next_consume_;
        }
        cur_valid_ = false;
        can_produce_.notify_all();
    }
}

bool BgzfReader::next_block(std::vector<uint8_t> &out) {
    for (;;) {
        const uint8_t *p;
        if (!peek(0, &p)) return false;             // waits for the next block; false: end of file
        Slot &s = ring_[next_consume_ % ring_.size()];
        const size_t left = s.out.size() - cur_off_;
        if (left == 0) { advance(0); continue; }    // an empty block (the end-of-file marker)
        if (cur_off_ == 0) out.swap(s.out);
        else out.assign(s.out.begin() + (long)cur_off_, s.out.end());
        cur_off_ = s.out.size();                    // (after the swap: whatever `out` held; the slot is released next)
        advance(0);
        return true;
    }
}

size_t BgzfReader::read(void *dst, size_t n) {
    size_t done = 0;
    while (done < n) {
        Slot &s = ring_[next_consume_ % ring_.size()];
        if (!cur_valid_) {
            std::unique_lock<std::mutex> lk(m_);
            can_consume_.wait(lk, [&] { return s.ready || next_consume_ >= end_seq_ || stop_; });
            if (!s.ready) break;                    // end of file
            if (!s.error.empty()) throw std::runtime_error("Error reader BAM record: " + path_ + " (" + s.error + ")");
            cur_valid_ = true;
            cur_off_ = 0;
        }
        const size_t take = std::min(n - done, s.out.size() - cur_off_);
        memcpy((char *)dst + done, s.out.data() + cur_off_, take);
        done += take;
        cur_off_ += take;
        if (cur_off_ == s.out.size()) {
            {
                std::lock_guard<std::mutex> lk(m_);
                s.ready = false;
                ++next_consume_;
            }
            cur_valid_ = false;
            can_produce_.notify_all();
        }
    }
    return done;
}

}  // namespace ipxhost
