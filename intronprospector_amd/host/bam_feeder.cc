// bam_feeder.cc -- see bam_feeder.hh.
#include "bam_feeder.hh"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

#include "aln_reader.hh"

namespace ipxhost {

static uint32_t le32(const uint8_t *p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }

void BamFeeder::bad() const { throw std::runtime_error("Error reader BAM record: " + path_); }

BamFeeder::BamFeeder(BgzfReader &src, const std::string &path, const std::vector<int32_t> &canon_tid, unsigned n_workers)
    : src_(src), path_(path), canon_tid_(canon_tid) {
    if (n_workers == 0) {
        const char *e = getenv("IPX_PARSE_THREADS");
        const long v = e ? atol(e) : 0;
        // the inflate pool takes three quarters of the usable CPUs (BgzfReader::default_threads); the parsers
        // share the rest with the splitting thread
        // parsing costs a tenth of inflating: two workers keep up with twelve inflate threads, and together with
        // the splitting thread they stay inside a 16-CPU quota (a seventeenth runnable thread gets the whole
        // cgroup throttled)
        n_workers = v > 0 ? (unsigned)v : (BgzfReader::default_threads() >= 6 ? 2u : 1u);
    }
    if (n_workers == SPLIT_ONLY) n_workers = 0;
    n_workers = std::min(n_workers, 32u);
    for (unsigned i = 0; i < n_workers; ++i) threads_.emplace_back(&BamFeeder::worker, this);
}

BamFeeder::~BamFeeder() {
    {
        std::lock_guard<std::mutex> lk(m_);
        stop_ = true;
    }
    have_task_.notify_all();
    for (auto &t : threads_) t.join();
    blk_.reset();
    strad_.reset();
    queue_.clear();
    for (auto *v : pool_) delete v;
}

std::shared_ptr<std::vector<uint8_t>> BamFeeder::new_buffer() {
    std::vector<uint8_t> *v = nullptr;
    {
        std::lock_guard<std::mutex> lk(pool_m_);
        if (!pool_.empty()) { v = pool_.back(); pool_.pop_back(); }
    }
    if (!v) v = new std::vector<uint8_t>();
    return std::shared_ptr<std::vector<uint8_t>>(v, [this](std::vector<uint8_t> *p) {
        std::lock_guard<std::mutex> lk(pool_m_);
        pool_.push_back(p);
    });
}

bool BamFeeder::fetch_block() {
    if (eof_) return false;
    auto nb = new_buffer();             // its storage goes to the ring in exchange for the block's
    if (!src_.next_block(*nb)) {
        eof_ = true;
        return false;
    }
    blk_ = std::move(nb);
    off_ = 0;
    return true;
}

// CIGAR words the record will occupy in the batch: n_cigar_op, or the CG tag's count behind a placeholder
uint32_t BamFeeder::words_of(const uint8_t *rec, uint32_t block) const {
    const uint32_t l_read_name = rec[8], n_cigar = rec[12] | (rec[13] << 8);
    if (32ull + l_read_name + 4ull * n_cigar > block) bad();
    if (n_cigar != 2) return n_cigar;
    BamFields f;                                   // "<l_seq>S<n>N" may stand for a long CIGAR in the CG tag: look
    parse_bam_record(rec, block, path_, f);
    return f.n_cig;
}

BamFeeder::Status BamFeeder::next_span(int32_t want_tid, uint64_t max_records, uint64_t max_words, Span &sp, int32_t *other_tid) {
    for (;;) {
        if (strad_) {                              // a record assembled from the blocks it straddles waits
            const uint8_t *rec = strad_->data() + 4;
            const uint32_t block = (uint32_t)strad_->size() - 4;
            const int32_t tid = (int32_t)le32(rec);
            if (want_tid != ANY_TARGET && tid != want_tid) { *other_tid = tid; return BOUNDARY; }
            const uint32_t nc = words_of(rec, block);
            if (max_records < 1 || nc > max_words) return FULL;
            sp.data = std::move(strad_);
            strad_.reset();
            sp.begin = 0;
            sp.end = sp.data->size();
            sp.n_records = 1;
            sp.n_words = nc;
            return SPAN;
        }
        if (!blk_ || off_ >= blk_->size()) {
            if (!fetch_block()) return END;
        }
        const uint8_t *p = blk_->data();
        const size_t size = blk_->size(), start = off_;
        uint64_t n = 0, w = 0;
        while (off_ + 4 <= size) {
            const uint32_t bs = le32(p + off_);
            if (bs < 32) bad();
            if (off_ + 4 + (size_t)bs > size) break;                   // the record runs on into the next block
            const uint8_t *rec = p + off_ + 4;
            const int32_t tid = (int32_t)le32(rec);
            if (want_tid != ANY_TARGET && tid != want_tid) {
                if (n) break;
                *other_tid = tid;
                return BOUNDARY;
            }
            const uint32_t nc = words_of(rec, bs);
            if (n + 1 > max_records || w + nc > max_words) {
                if (n) break;
                return FULL;
            }
            ++n;
            w += nc;
            off_ += 4 + (size_t)bs;
        }
        if (n) {
            sp.data = blk_;
            sp.begin = start;
            sp.end = off_;
            sp.n_records = n;
            sp.n_words = w;
            return SPAN;
        }
        // the record at the head is not whole in this block: gather it from the blocks it runs through
        auto s = std::make_shared<std::vector<uint8_t>>(p + off_, p + size);
        off_ = size;
        size_t need = 4;
        for (;;) {
            if (s->size() >= 4) need = 4 + (size_t)le32(s->data());
            if (s->size() >= 4 && le32(s->data()) < 32) bad();
            if (s->size() >= need && need > 4) break;
            if (off_ >= blk_->size() && !fetch_block()) {
                if (s->empty()) return END;
                bad();                                                  // the file ends inside a record
            }
            const size_t take = std::min(need - s->size(), blk_->size() - off_);
            s->insert(s->end(), blk_->begin() + (long)off_, blk_->begin() + (long)(off_ + take));
            off_ += take;
        }
        strad_ = std::move(s);
    }
}

void BamFeeder::dispatch(Span &&sp, const Dest &dst, uint64_t rec_base, uint64_t word_base) {
    {
        std::lock_guard<std::mutex> lk(m_);
        queue_.push_back(Task{std::move(sp), dst, rec_base, word_base});
        ++outstanding_;
    }
    have_task_.notify_one();
}

void BamFeeder::wait_all() {
    std::unique_lock<std::mutex> lk(m_);
    all_done_.wait(lk, [&] { return outstanding_ == 0; });
    if (error_) {
        std::exception_ptr e = error_;
        error_ = nullptr;
        std::rethrow_exception(e);
    }
}

void BamFeeder::worker() {
    for (;;) {
        Task t;
        bool stopping = false;
        {
            std::unique_lock<std::mutex> lk(m_);
            have_task_.wait(lk, [&] { return stop_ || !queue_.empty(); });
            if (queue_.empty()) return;                                 // stop_ and nothing left
            t = std::move(queue_.front());
            queue_.pop_front();
            stopping = stop_;
        }
        try {
            if (!stopping) run(t);                                      // stopping: queued spans are dropped, not parsed
        } catch (...) {
            std::lock_guard<std::mutex> lk(m_);
            if (!error_) error_ = std::current_exception();
        }
        t.sp.data.reset();
        {
            std::lock_guard<std::mutex> lk(m_);
            --outstanding_;
            if (outstanding_ == 0) all_done_.notify_all();
        }
    }
}

void BamFeeder::run(const Task &t) {
    const uint8_t *p = t.sp.data->data();
    size_t off = t.sp.begin;
    uint64_t i = t.rec_base, w = t.word_base;
    const Dest &d = t.dst;
    BamFields f;
    while (off < t.sp.end) {
        const uint32_t bs = le32(p + off);
        const uint8_t *rec = p + off + 4;
        parse_bam_record(rec, bs, path_, f);
        int32_t tid = f.tid;
        if (tid >= 0 && (size_t)tid < canon_tid_.size()) tid = canon_tid_[tid];
        d.tid[i] = tid;
        d.pos[i] = f.pos;
        d.flag[i] = f.flag;
        d.xs[i] = f.xs;
        d.nh[i] = f.nh;
        d.off[i] = (uint32_t)w;
        d.nc[i] = f.n_cig;
        if (f.n_cig) memcpy(d.cigar + w, f.cig, (size_t)f.n_cig * 4);
        w += f.n_cig;
        ++i;
        off += 4 + (size_t)bs;
    }
    if (i != t.rec_base + t.sp.n_records || w != t.word_base + t.sp.n_words) bad();      // splitter and parser disagree
}

}  // namespace ipxhost
