// junctions_extractor.cc -- see junctions_extractor.hh.
#include "junctions_extractor.hh"
#include <chrono>
#include <sys/stat.h>
#include <sys/mman.h>
#include <fcntl.h>
#include <unistd.h>
#include <cstdlib>
#include <cstring>

#include <algorithm>
#include <map>
#include <cctype>
#include <stdexcept>
#include <exception>
#include <system_error>
#include <thread>

namespace ipxhost {

static const size_t BATCH_ALIGNMENTS = 1u << 20;

JunctionFilter junction_filter_parse(const std::string &spec) {
    if (spec == "all") return ALL_SJ_FILTER;
    if (spec == "canon") return CANON_SJ_FILTER;
    throw std::invalid_argument("invalid splice junction filter '" + spec + "', expected 'all' or 'canon'");
}

static double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
struct ScopeTimer { double &acc; double t0; ~ScopeTimer() { acc += now_s() - t0; } };

JunctionsExtractor::JunctionsExtractor(uint32_t min_anchor_length, uint32_t max_anchor_indel_size,
                                       uint32_t min_intron_length, uint32_t max_intron_length,
                                       JunctionFilter junction_filter, Strandness strandness, unsigned excludes,
                                       Genome *genome, bool skip_missing_targets, int device, std::ostream *warn,
                                       std::ostream *trace_fh, bool set_XS_strand_tag, bool set_TS_strand_tag,
                                       const std::vector<int> &devices)
    : genome_(genome), skip_missing_targets_(skip_missing_targets), warn_(warn), trace_fh_(trace_fh),
      set_xs_(set_XS_strand_tag), set_ts_(set_TS_strand_tag) {
    ipx_params p{};
    p.min_anchor_length = min_anchor_length;
    p.max_anchor_indel_size = max_anchor_indel_size;
    p.min_intron_length = min_intron_length;
    p.max_intron_length = max_intron_length;
    p.sj_filter = (int32_t)junction_filter;
    p.strandness = (int32_t)strandness;
    p.excludes = excludes;
    // Keys are (first index of the chromosome name, start, end): names that repeat in the
    // concatenated target vector are folded here, rows are put in std::map order on the host.
    p.scope = IPX_SCOPE_TARGET;
    p.skip_missing_targets = skip_missing_targets ? 1 : 0;
    p.device = devices.empty() ? device : devices[0];
    t_born_ = now_s();
    { const char *e = getenv("IPX_TIMELINE"); timeline_ = e && e[0] == '1'; }
    // the trace and the strand tags walk one device's record list in input order
    if (devices.size() > 1 && (trace_fh_ || set_xs_ || set_ts_))
        throw std::invalid_argument("--devices with more than one GPU cannot be combined with --debug-trace or the strand-tag options");
    params_ = p;
    devices_ = devices;
    init_thread_ = std::thread([this] {
        try {
            init_device();
        } catch (...) {
            init_error_ = std::current_exception();
        }
    });
}

void JunctionsExtractor::init_device() {
    int rc;
    if (devices_.size() > 1) {
        std::vector<int32_t> devs(devices_.begin(), devices_.end());
        rc = ipx_group_create(&params_, (uint32_t)devs.size(), devs.data(), &g_);
    } else {
        rc = ipx_create(&params_, &h_);
    }
    if (rc == IPX_ENODEVICE)
        throw std::runtime_error("no MI355X (gfx950) device available: this build has no CPU path");
    if (rc != IPX_OK) throw std::runtime_error(devices_.size() > 1 ? "ipx_group_create failed" : "ipx_create failed");
    // the trace and the strand tags walk the single junction records (export_records)
    if (trace_fh_ || set_xs_ || set_ts_) {
        rc = ipx_set_option(h_, IPX_OPT_COMBINE, 0);
        if (rc != IPX_OK) throw_last_error(rc);
    }
    if (genome_) {
        const size_t n = genome_->n_seqs();
        std::vector<const char *> names(n);
        std::vector<int64_t> lens(n);
        std::vector<const uint8_t *> two(n), msk(n);
        for (size_t i = 0; i < n; ++i) {
            names[i] = genome_->name(i).c_str();
            lens[i] = genome_->length(i);
            two[i] = genome_->twobit(i);
            msk[i] = genome_->mask(i);
        }
        rc = g_ ? ipx_group_set_genome_2bit(g_, (uint32_t)n, names.data(), lens.data(), two.data(), msk.data())
                : ipx_set_genome_2bit(h_, (uint32_t)n, names.data(), lens.data(), two.data(), msk.data());
        if (rc != IPX_OK) throw_last_error(rc);
    }
    t_create_ = now_s() - t_born_;
}

// before anything that touches the device: the opening thread has finished, the target list is the current one
void JunctionsExtractor::ensure_device() {
    if (init_thread_.joinable()) init_thread_.join();
    if (init_error_) {
        std::exception_ptr e = init_error_;
        std::rethrow_exception(e);                  // (kept: every later call fails the same way)
    }
    if (targets_dirty_) {
        targets_dirty_ = false;
        std::vector<const char *> names(targets_.size());
        for (size_t i = 0; i < targets_.size(); ++i) names[i] = targets_[i].c_str();
        int rc = g_ ? ipx_group_set_targets(g_, (uint32_t)names.size(), names.data(), nullptr)
                    : ipx_set_targets(h_, (uint32_t)names.size(), names.data(), nullptr);
        if (rc != IPX_OK) throw_last_error(rc);
    }
}

JunctionsExtractor::~JunctionsExtractor() {
    if (timeline_)
        std::cerr << "{\"ipx_timeline\": {\"alignments\": " << next_read_index_ << ", \"pushes\": " << n_push_
                  << ", \"files_decoded_on_device\": " << n_device_files_
                  << ", \"read_s\": " << t_read_ << ", \"queue_s\": " << t_queue_ << ", \"push_wait_s\": " << t_push_
                  << ", \"finalize_s\": " << t_finalize_ << ", \"table_to_host_rows_s\": " << t_table_
                  << ", \"device_create_s\": " << t_create_ << ", \"pinned_alloc_s\": " << t_alloc_
                  << ", \"open_s\": " << t_open_ << ", \"identify_s\": " << t_identify_ << ", \"get_junctions_s\": " << t_getj_
                  << ", \"close_s\": " << t_close_ << ", \"extractor_lifetime_s\": " << now_s() - t_born_ << "}}" << std::endl;
    if (init_thread_.joinable()) init_thread_.join();
    // The feeder's workers parse straight into the pinned batches: stop and join them BEFORE that memory goes (an
    // exception that unwinds out of the read loop -- a damaged block, a truncated record -- leaves spans dispatched;
    // as a member the feeder would only be destroyed after this body had freed the batches under the workers).
    feeder_.reset();
    if (h_) ipx_destroy(h_);
    if (g_) ipx_group_destroy(g_);
    for (HostBatch &b : batch_) {
        if (b.mem) ipx_host_free(b.mem);
        if (b.raw) ipx_host_free(b.raw);
    }
    if (timeline_) fprintf(stderr, "ipx_timeline: extractor gone %.3f s after it was made\n", now_s() - t_born_);
}

// (re)allocate a batch's pinned byte buffer for BAM records; what it holds is kept
void JunctionsExtractor::size_raw(HostBatch &b, size_t cap_bytes) {
    ensure_device();
    struct AllocTimer { double &acc; double t0; ~AllocTimer() { acc += now_s() - t0; } } alloc_timer{t_alloc_, now_s()};
    void *m = nullptr;
    if (ipx_host_alloc(&m, cap_bytes) != IPX_OK) throw std::runtime_error("out of pinned host memory for the alignment batches");
    if (b.raw) {
        memcpy(m, b.raw, b.raw_n);
        ipx_host_free(b.raw);
    }
    b.raw = m;
    b.raw_cap = cap_bytes;
}


// (re)allocate a batch's pinned arrays; the batch must be empty
void JunctionsExtractor::size_batch(HostBatch &b, size_t cap_aln, size_t cap_words) {
    ensure_device();             // pinned memory comes from the HIP runtime: without a device say so, not "out of memory"
    struct AllocTimer { double &acc; double t0; ~AllocTimer() { acc += now_s() - t0; } } alloc_timer{t_alloc_, now_s()};
    if (b.mem) ipx_host_free(b.mem);
    b = HostBatch();
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t o_tid = 0, o_pos = al(o_tid + cap_aln * 4), o_flag = al(o_pos + cap_aln * 4), o_xs = al(o_flag + cap_aln * 2),
                 o_nh = al(o_xs + cap_aln), o_off = al(o_nh + cap_aln), o_nc = al(o_off + cap_aln * 4),
                 o_cig = al(o_nc + cap_aln * 4), total = al(o_cig + cap_words * 4);
    if (ipx_host_alloc(&b.mem, total) != IPX_OK) throw std::runtime_error("out of pinned host memory for the alignment batches");
    char *m = (char *)b.mem;
    b.tid = (int32_t *)(m + o_tid); b.pos = (int32_t *)(m + o_pos); b.flag = (uint16_t *)(m + o_flag);
    b.xs = (uint8_t *)(m + o_xs); b.nh = (uint8_t *)(m + o_nh); b.off = (uint32_t *)(m + o_off);
    b.nc = (uint32_t *)(m + o_nc); b.cigar = (uint32_t *)(m + o_cig);
    b.cap_aln = cap_aln;
    b.cap_words = cap_words;
}

void JunctionsExtractor::throw_last_error(int status) {
    ipx_error e;
    if (g_) ipx_group_last_error(g_, &e);
    else ipx_last_error(h_, &e);
    const std::string msg = e.detail[0] ? std::string(e.detail) : ("ipx status " + std::to_string(status));
    // a record the device could not parse: what AlnReader::read throws for it
    if (status == IPX_EINVAL && msg.compare(0, 23, "ipx_push_bam: malformed") == 0)
        throw std::runtime_error("Error reader BAM record: " + bam_);
    if (status == IPX_EDATA) {
        switch (e.kind) {
            case IPX_DATA_CIGAR: throw std::invalid_argument(msg);      // junctions_extractor.cc:313
            case IPX_DATA_CATEGORY: throw std::logic_error(msg);        // :74
            default: throw std::runtime_error(msg);                     // genome.cc:66,69
        }
    }
    throw std::runtime_error(msg);
}

// bytes of a BGZF file that may still be decoded whole on the device (see open())
static uint64_t device_inflate_limit() {
    uint64_t limit = 8192ull << 20;
    if (const char *m = getenv("IPX_DEVICE_INFLATE_MAX")) {
        const long long v = atoll(m);
        if (v > 0) limit = (uint64_t)v << 20;
    }
    // what the host can spare: MemAvailable, and under a cgroup its limit minus what is in use
    uint64_t avail = ~0ull;
    if (FILE *f = fopen("/proc/meminfo", "r")) {
        char line[256];
        while (fgets(line, sizeof line, f)) {
            unsigned long long kb = 0;
            if (sscanf(line, "MemAvailable: %llu kB", &kb) == 1) { avail = (uint64_t)kb << 10; break; }
        }
        fclose(f);
    }
    auto read_u64 = [](const char *path, uint64_t &out) {
        FILE *f = fopen(path, "r");
        if (!f) return false;
        char buf[64] = {0};
        const bool ok = fgets(buf, sizeof buf, f) != nullptr && buf[0] >= '0' && buf[0] <= '9';
        fclose(f);
        if (ok) out = strtoull(buf, nullptr, 10);
        return ok;
    };
    uint64_t cg_max = 0, cg_cur = 0;
    if (read_u64("/sys/fs/cgroup/memory.max", cg_max) && read_u64("/sys/fs/cgroup/memory.current", cg_cur) && cg_max > cg_cur)
        avail = std::min(avail, cg_max - cg_cur);
    if (avail != ~0ull) limit = std::min(limit, avail / 2);
    return limit;
}

// junctions_extractor.cc:571-586 (save_targets :564-568 appends)
void JunctionsExtractor::open(const std::string &bam) {
    ScopeTimer scope_timer{t_open_, now_s()};
    bam_ = bam;
    reader_.reset(new AlnReader(bam));
    reader_->keep_qname(trace_fh_ != nullptr);        // read names only go into the --debug-trace rows
    for (const std::string &nm : reader_->target_names()) {
        int32_t first = (int32_t)targets_.size();
        for (size_t i = 0; i < targets_.size(); ++i)
            if (targets_[i] == nm) { first = (int32_t)i; break; }
        targets_.push_back(nm);
        canon_tid_.push_back(first);
    }
    targets_dirty_ = true;          // handed to the device by ensure_device()
    // A BGZF BAM file that the device is to decode whole (try_device_file) is read from here on, beside the thread
    // that opens the device: by the time the first target is asked for, both are done.  (Read in vain if a
    // pass-through or a trace then asks for the records one by one: they are rarely wanted of files this size.)
    // The whole file then lies in host memory, and its bytes, what they inflate to (4-5 times as much) and the decoded
    // batch lie on the device: that path is for files that FIT -- at most IPX_DEVICE_INFLATE_MAX megabytes (default
    // 8192: ~60 GB on the device) and at most half of the memory this process may still take; larger files are streamed
    // through the batches of the ordinary reader, as the reference streams them (junctions_extractor.cc:620-634).
    // (Round 3 had no bound: a 50 GB file was read whole, in vain or until the host ran out of memory.)
    {
        const char *e = getenv("IPX_DEVICE_INFLATE");
        struct stat st;
        if (!(e && e[0] == '0') && !g_ && !trace_fh_ && reader_->bgzf() && stat(bam.c_str(), &st) == 0 && S_ISREG(st.st_mode) &&
            st.st_size >= 28 && ((e && e[0] == '1') || st.st_size >= (16 << 20)) && (uint64_t)st.st_size <= device_inflate_limit())
            file_load_.start(bam, (size_t)st.st_size, 4);
    }
    if (trace_fh_)          // junctions_extractor.cc:579-584 (nine names over ten columns, as there)
        *trace_fh_ << "chrom" << "\t" << "intron_start" << "\t" << "intron_end" << "\t" << "qname" << "\t" << "flag" << "\t"
                   << "strand" << "\t" << "splice_sites" << "\t" << "anchor_start" << "\t" << "anchor_end" << std::endl;
}

// junctions_extractor.cc:589-602: same container as the input, header first
void JunctionsExtractor::open_pass_through(const std::string &bam_pass_through) {
    pass_through_.reset(new PassThroughWriter(bam_pass_through, *reader_));
    reader_->keep_raw(true);
    file_load_.drop();              // the records are wanted one by one: the file is not decoded on the device
}

// junctions_extractor.cc:667-674: whatever the target loop left unread (unaligned reads, and
// anything the sorted-mode loop never got to)
void JunctionsExtractor::copy_unaligned_reads() {
    if (!pass_through_) return;
    while (read_align()) pass_through_->write(aln_buf_, 0, false, false);
}

// junctions_extractor.cc:605-617
void JunctionsExtractor::close() {
    ScopeTimer scope_timer{t_close_, now_s()};
    if (pass_through_) {
        pass_through_->close();
        pass_through_.reset();
    }
    feeder_.reset();
    if (device_parse_) {
        // the next input may be SAM text, read record by record into the other kind of batch: nothing of this kind
        // stays behind (an exception may have left a batch half filled; a push still in flight reads the span tables)
        if (h_) (void)ipx_sync(h_);
        for (HostBatch &b : batch_) {
            b.n = b.nw = b.raw_n = 0;
            b.span_begin.clear();
            b.span_first.clear();
            b.tables_lent = false;
        }
        device_parse_ = false;
    }
    reader_.reset();
    file_load_.drop();
    previous_target_ = -1;
    aln_pending_ = false;
    device_file_ = device_file_tried_ = false;
    dev_tid_.clear();
    dev_cursor_ = 0;
}

// junctions_extractor.cc:620-634
bool JunctionsExtractor::read_align() {
    if (aln_pending_) {
        aln_pending_ = false;
        return true;
    }
    if (timeline_ && (++tick_ & 63u) == 0) {            // every 64th call is timed, the sums are scaled
        const double t0 = now_s();
        const bool ok = reader_->read(aln_buf_);
        t_read_ += 64.0 * (now_s() - t0);
        return ok;
    }
    return reader_->read(aln_buf_);
}

// process_alignment (junctions_extractor.cc:533-549) is the device's job: queue the record
void JunctionsExtractor::process_alignment(const AlnRecord &aln) {
    struct QueueTimer { double &acc; double t0; bool on; ~QueueTimer() { if (on) acc += 64.0 * (now_s() - t0); } }
        queue_timer{t_queue_, 0.0, timeline_ && (tick_ & 63u) == 0};
    if (queue_timer.on) queue_timer.t0 = now_s();
    int32_t tid = aln.tid;
    if (tid >= 0 && (size_t)tid < canon_tid_.size()) tid = canon_tid_[tid];
    if (trace_fh_) {
        trace_qname_.push_back(aln.qname);
        trace_flag_.push_back(aln.flag);
    }
    const size_t nc = aln.n_cig;
    HostBatch *b = &batch_[cur_];
    if (b->n == b->cap_aln || b->nw + nc > b->cap_words) {
        // full: hand it to the device and go on with the other set, grown if the input turns out to be large
        const size_t had = b->cap_aln;
        flush_batch();
        b = &batch_[cur_];
        size_t want_aln = std::max<size_t>(b->cap_aln, std::min<size_t>(BATCH_ALIGNMENTS, std::max<size_t>(had * 8, 1u << 14)));
        size_t want_words = std::max<size_t>(std::max(b->cap_words, want_aln * 8), nc);
        if (want_aln > b->cap_aln || want_words > b->cap_words) size_batch(*b, want_aln, want_words);
    }
    const size_t i = b->n++;
    b->tid[i] = tid;
    b->pos[i] = (int32_t)aln.pos;
    b->flag[i] = aln.flag;
    b->xs[i] = aln.xs;
    b->nh[i] = aln.nh;
    b->off[i] = (uint32_t)b->nw;
    b->nc[i] = (uint32_t)nc;
    if (nc) memcpy(b->cigar + b->nw, aln.cig, nc * sizeof(uint32_t));
    b->nw += nc;
    table_valid_ = false;
}

void JunctionsExtractor::flush_batch() {
    HostBatch &hb = batch_[cur_];
    if (feeder_) {                               // the parsers are still filling the batch
        const double tw = timeline_ ? now_s() : 0.0;
        feeder_->wait_all();
        if (timeline_) t_queue_ += now_s() - tw;
    }
    if (hb.n == 0) return;
    ensure_device();
    if (device_parse_) {
        hb.span_begin.push_back(hb.raw_n);
        hb.span_first.push_back(hb.n);
        hb.tid_map = canon_tid_;
        ipx_bam_batch bb{};
        bb.n_alignments = hb.n;
        bb.n_cigar_words = hb.nw;
        bb.base_read_index = next_read_index_;
        bb.bytes = (const uint8_t *)hb.raw;
        bb.n_bytes = hb.raw_n;
        bb.span_begin = hb.span_begin.data();
        bb.span_first = hb.span_first.data();
        bb.n_spans = (uint32_t)(hb.span_begin.size() - 1);
        bb.tid_map = hb.tid_map.data();
        bb.n_tid_map = (uint32_t)hb.tid_map.size();
        bb.memory = IPX_MEM_HOST;
        const double t0 = timeline_ ? now_s() : 0.0;
        const int rc = ipx_push_bam_async(h_, &bb);       // completes the other set's push, queues this one
        if (timeline_) { t_push_ += now_s() - t0; ++n_push_; }
        next_read_index_ += hb.n;
        hb.n = hb.nw = hb.raw_n = 0;
        hb.tables_lent = true;                            // cleared when this set is filled again: the push is over by then
        cur_ ^= 1;
        if (rc != IPX_OK) throw_last_error(rc);
        return;
    }
    ipx_batch b{};
    b.n_alignments = hb.n;
    b.n_cigar_words = hb.nw;
    b.base_read_index = next_read_index_;
    b.tid = hb.tid; b.pos = hb.pos; b.flag = hb.flag; b.xs = hb.xs;
    b.nh = hb.nh; b.cigar_off = hb.off; b.n_cigar = hb.nc; b.cigar = hb.cigar;
    b.memory = IPX_MEM_HOST;
    const double t0 = timeline_ ? now_s() : 0.0;
    // one GPU: the copies and the scan are queued and this set of arrays stays with the device until the NEXT
    // push (which first completes this one) -- by then the loop has filled the other set
    int rc = g_ ? ipx_group_push_batch(g_, &b) : ipx_push_batch_async(h_, &b);
    if (timeline_) { t_push_ += now_s() - t0; ++n_push_; }
    next_read_index_ += hb.n;
    hb.n = hb.nw = 0;
    if (!g_) cur_ ^= 1;
    if (rc != IPX_OK) throw_last_error(rc);
}

void JunctionsExtractor::FileLoad::start(const std::string &path, size_t n_bytes, unsigned n_threads) {
    drop();
    void *p = nullptr;
    if (posix_memalign(&p, (size_t)2 << 20, n_bytes + 64) != 0) return;
    (void)madvise(p, n_bytes + 64, MADV_HUGEPAGE);
    buf = (uint8_t *)p;
    size = n_bytes;
    failed = false;
    const size_t step = ((n_bytes / n_threads) + ((size_t)1 << 20)) & ~(((size_t)1 << 20) - 1);
    for (unsigned k = 0; k < n_threads; ++k) {
        const size_t lo = std::min(n_bytes, (size_t)k * step), hi = std::min(n_bytes, lo + step);
        if (hi == lo) break;
        readers.emplace_back([this, path, lo, hi] {
            const int fd = ::open(path.c_str(), O_RDONLY);
            if (fd < 0) { failed = true; return; }
            size_t at = lo;
            while (at < hi && !failed) {
                const ssize_t got = pread(fd, buf + at, std::min<size_t>(hi - at, (size_t)16 << 20), (off_t)at);
                if (got <= 0) { failed = true; break; }
                at += (size_t)got;
            }
            ::close(fd);
        });
    }
}

bool JunctionsExtractor::FileLoad::wait() {
    for (std::thread &t : readers) t.join();
    readers.clear();
    return buf && !failed;
}

void JunctionsExtractor::FileLoad::drop() {
    failed = true;
    for (std::thread &t : readers) t.join();
    readers.clear();
    // (giving a gigabyte back to the system takes tens of milliseconds: nobody has to wait for it)
    if (buf && size >= ((size_t)64 << 20)) {
        uint8_t *p = buf;
        std::thread([p] { free(p); }).detach();
    } else {
        free(buf);
    }
    buf = nullptr;
    size = 0;
}

// The whole BGZF file to the device (ipx_decode_bgzf).  false: not this file (too small, not asked for, damaged,
// unaligned, no memory): the caller goes on with the reader it has, which has not consumed a record yet.
bool JunctionsExtractor::try_device_file() {
    const char *e = getenv("IPX_DEVICE_INFLATE");
    if (e && e[0] == '0') return false;
    if (!file_load_.buf && file_load_.readers.empty()) return false;          // open() did not start it: not this file
    const double t0 = timeline_ ? now_s() : 0.0;
    struct Drop { FileLoad &f; ~Drop() { f.drop(); } } drop{file_load_};
    if (!file_load_.wait()) return false;
    if (timeline_) fprintf(stderr, "ipx_timeline: waited %.3f s for the file's bytes\n", now_s() - t0);
    const size_t size = file_load_.size;
    // the block headers (SAM spec 4.1: gzip member with a BC extra subfield that holds BSIZE - 1)
    const uint8_t *d = file_load_.buf;
    std::vector<uint64_t> pb, pe;
    std::vector<uint32_t> isz, crc;
    size_t p = 0;
    while (p < size) {
        if (p + 18 > size || d[p] != 31 || d[p + 1] != 139 || d[p + 2] != 8 || !(d[p + 3] & 4)) return false;
        const size_t xlen = d[p + 10] | (d[p + 11] << 8);
        if (p + 12 + xlen + 8 > size) return false;
        size_t bsize = 0;
        for (size_t q = p + 12; q + 4 <= p + 12 + xlen;) {
            const size_t slen = d[q + 2] | (d[q + 3] << 8);
            if (d[q] == 'B' && d[q + 1] == 'C' && slen == 2 && q + 6 <= p + 12 + xlen) bsize = (size_t)(d[q + 4] | (d[q + 5] << 8)) + 1;
            q += 4 + slen;
        }
        if (bsize < 12 + xlen + 8 || p + bsize > size) return false;
        pb.push_back(p + 12 + xlen);
        pe.push_back(p + bsize - 8);
        uint32_t w[2];
        memcpy(w, d + p + bsize - 8, 8);
        crc.push_back(w[0]);
        isz.push_back(w[1]);
        p += bsize;
    }
    if (pb.empty()) return false;
    ensure_device();
    ipx_bgzf_file f{};
    f.bytes = d;
    f.n_bytes = size;
    f.n_blocks = pb.size();
    f.payload_begin = pb.data();
    f.payload_end = pe.data();
    f.isize = isz.data();
    f.crc32 = crc.data();
    f.records_begin = 4 + reader_->raw_header().size();          // magic, then what raw_header() holds: l_text ... the references
    f.tid_map = canon_tid_.data();
    f.n_tid_map = (uint32_t)canon_tid_.size();
    const double t1 = timeline_ ? now_s() : 0.0;
    if (ipx_decode_bgzf(h_, &f, &dev_batch_) != IPX_OK) return false;
    if (timeline_) fprintf(stderr, "ipx_timeline: block headers + device %.3f s, ipx_decode_bgzf %.3f s\n", t1 - t0, now_s() - t1);
    dev_tid_.resize(dev_batch_.n_alignments);
    if (!dev_tid_.empty() && ipx_copy(h_, dev_tid_.data(), dev_batch_.tid, dev_tid_.size() * sizeof(int32_t)) != IPX_OK) return false;
    dev_cursor_ = 0;
    ++n_device_files_;
    if (timeline_) t_read_ += now_s() - t0;
    return true;
}

// feed_target over the device-resident batch: the refIDs tell where the target's run of alignments ends
bool JunctionsExtractor::feed_target_device(int32_t target_index, int32_t *other_tid) {
    const uint64_t n = dev_batch_.n_alignments;
    if (dev_cursor_ >= n) return false;
    uint64_t j = n;
    if (target_index != BamFeeder::ANY_TARGET) {
        j = dev_cursor_;
        if (dev_tid_[j] != target_index) { *other_tid = dev_tid_[j]; return true; }
        while (j < n && dev_tid_[j] == target_index) ++j;
    }
    const uint64_t lo = dev_cursor_;
    ipx_batch b = dev_batch_;                       // cigar, n_cigar_words stay: cigar_off is absolute
    b.n_alignments = j - lo;
    b.base_read_index = next_read_index_;
    b.tid += lo; b.pos += lo; b.flag += lo; b.xs += lo; b.nh += lo; b.cigar_off += lo; b.n_cigar += lo;
    const double t0 = timeline_ ? now_s() : 0.0;
    const int rc = ipx_push_batch(h_, &b);
    if (timeline_) { t_push_ += now_s() - t0; ++n_push_; }
    next_read_index_ += j - lo;
    dev_cursor_ = j;
    table_valid_ = false;
    if (rc != IPX_OK) throw_last_error(rc);
    if (j < n) { *other_tid = dev_tid_[j]; return true; }
    return false;
}

bool JunctionsExtractor::use_feeder() {
    if (feeder_ || device_file_) return true;
    if (!reader_ || !reader_->bgzf() || pass_through_ || trace_fh_ || aln_pending_) return false;
    if (!g_ && !device_file_tried_) {
        device_file_tried_ = true;
        if (try_device_file()) {
            device_file_ = true;
            return true;
        }
    }
    // One GPU: the device can decode the records itself (ipx_push_bam); this thread then only cuts the inflated
    // blocks into runs of whole records and copies them to pinned memory (0.13 s for the 2.3 GB of a 10 M-record
    // BAM).  IPX_DEVICE_PARSE=1 asks for it, 0 forbids it; unasked it is used where the feeder below would be:
    // on the GPU box's 16-CPU share the command line is bound by the twelve inflate threads either way
    // (1.3-1.4 s with the record-by-record loop, 1.4-1.5 s with this, profiles/r02_cli_timeline.jsonl)
    const char *dp = getenv("IPX_DEVICE_PARSE");
    const bool dp_on = dp ? dp[0] != '0' : (!getenv("IPX_PARSE_THREADS") && BgzfReader::usable_cpus() >= 24);
    if (!g_ && dp_on) {
        device_parse_ = true;
        feeder_.reset(new BamFeeder(*reader_->bgzf(), reader_->path(), canon_tid_, BamFeeder::SPLIT_ONLY));
        return true;
    }
    // IPX_PARSE_THREADS=N (N >= 1) asks for it, 0 forbids it.  Unasked, it is used where the parsers do not take
    // CPUs away from the inflate pool: on the GPU box's 16-CPU share (12 inflate threads + the loop) a 10 M-record
    // BAM took 1.3-1.4 s without it and 1.3-1.5 s with it -- inflating is 7 of the 8 CPU-seconds either way
    const char *e = getenv("IPX_PARSE_THREADS");
    const long asked = e ? atol(e) : -1;
    if (asked == 0) return false;
    if (asked < 0 && BgzfReader::usable_cpus() < 24) return false;
    feeder_.reset(new BamFeeder(*reader_->bgzf(), reader_->path(), canon_tid_));
    return true;
}

static const size_t RAW_BATCH_BYTES = (size_t)96 << 20;

bool JunctionsExtractor::feed_target(int32_t target_index, int32_t *other_tid) {
    if (device_file_) return feed_target_device(target_index, other_tid);
    while (device_parse_) {
        HostBatch *b = &batch_[cur_];
        if (b->tables_lent) {                      // the push that read them completed when the other set was pushed
            b->span_begin.clear();
            b->span_first.clear();
            b->tables_lent = false;
        }
        if (!b->raw) {
            struct stat st;
            const bool large = stat(reader_->path().c_str(), &st) == 0 && (uint64_t)st.st_size > (16ull << 20);
            size_raw(*b, large ? RAW_BATCH_BYTES : (size_t)1 << 20);
        }
        BamFeeder::Span sp;
        const double t0 = timeline_ ? now_s() : 0.0;
        const BamFeeder::Status st = feeder_->next_span(target_index, BATCH_ALIGNMENTS - b->n, (BATCH_ALIGNMENTS * 64) - b->nw, sp, other_tid);
        const double t1 = timeline_ ? now_s() : 0.0;
        if (timeline_) t_read_ += t1 - t0;
        if (st == BamFeeder::SPAN) {
            const size_t bytes = sp.end - sp.begin;
            if (b->raw_n + bytes > b->raw_cap) {
                size_t filled_cap = 0;
                if (b->n) {                        // this set is full: the run opens the other one
                    filled_cap = b->raw_cap;
                    flush_batch();
                    b = &batch_[cur_];
                    if (b->tables_lent) { b->span_begin.clear(); b->span_first.clear(); b->tables_lent = false; }
                }
                // the second set is sized like the first (it used to start at 1 MB and, a span being at most 64 KiB,
                // never grew: batches alternated 96 MB / 1 MB and the small one's push waited for the large one)
                if (!b->raw || bytes > b->raw_cap || b->raw_cap < filled_cap) {
                    const size_t want = std::min(RAW_BATCH_BYTES, std::max(b->raw_cap * 8, (size_t)1 << 20));
                    size_raw(*b, std::max(std::max(want, bytes), filled_cap));
                }
            }
            b->span_begin.push_back(b->raw_n);
            b->span_first.push_back(b->n);
            memcpy((char *)b->raw + b->raw_n, sp.data->data() + sp.begin, bytes);
            b->raw_n += bytes;
            b->n += sp.n_records;
            b->nw += sp.n_words;
            table_valid_ = false;
            if (timeline_) t_queue_ += now_s() - t1;
        } else if (st == BamFeeder::FULL) {
            if (b->n == 0) throw std::runtime_error("Error reader BAM record: " + reader_->path() + " (a CIGAR of more than 2^26 operations)");
            flush_batch();
        } else {
            return st == BamFeeder::BOUNDARY;
        }
    }
    for (;;) {
        HostBatch *b = &batch_[cur_];
        if (b->cap_aln == 0) {
            // a large file gets full-size arrays at once (pinned allocations cost milliseconds each); a small one
            // starts small and grows, see below
            struct stat st;
            const bool large = stat(reader_->path().c_str(), &st) == 0 && (uint64_t)st.st_size > (64ull << 20);
            size_batch(*b, large ? BATCH_ALIGNMENTS : (size_t)1 << 16, large ? BATCH_ALIGNMENTS * 8 : (size_t)1 << 19);
        }
        BamFeeder::Span sp;
        const double t0 = timeline_ ? now_s() : 0.0;
        const BamFeeder::Status st = feeder_->next_span(target_index, b->cap_aln - b->n, b->cap_words - b->nw, sp, other_tid);
        if (timeline_) t_read_ += now_s() - t0;
        if (st == BamFeeder::SPAN) {
            const BamFeeder::Dest d{b->tid, b->pos, b->flag, b->xs, b->nh, b->off, b->nc, b->cigar};
            const uint64_t n = sp.n_records, w = sp.n_words;
            feeder_->dispatch(std::move(sp), d, b->n, b->nw);
            b->n += n;
            b->nw += w;
            table_valid_ = false;
        } else if (st == BamFeeder::FULL) {
            const size_t had_aln = b->cap_aln, had_words = b->cap_words;
            const bool empty = b->n == 0;
            flush_batch();
            b = &batch_[cur_];
            // the next set of arrays: eight times the last one up to BATCH_ALIGNMENTS; a record that did not even
            // fit an empty batch needs more room for its CIGAR
            const size_t want_aln = std::max(b->cap_aln, std::min<size_t>(BATCH_ALIGNMENTS, had_aln * 8));
            const size_t want_words = std::max(std::max(b->cap_words, want_aln * 8), empty ? had_words * 4 : (size_t)0);
            if (want_aln > b->cap_aln || want_words > b->cap_words) size_batch(*b, want_aln, want_words);
        } else {
            return st == BamFeeder::BOUNDARY;
        }
    }
}

// identify_junctions_for_target (junctions_extractor.cc:656-664) with
// process_target_alignment (:637-652)
void JunctionsExtractor::identify_junctions_for_target(int target_index) {
    ScopeTimer scope_timer{t_identify_, now_s()};
    if (use_feeder()) {
        int32_t other = -1;
        if (feed_target(target_index, &other) && other >= 0 && other < previous_target_) {
            finalize_table();
            throw std::runtime_error("BAM appears to not be coordinate sorted");
        }
        previous_target_ = target_index;
        finalize_table();
        return;
    }
    while (read_align()) {
        if (aln_buf_.tid != target_index) {
            if (aln_buf_.tid >= 0 && aln_buf_.tid < previous_target_) {
                // alignments of this target queued so far are processed first: an exception one of
                // them raises comes earlier in the input than this one
                finalize_table();
                throw std::runtime_error("BAM appears to not be coordinate sorted");
            }
            aln_pending_ = true;
            break;
        }
        process_alignment(aln_buf_);
        if (pass_through_) {                    // process_target_alignment, :647-650
            if (set_xs_ || set_ts_) {
                pt_pending_.add(aln_buf_, next_read_index_ + batch_[cur_].n - 1);
            } else {
                pass_through_->write(aln_buf_, 0, false, false);
            }
        }
    }
    previous_target_ = target_index;
    finalize_table();       // surfaces the reference's exceptions where its loop would throw
}

// identify_junctions_for_bam (junctions_extractor.cc:678-683)
void JunctionsExtractor::identify_junctions_for_bam() {
    ScopeTimer scope_timer{t_identify_, now_s()};
    if (use_feeder()) {
        int32_t other = -1;
        (void)feed_target(BamFeeder::ANY_TARGET, &other);
        finalize_table();
        return;
    }
    while (read_align()) process_alignment(aln_buf_);
    finalize_table();
}

static char complement(char c) {
    switch (c) {
        case 'A': return 'T'; case 'T': return 'A'; case 'G': return 'C'; case 'C': return 'G';
        case 'a': return 't'; case 't': return 'a'; case 'g': return 'c'; case 'c': return 'g';
        default: return c;
    }
}

void JunctionsExtractor::finalize_table() {
    flush_batch();
    ensure_device();
    ipx_table t;
    const double t0 = timeline_ ? now_s() : 0.0;
    int rc = g_ ? ipx_group_finalize(g_, &t) : ipx_finalize(h_, IPX_MEM_HOST, &t);
    const double t1 = timeline_ ? now_s() : 0.0;
    t_finalize_ += t1 - t0;
    struct RowTimer { double &acc; double t0; bool on; ~RowTimer() { if (on) acc += now_s() - t0; } } row_timer{t_table_, t1, timeline_};
    if (rc != IPX_OK) throw_last_error(rc);
    // Rows are independent of each other (a row's confidence is an entropy over ITS read offsets: 2 M logarithms for
    // 200 000 junctions): large tables are filled by a few threads, each its own range of rows.
    table_.clear();
    table_.resize(t.n_junctions);
    auto fill_rows = [&](uint64_t row_lo, uint64_t row_hi) {
    for (uint64_t i = row_lo; i < row_hi; ++i) {
        Junction j;
        j.chrom = targets_[t.target[i]];
        j.intron_start = t.intron_start[i];
        j.intron_end = t.intron_end[i];
        j.anchor_start = t.anchor_start[i];
        j.anchor_end = t.anchor_end[i];
        j.strand = (char)t.strand[i];
        j.ijunc = ijunc_ + t.ijunc[i];
        for (int c = 0; c < 3; ++c) j.read_counts[c] = t.read_counts[3 * i + c];
        // splice_sites as get_splice_sites / correct_strand leave it (junctions_extractor.cc:405-444):
        // upper(donor/acceptor); reverse-complemented when the junction ends up on '-';
        // lower-cased when that spelling is not canonical
        const uint16_t m = t.motif[i];
        if (m & IPX_MOTIF_PRESENT) {
            char g[5];
            if (m & IPX_MOTIF_NONACGT) {
                g[0] = genome_->raw(j.chrom, j.intron_start); g[1] = genome_->raw(j.chrom, (int64_t)j.intron_start + 1);
                g[3] = genome_->raw(j.chrom, (int64_t)j.intron_end - 2); g[4] = genome_->raw(j.chrom, (int64_t)j.intron_end - 1);
            } else {
                static const char acgt[] = "ACGT";
                g[0] = acgt[(m >> 6) & 3]; g[1] = acgt[(m >> 4) & 3]; g[3] = acgt[(m >> 2) & 3]; g[4] = acgt[m & 3];
            }
            g[2] = '/';
            for (char &c : g) c = (char)toupper((unsigned char)c);
            bool canon;
            if (j.strand == '-') {
                char r[5];
                for (int k = 0; k < 5; ++k) r[4 - k] = complement(g[k]);
                for (int k = 0; k < 5; ++k) g[k] = r[k];
                canon = (m & IPX_MOTIF_CANON_RC) != 0;
            } else {
                canon = (m & IPX_MOTIF_CANON_P) != 0;
            }
            if (!canon)
                for (char &c : g) c = (char)tolower((unsigned char)c);
            j.splice_sites.assign(g, 5);
        }
        const uint64_t hb = t.hist_begin[i], he = t.hist_begin[i + 1];
        j.confidence = confidence_from_histogram(t.hist_count + hb, he - hb, j.total_read_count());
        table_[i] = std::move(j);
    }
    };
    {
        const unsigned want = t.n_junctions >= 20000 ? std::min(8u, std::max(1u, BgzfReader::usable_cpus() / 2)) : 1u;
        if (want <= 1) {
            fill_rows(0, t.n_junctions);
        } else {
            // A worker's exception (std::bad_alloc from a row's strings) is carried back and thrown here, after every
            // thread that started has been joined; a thread that cannot be started has its range filled on this one.
            std::vector<std::thread> workers;
            std::vector<std::exception_ptr> failed(want);
            const uint64_t step = (t.n_junctions + want - 1) / want;
            for (unsigned k = 0; k < want; ++k) {
                const uint64_t lo = std::min<uint64_t>(t.n_junctions, (uint64_t)k * step), hi = std::min<uint64_t>(t.n_junctions, lo + step);
                if (lo >= hi) continue;
                auto guarded = [&fill_rows, &failed, k, lo, hi]() {
                    try { fill_rows(lo, hi); } catch (...) { failed[k] = std::current_exception(); }
                };
                try {
                    workers.emplace_back(guarded);
                } catch (const std::system_error &) {
                    guarded();
                }
            }
            for (std::thread &w : workers) w.join();
            for (const std::exception_ptr &e : failed)
                if (e) std::rethrow_exception(e);
        }
    }
    // missing-sequence warnings, once per chromosome, in the order the reference meets them
    // (junctions_extractor.cc:411-416): first record on a chromosome the genome does not hold
    if (genome_ && skip_missing_targets_) {
        std::vector<std::pair<uint64_t, std::string>> first_seen;
        for (uint64_t i = 0; i < t.n_junctions; ++i) {
            const std::string &chrom = table_[i].chrom;
            if (genome_->has_seq(chrom) || missing_genomic_warned_.count(chrom)) continue;
            bool found = false;
            for (auto &fs : first_seen)
                if (fs.second == chrom) { fs.first = std::min(fs.first, t.creator_order[i]); found = true; break; }
            if (!found) first_seen.emplace_back(t.creator_order[i], chrom);
        }
        std::sort(first_seen.begin(), first_seen.end());
        for (auto &fs : first_seen) {
            missing_genomic_warned_.insert(fs.second);
            *warn_ << "Warning: genomic sequence not found for " << fs.second << " splice junctions not available" << std::endl;
        }
    }
    // std::map<JunctionKey,...> order: chromosome NAME, start, end (junctions.hh:77-100).  Rows
    // arrive ordered by (first index of the name, start, end): a stable sort by name finishes it.
    if (trace_fh_ || !pt_pending_.empty()) {     // both need table_ in the device's row order
        const std::vector<ipx_record> recs = export_records();
        if (trace_fh_) write_trace(t, recs);
        if (!pt_pending_.empty()) write_pending_pass_through(t, recs);
    }
    // (through the names' ranks: the same order as comparing the strings, see junctions.cc)
    {
        std::map<std::string, uint32_t> rank_of;
        for (const std::string &name : targets_) rank_of.emplace(name, 0u);
        uint32_t next = 0;
        for (auto &kv : rank_of) kv.second = next++;
        std::vector<uint32_t> rank_of_target(targets_.size());
        for (size_t k = 0; k < targets_.size(); ++k) rank_of_target[k] = rank_of[targets_[k]];
        std::vector<uint32_t> idx(table_.size());
        for (size_t i = 0; i < idx.size(); ++i) idx[i] = (uint32_t)i;
        std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return rank_of_target[t.target[a]] < rank_of_target[t.target[b]]; });
        JunctionVector sorted;
        sorted.reserve(table_.size());
        for (uint32_t i : idx) sorted.push_back(std::move(table_[i]));
        table_.swap(sorted);
    }
    table_valid_ = true;
}

// --debug-trace: one line per record a junction counted, in input order.  create_junction
// (junctions_extractor.cc:451-457) prints the junction's (corrected) strand and an empty column,
// update_junction (:471-477) the read's strand and the junction's; both the junction's motif.
// A table that is finalized again after more alignments only adds the lines of the new alignments.
// the handle's records in the order the reference's loop met them
std::vector<ipx_record> JunctionsExtractor::export_records() {
    ensure_device();
    uint64_t n = 0;
    int rc = ipx_records_pending(h_, &n);
    if (rc != IPX_OK) throw_last_error(rc);
    std::vector<ipx_record> recs(n);
    if (n) {
        rc = ipx_records_export(h_, recs.data(), n, &n);
        if (rc != IPX_OK) throw_last_error(rc);
    }
    std::sort(recs.begin(), recs.end(), [](const ipx_record &a, const ipx_record &b) { return a.order < b.order; });
    return recs;
}

// row of the junction that counted record r, or n_junctions: the key has no row (no record of it
// passed the filter) or r came before the row's creator (left no trace)
static uint64_t counting_row(const ipx_table &t, const ipx_record &r) {
    uint64_t lo = 0, hi = t.n_junctions;               // rows are ordered by (target, start, end)
    while (lo < hi) {
        const uint64_t mid = (lo + hi) / 2;
        const bool less = t.target[mid] != r.target ? t.target[mid] < r.target
                          : t.intron_start[mid] != r.intron_start ? t.intron_start[mid] < r.intron_start
                                                                   : t.intron_end[mid] < r.intron_end;
        if (less) lo = mid + 1; else hi = mid;
    }
    if (lo >= t.n_junctions || t.target[lo] != r.target || t.intron_start[lo] != r.intron_start ||
        t.intron_end[lo] != r.intron_end || IPX_RECORD_SEQ(r.order) < t.creator_order[lo])
        return t.n_junctions;
    return lo;
}

// record_junction's orientCnt (junctions_extractor.cc:526-528) per waiting alignment, then
// update_strand_tag + write_pass_through (:540-547,647-650)
void JunctionsExtractor::PendingSpool::add(const AlnRecord &rec, uint64_t read_index) {
    if (n == 0) {
        first_read_index = read_index;
        const char *e = getenv("IPX_PASS_THROUGH_BUFFER_MB");
        if (e && atol(e) >= 0) cap_bytes = (size_t)atol(e) << 20;
    }
    const uint32_t hdr[2] = {(uint32_t)rec.raw.size(), rec.aux_off};
    buf.append(reinterpret_cast<const char *>(hdr), sizeof hdr);
    buf.append(rec.raw);
    ++n;
    if (buf.size() > cap_bytes) {                   // the memory share is full: the rest waits on disk
        if (!spill && !(spill = std::tmpfile())) throw std::runtime_error("cannot create a temporary file for the waiting pass-through alignments");
        if (fwrite(buf.data(), 1, buf.size(), spill) != buf.size()) throw std::runtime_error("Error writing the waiting pass-through alignments to their temporary file");
        spilled_bytes += buf.size();
        buf.clear();
    }
}

template <typename F> void JunctionsExtractor::PendingSpool::drain(F fn) {
    uint64_t k = 0;
    std::string raw;
    auto walk = [&](const char *p, size_t len) -> size_t {      // whole records of [p, p + len): bytes consumed
        size_t off = 0;
        while (off + 8 <= len) {
            uint32_t hdr[2];
            memcpy(hdr, p + off, 8);
            if (off + 8 + hdr[0] > len) break;
            raw.assign(p + off + 8, hdr[0]);
            fn(raw, hdr[1], k++);
            off += 8 + (size_t)hdr[0];
        }
        return off;
    };
    if (spill) {
        rewind(spill);
        std::string chunk;
        uint64_t left = spilled_bytes;
        const size_t step = (size_t)64 << 20;
        while (left) {
            const size_t old = chunk.size(), want = (size_t)std::min<uint64_t>(left, step);
            chunk.resize(old + want);
            if (fread(&chunk[old], 1, want, spill) != want) throw std::runtime_error("Error reading the waiting pass-through alignments back");
            left -= want;
            const size_t used = walk(chunk.data(), chunk.size());
            chunk.erase(0, used);
        }
        if (!chunk.empty()) throw std::runtime_error("Error reading the waiting pass-through alignments back");
        fclose(spill);
        spill = nullptr;
        spilled_bytes = 0;
    }
    if (walk(buf.data(), buf.size()) != buf.size() || k != n) throw std::runtime_error("Error reading the waiting pass-through alignments back");
    buf.clear();
    buf.shrink_to_fit();
    n = 0;
}

void JunctionsExtractor::write_pending_pass_through(const ipx_table &t, const std::vector<ipx_record> &recs) {
    // update_strand_tag looks at the SIGN of orientCnt only (:540-547): one saturating byte per waiting alignment
    std::vector<int8_t> orient(pt_pending_.n, 0);
    // read indices grow by one per processed alignment and every waiting alignment was processed,
    // so the waiting ones hold consecutive indices
    const uint64_t first = pt_pending_.first_read_index;
    for (const ipx_record &r : recs) {
        const uint64_t read = IPX_RECORD_READ(r.order);
        if (read < first || read - first >= orient.size()) continue;
        const uint64_t row = counting_row(t, r);
        if (row == t.n_junctions) continue;
        const Junction &j = table_[row];
        // Junction::is_canonical, junctions.hh:187-191: canonical sites are capitalized
        if (!j.splice_sites.empty() && j.splice_sites[0] >= 'A' && j.splice_sites[0] <= 'Z') {
            int8_t &o = orient[read - first];
            if (j.strand == '+') { if (o < 127) ++o; } else { if (o > -127) --o; }
        }
    }
    AlnRecord rec;
    pt_pending_.drain([&](const std::string &raw, uint32_t aux_off, uint64_t k) {
        rec.raw = raw;
        rec.aux_off = aux_off;
        pass_through_->write(rec, orient[k], set_xs_, set_ts_);
    });
}

void JunctionsExtractor::write_trace(const ipx_table &t, const std::vector<ipx_record> &recs) {
    for (const ipx_record &r : recs) {
        const uint64_t read = IPX_RECORD_READ(r.order);
        if (read < trace_done_) continue;
        const uint64_t lo = counting_row(t, r);
        if (lo == t.n_junctions) continue;
        const uint64_t seq = IPX_RECORD_SEQ(r.order);
        const Junction &j = table_[lo];
        const std::string &qname = trace_qname_[read - trace_base_];
        const unsigned flag = trace_flag_[read - trace_base_];
        *trace_fh_ << j.chrom << "\t" << r.intron_start << "\t" << r.intron_end << "\t" << qname << "\t" << flag << "\t";
        if (seq == t.creator_order[lo]) *trace_fh_ << j.strand << "\t";
        else *trace_fh_ << IPX_RECORD_STRAND(r.order) << "\t" << j.strand;
        *trace_fh_ << "\t" << j.splice_sites << "\t" << r.anchor_start << "\t" << r.anchor_end << std::endl;
    }
    trace_done_ = next_read_index_;
}

// JunctionTable::get_junctions, junctions.hh:232-238
JunctionVector JunctionsExtractor::get_junctions() {
    ScopeTimer scope_timer{t_getj_, now_s()};
    if (!table_valid_) finalize_table();
    return table_;
}

// JunctionTable::clear via JunctionsExtractor::clear (junctions_extractor.hh:223-225); ijunc_ keeps counting
void JunctionsExtractor::clear() {
    if (!table_valid_) finalize_table();
    ijunc_ += (uint32_t)table_.size();
    table_.clear();
    table_valid_ = true;          // an empty table is the table of no records
    ensure_device();
    int rc = g_ ? ipx_group_clear(g_) : ipx_clear(h_);
    if (rc != IPX_OK) throw_last_error(rc);
    trace_qname_.clear();
    trace_flag_.clear();
    trace_base_ = trace_done_ = next_read_index_;
}

}  // namespace ipxhost
