// ipx_api.hip -- the C ABI of include/ipx.h: handle, device memory, streams,
// and the order in which the kernels of ipx_scan/ipx_sort/ipx_reduce run.
//
// Stands where the reference's JunctionsExtractor object stands
// (src/junctions_extractor.hh:73-256): construction with by-value thresholds,
// a target list, an optional genome, process_alignment over the input, and a
// junction table handed to the printers.  There is no CPU path in this
// library: every entry point needs a gfx950 device.
#include "ipx_common.h"
#include "ipx_reduce.h"

#include <algorithm>
#include <chrono>
#include <dlfcn.h>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace {

// entries of each of the two lists that close the gaps between the record stripes (ipx_scan.hip); a
// balanced launch leaves a few tens of thousands
static const uint64_t CLOSE_CAP = 1u << 20;

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    template <typename T> T *as() const { return (T *)p; }
};

struct Arena {
    char *base = nullptr;
    size_t off = 0, cap = 0;
    bool measuring = true;
    template <typename T> T *take(uint64_t count) {
        const size_t bytes = ((size_t)count * sizeof(T) + 255) & ~(size_t)255;
        T *r = measuring ? nullptr : (T *)(base + off);
        off += bytes;
        return r;
    }
};

}  // namespace

// event-pair profiler: a timeline of marks on the handle's stream
struct Profiler {
    bool on = false;
    int only = 0;                  // > 0: time the launches of this kernel class alone (an event before, one after)
    std::vector<hipEvent_t> pool;
    std::vector<int> ids;          // id of mark k (interval k-1 -> k is charged to ids[k])
    double total_ms[IPXK_COUNT] = {};
    uint64_t launches[IPXK_COUNT] = {};
    void mark(hipStream_t s, int id) {
        if (only && id != only && id != IPXK_BEGIN) return;
        if (only && id == IPXK_BEGIN) return;          // stage starts: the interval opens at pre()
        record(s, id);
    }
    void pre(hipStream_t s, int id) {
        if (only && id == only) record(s, IPXK_BEGIN);
    }
    void record(hipStream_t s, int id) {
        if (ids.size() == pool.size()) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return;
            pool.push_back(e);
        }
        if (hipEventRecord(pool[ids.size()], s) == hipSuccess) ids.push_back(id);
    }
    // call after the stream was synchronized
    void collect() {
        for (size_t k = 1; k < ids.size(); ++k) {
            if (ids[k] == IPXK_BEGIN) continue;
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, pool[k - 1], pool[k]) == hipSuccess) {
                total_ms[ids[k]] += ms;
                launches[ids[k]] += 1;
            }
        }
        ids.clear();
    }
    void reset() {
        ids.clear();
        for (int i = 0; i < IPXK_COUNT; ++i) { total_ms[i] = 0; launches[i] = 0; }
    }
    ~Profiler() { for (auto e : pool) (void)hipEventDestroy(e); }
};

static thread_local Profiler *g_prof = nullptr;
void ipx_prof_mark(hipStream_t s, int id) {
    if (g_prof && g_prof->on) g_prof->mark(s, id);
}
void ipx_prof_pre(hipStream_t s, int id) {
    if (g_prof && g_prof->on) g_prof->pre(s, id);
}
struct ProfScope {
    explicit ProfScope(Profiler *p) { g_prof = p; }
    ~ProfScope() { g_prof = nullptr; }
};

// a batch whose copies and scan are queued but not yet booked (ipx_push_batch_async)
struct PendingPush {
    bool active = false;
    ipx_batch db{};              // the batch as the device sees it (staged copy or the caller's device arrays)
    uint64_t n = 0, nw = 0;
    ScanParams sp{};
    Counters before{};           // counters to return to if the scan has to be repeated
    bool use_ops = false, use_stripes = true;
    int attempt = 0;
    bool is_bam = false;         // the batch came as BAM records (ipx_push_bam): the decode's error word counts
};

struct ipx_handle {
    ipx_params params{};
    Profiler prof;
    hipStream_t stream = nullptr;      // where everything is queued: the handle's own stream, or the caller's (ipx_use_stream)
    hipStream_t own_stream = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    // targets
    std::vector<std::string> names;
    std::vector<int64_t> target_len;
    std::vector<uint32_t> rank_of_tid, tid_of_rank;
    uint32_t n_ranks = 0;
    DevBuf d_rank_of_tid, d_tid_of_rank;
    KeyLayout key{};
    // genome
    bool have_genome = false;
    std::vector<std::string> gnames;
    std::vector<int64_t> glen;
    DevBuf d_twobit, d_mask, d_seq_len, d_seq_word2, d_seq_wordm, d_seq_of_rank, d_rank_seq;
    std::vector<uint64_t> g_word2, g_wordm;      // first twobit / mask word of each genome sequence
    // records
    DevBuf items_a, items_b, pays, cnts, dig_a, dig_b;
    bool opt_combine = true;       // IPX_OPT_COMBINE: the short-read scan may merge equal-key records of a tile
    uint32_t bsort_skip = 0;       // finalizes that go straight to the radix passes after the bucket sort overflowed
    // the bit runs on which the creation orders of the last finalize's junctions differed (ipx_finalize ranks the NEXT
    // table's through them before it knows that table's own: see "speculated" there)
    RankBits rank_hint{};
    bool rank_hint_ok = false;
    unsigned long long spl_host[64] = {};      // the splitters the device copy holds (ipx_records_partition_slots)
    bool spl_valid = false;
    void *spl_where = nullptr;
    uint64_t n_raw = 0;            // junction records that passed junction_qc (>= n_records once records were combined)
    uint32_t key_or[3] = {0, 0, 0}, key_and[3] = {~0u, ~0u, ~0u};   // host copy of Counters::key_or/and
    uint64_t rec_cap = 0, n_records = 0, n_alignments = 0;
    uint64_t seen_words = 0, seen_records = 0;     // for the capacity estimate
    uint64_t reserved_records = 0;
    bool force_lane_scan = true;   // false with IPX_SCAN=ops in the environment: try ipx_scan_ops.hip first
    unsigned long long err_word_host = IPX_ERRW_NONE;
    bool digs_stale = false;     // dig_a no longer holds the first-pass digits of items_a (after a sort)
    // Key bits of the digit bytes that scan / adopt write next to the records.  Fixed while the handle
    // holds records; an empty handle takes `first_hint`: the window the last sort would have liked
    // (the lowest 8 varying key bits), so that pass saves a whole pass over bits 0..7 that hardly vary.
    DigitSpec first_window{0, 8, 0, 0}, first_hint{0, 8, 0, 0};
    DevBuf counters, stage, sort_ws, scratch, out_arena, rank_bitmap, d_small, out_arena2, d_xchg, close_ws, defer, bam_raw, bs_ws;
    DevBuf bz_comp, bz_out, bz_tab, bz_soa, bz_cigar, bz_ws;      // ipx_decode_bgzf: file bytes, inflated stream, tables, the decoded batch
    // pinned host
    void *pin_small = nullptr;     // 4 KiB of read-back scalars
    uint32_t *mail_host = nullptr, *mail_dev = nullptr;      // k_publish's mailbox (pinned, host-coherent)
    uint32_t mail_seq = 0;
    void *pin_out = nullptr, *pin_out2 = nullptr;
    size_t pin_out_cap = 0, pin_out2_cap = 0;
    ipx_table table{};
    ipx_error err{};
    float push_ms = 0.f, fin_ms = 0.f;
    bool fin_timing_pending = false;
    PendingPush pending;
    uint32_t rest_hint = 0;        // alignments the last short-CIGAR scan left to its second stage
    // a read-back that was begun but not yet waited for (read_back_begin / read_back_end)
    uint32_t rb_seq = 0, rb_words = 0;
    bool rb_by_copy = false;
};

namespace {

int fail(ipx_handle *h, int status, const char *fmt, const char *a = "", long b = 0) {
    if (h) {
        h->err.status = status;
        h->err.order = ~0ull;
        snprintf(h->err.detail, sizeof h->err.detail, fmt, a, b);
    }
    return status;
}

int fail_hip(ipx_handle *h, hipError_t e, const char *what, int line) {
    if (h) {
        h->err.status = (e == hipErrorOutOfMemory) ? IPX_ENOMEM : IPX_EDEVICE;
        h->err.order = ~0ull;
        snprintf(h->err.detail, sizeof h->err.detail, "%s: %s (ipx_api.hip:%d)", what, hipGetErrorString(e), line);
        return h->err.status;
    }
    return IPX_EDEVICE;
}

// Waiting for the stream: the pipeline reads a few words back between its stages (record count and key
// masks after the scan, segment counts after the sort, table sizes at the end), each a wait of well under
// a millisecond.  Polling hipStreamQuery returns sooner than the blocking hipStreamSynchronize (which may
// sleep until an interrupt).
static hipError_t ipx_stream_wait(hipStream_t s) {
    hipError_t e;
    while ((e = hipStreamQuery(s)) == hipErrorNotReady) {
    }
    return e;
}

// smallest index of a block whose status is not 0, + 1, into *first (0: none); *first starts at 0
__global__ void k_first_bad_block(const uint32_t *__restrict__ status, uint32_t n, uint32_t *first) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && status[i]) {
        // (0 means none: take the minimum over i + 1 with 0 standing for "unset")
        uint32_t old = *first;
        while (old == 0 || old > i + 1) {
            const uint32_t prev = atomicCAS(first, old, i + 1);
            if (prev == old) break;
            old = prev;
        }
    }
}

// the handle's counters, set in stream order (ipx_clear, re-scans)
__global__ void k_set_counters(Counters *dst, Counters c) {
    if (threadIdx.x == 0) *dst = c;
}

// Reading a few words back between stages: a copy command, its completion signal and the host's stream
// query cost 25-45 us per stage.  Instead the LAST kernel of the stage is k_publish: it writes the words
// straight into pinned, host-coherent memory and then a sequence number, and the host spins on that number
// in its own memory.  Falls back to the stream wait if the number does not arrive (a faulted kernel).
constexpr uint32_t MAIL_WORDS = 128, MAIL_SEQ = 128;         // words 0..127 payload, word 128 the sequence number
__global__ void k_publish(const uint32_t *__restrict__ a, uint32_t na, const uint32_t *__restrict__ b, uint32_t nb,
                          uint32_t *__restrict__ mail, uint32_t seq) {
    const uint32_t t = threadIdx.x;
    if (t < na) mail[t] = a[t];
    else if (t < na + nb) mail[t] = b[t - na];
    __threadfence_system();
    __syncthreads();
    if (t == 0) __hip_atomic_store(&mail[MAIL_SEQ], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

#define HIPCHK(call)                                                  \
    do {                                                              \
        hipError_t e__ = (call);                                      \
        if (e__ != hipSuccess) return fail_hip(h, e__, #call, __LINE__); \
    } while (0)

// words a[0..na) and b[0..nb) (device) -> host once everything queued on the stream before is done; in two
// halves so that a caller can leave between them (ipx_push_batch_async): begin queues the publication,
// end waits for it and copies the words to dst.  One read-back at a time.
int read_back_begin(ipx_handle *h, const void *a, uint32_t na, const void *b = nullptr, uint32_t nb = 0) {
    h->rb_words = na + nb;
    h->rb_by_copy = na + nb > MAIL_WORDS;          // more than the mailbox holds: a copy and a stream wait
    if (h->rb_by_copy) {
        if (na) HIPCHK(hipMemcpyAsync(h->pin_small, a, (size_t)na * 4, hipMemcpyDeviceToHost, h->stream));
        if (nb) HIPCHK(hipMemcpyAsync((uint32_t *)h->pin_small + na, b, (size_t)nb * 4, hipMemcpyDeviceToHost, h->stream));
        return IPX_OK;
    }
    h->rb_seq = ++h->mail_seq;
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(MAIL_WORDS), 0, h->stream, (const uint32_t *)a, na, (const uint32_t *)b, nb,
                       h->mail_dev, h->rb_seq);
    HIPCHK(hipGetLastError());
    return IPX_OK;
}

int read_back_end(ipx_handle *h, void *dst) {
    if (h->rb_by_copy) {
        HIPCHK(ipx_stream_wait(h->stream));
        if (dst != h->pin_small) memcpy(dst, h->pin_small, (size_t)h->rb_words * 4);
        return IPX_OK;
    }
    bool arrived = false;
    for (uint64_t spin = 0; spin < (1ull << 26); ++spin) {
        if (__atomic_load_n(&h->mail_host[MAIL_SEQ], __ATOMIC_ACQUIRE) == h->rb_seq) { arrived = true; break; }
    }
    if (!arrived) HIPCHK(hipStreamSynchronize(h->stream));          // reports the fault, if that is what happened
    memcpy(dst, h->mail_host, (size_t)h->rb_words * 4);
    return IPX_OK;
}

int read_back(ipx_handle *h, void *dst, const void *a, uint32_t na, const void *b = nullptr, uint32_t nb = 0) {
    int rc = read_back_begin(h, a, na, b, nb);
    if (rc) return rc;
    return read_back_end(h, dst);
}

int ensure(ipx_handle *h, DevBuf &b, size_t bytes, bool keep = false, size_t keep_bytes = 0) {
    if (bytes <= b.cap) return IPX_OK;
    size_t want = keep ? std::max(bytes, b.cap + b.cap / 2) : bytes;
    void *np = nullptr;
    hipError_t e = hipMalloc(&np, want);
    if (e != hipSuccess && want > bytes) { want = bytes; e = hipMalloc(&np, want); }
    if (e != hipSuccess) return fail_hip(h, e, "hipMalloc", __LINE__);
    if (keep && b.p && keep_bytes) {
        e = hipMemcpyAsync(np, b.p, keep_bytes, hipMemcpyDeviceToDevice, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) { (void)hipFree(np); return fail_hip(h, e, "hipMemcpy(grow)", __LINE__); }
    }
    if (b.p) (void)hipFree(b.p);
    b.p = np;
    b.cap = want;
    return IPX_OK;
}

void release(DevBuf &b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

uint32_t bits_for(uint64_t max_value) {
    uint32_t b = 1;
    while (b < 64 && (max_value >> b)) ++b;
    return b;
}

int compute_layout(ipx_handle *h) {
    KeyLayout k{};
    k.len_bits = bits_for(h->params.max_intron_length);
    k.start_bits = 32;
    k.rank_bits = bits_for(h->n_ranks ? h->n_ranks - 1 : 0);
    k.total_bits = 16 + k.len_bits + k.start_bits + k.rank_bits;
    if (k.total_bits > 96)
        return fail(h, IPX_EINVAL, "sort key needs %s%ld bits (> 96): too many targets for this max_intron_length", "", (long)k.total_bits);
    k.n_digits = (k.total_bits + 7) / 8;
    h->key = k;
    return IPX_OK;
}

int upload(ipx_handle *h, DevBuf &b, const void *src, size_t bytes) {
    int rc = ensure(h, b, bytes ? bytes : 4);
    if (rc) return rc;
    if (bytes) HIPCHK(hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, h->stream));
    return IPX_OK;
}

int rebuild_ranks(ipx_handle *h) {
    const uint32_t n = (uint32_t)h->names.size();
    h->rank_of_tid.assign(n, 0);
    h->tid_of_rank.clear();
    if (h->params.scope == IPX_SCOPE_TARGET) {
        for (uint32_t i = 0; i < n; ++i) { h->rank_of_tid[i] = i; h->tid_of_rank.push_back(i); }
    } else {
        // std::map<JunctionKey> orders chromosomes as std::string (junctions.hh:77-83)
        std::vector<uint32_t> order(n);
        for (uint32_t i = 0; i < n; ++i) order[i] = i;
        std::stable_sort(order.begin(), order.end(),
                         [&](uint32_t a, uint32_t b) { return h->names[a] < h->names[b]; });
        for (uint32_t k = 0; k < n; ++k) {
            const uint32_t t = order[k];
            if (k == 0 || h->names[t] != h->names[order[k - 1]]) h->tid_of_rank.push_back(t);   // first tid with the name
            h->rank_of_tid[t] = (uint32_t)h->tid_of_rank.size() - 1;
        }
    }
    h->n_ranks = (uint32_t)h->tid_of_rank.size();
    int rc = compute_layout(h);
    if (rc) return rc;
    if ((rc = upload(h, h->d_rank_of_tid, h->rank_of_tid.data(), n * sizeof(uint32_t)))) return rc;
    if ((rc = upload(h, h->d_tid_of_rank, h->tid_of_rank.data(), h->n_ranks * sizeof(uint32_t)))) return rc;
    // genome sequence per rank
    if (h->have_genome) {
        std::vector<int32_t> sor(h->n_ranks ? h->n_ranks : 1, -1);
        for (uint32_t r = 0; r < h->n_ranks; ++r) {
            const std::string &nm = h->names[h->tid_of_rank[r]];
            for (size_t g = 0; g < h->gnames.size(); ++g)
                if (h->gnames[g] == nm) { sor[r] = (int32_t)g; break; }
        }
        if ((rc = upload(h, h->d_seq_of_rank, sor.data(), sor.size() * sizeof(int32_t)))) return rc;
        std::vector<RankSeq> rs(sor.size());
        for (size_t r = 0; r < sor.size(); ++r) {
            const int32_t g = sor[r];
            rs[r] = g < 0 ? RankSeq{-1, 0, 0, 0} : RankSeq{(long long)h->glen[g], h->g_word2[g], h->g_wordm[g], 0};
        }
        if ((rc = upload(h, h->d_rank_seq, rs.data(), rs.size() * sizeof(RankSeq)))) return rc;
    }
    HIPCHK(ipx_stream_wait(h->stream));
    return IPX_OK;
}

int reset_counters(ipx_handle *h) {
    Counters c{};
    c.n_records = h->n_records;
    c.err_word = IPX_ERRW_NONE;
    for (int k = 0; k < 3; ++k) { c.key_or[k] = 0; c.key_and[k] = ~0u; h->key_or[k] = 0; h->key_and[k] = ~0u; }
    h->err_word_host = IPX_ERRW_NONE;
    // stream-ordered, by a one-wave kernel that takes the values as its argument: a copy from this stack frame
    // would be staged and waited for (25-80 us per ipx_clear)
    hipLaunchKernelGGL(k_set_counters, dim3(1), dim3(64), 0, h->stream, h->counters.as<Counters>(), c);
    HIPCHK(hipGetLastError());
    return IPX_OK;
}

int grow_records(ipx_handle *h, uint64_t cap) {
    if (cap <= h->rec_cap) return IPX_OK;
    if (cap > IPX_MAX_RECORDS) return fail(h, IPX_EINVAL, "more than 2^31-1 junction records in one table%s%ld", "", 0);
    int rc;
    if ((rc = ensure(h, h->items_a, cap * sizeof(uint4), true, h->n_records * sizeof(uint4)))) return rc;
    if ((rc = ensure(h, h->pays, cap * sizeof(uint4), true, h->n_records * sizeof(uint4)))) return rc;
    if ((rc = ensure(h, h->cnts, cap * sizeof(uint32_t), true, h->n_records * sizeof(uint32_t)))) return rc;
    if ((rc = ensure(h, h->dig_a, cap + 64, true, h->n_records))) return rc;
    h->rec_cap = std::min(std::min(std::min(h->items_a.cap, h->pays.cap) / sizeof(uint4), h->cnts.cap / sizeof(uint32_t)),
                          h->dig_a.cap - 64);
    h->rec_cap = std::min<uint64_t>(h->rec_cap, IPX_MAX_RECORDS);       // an item's index word has 31 bits
    return IPX_OK;
}

void fill_data_error(ipx_handle *h, unsigned long long w) {
    ipx_error &e = h->err;
    e.status = IPX_EDATA;
    e.order = w;
    e.read_index = IPX_ORD_READ(w);
    e.target = 0xffffffffu;
    const uint32_t kind = (uint32_t)(w >> 8) & 3u, detail = (uint32_t)w & 0xffu;
    switch (kind) {
        case IPX_ERRK_CIGAR:
            e.kind = IPX_DATA_CIGAR;
            e.op = (char)detail;
            snprintf(e.detail, sizeof e.detail, "Unknown cigar operation '%c'", (char)detail);
            break;
        case IPX_ERRK_GENOME:
            e.kind = detail ? IPX_DATA_GENOME_SHORT : IPX_DATA_GENOME_MISSING;
            break;
        case IPX_ERRK_CATEGORY:
            e.kind = IPX_DATA_CATEGORY;
            snprintf(e.detail, sizeof e.detail, "should never look at FQCAIL, FUNMAP, FDUP reads");
            break;
        default:
            e.kind = IPX_DATA_TARGET;
            snprintf(e.detail, sizeof e.detail, "%s",
                     detail == 1 ? "more than 2^20-1 junction records in one alignment"
                     : detail == 2 ? "alignment spans 2^32 or more reference bases"
                     : detail == 3 ? "alignment's CIGAR words lie outside the batch's CIGAR array"
                                   : "alignment tid is not a valid target index");
            break;
    }
}

GenomeDev genome_dev(const ipx_handle *h) {
    GenomeDev g{};
    if (h->have_genome) {
        g.twobit = h->d_twobit.as<uint32_t>(); g.mask = h->d_mask.as<uint32_t>();
        g.seq_len = h->d_seq_len.as<int64_t>(); g.seq_word2 = h->d_seq_word2.as<uint64_t>();
        g.seq_wordm = h->d_seq_wordm.as<uint64_t>(); g.seq_of_rank = h->d_seq_of_rank.as<int32_t>();
        g.rank_seq = h->d_rank_seq.as<RankSeq>();
    }
    return g;
}

}  // namespace

// the one-lane-per-operation scan (csrc/ipx_scan_ops.hip), from libipx_scanops.so beside this library; null: not there
typedef hipError_t (*scanops_fn)(hipStream_t, const ScanParams *, const ipx_batch *, const uint32_t *, uint4 *, uint4 *, uint32_t *,
                                 uint8_t *, uint64_t, Counters *);
static scanops_fn scanops_entry() {
    static const scanops_fn fn = []() -> scanops_fn {
        Dl_info me;
        if (!dladdr(reinterpret_cast<const void *>(&ipx_abi_version), &me) || !me.dli_fname) return nullptr;
        std::string path(me.dli_fname);
        const size_t slash = path.rfind('/');
        path = (slash == std::string::npos ? std::string() : path.substr(0, slash + 1)) + "libipx_scanops.so";
        void *lib = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
        return lib ? reinterpret_cast<scanops_fn>(dlsym(lib, "ipx_scanops_launch")) : nullptr;
    }();
    return fn;
}

extern "C" {

static int push_complete(ipx_handle *h);      // books a batch that ipx_push_batch_async left in flight
static int report_data_error(ipx_handle *h, unsigned long long err_word);

uint32_t ipx_abi_version(void) { return IPX_ABI_VERSION; }

int ipx_create(const ipx_params *params, ipx_handle **out) {
    if (!params || !out) return IPX_EINVAL;
    *out = nullptr;
    // IPX_DEBUG_PLAN: where the time of opening the device goes (it is a quarter of the command line's wall time)
    const bool laps = getenv("IPX_DEBUG_PLAN") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!laps) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "ipx_create: %-32s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return IPX_ENODEVICE;
    lap("hipGetDeviceCount (runtime init)");
    if (params->device < 0 || params->device >= ndev) return IPX_ENODEVICE;
    if (hipSetDevice(params->device) != hipSuccess) return IPX_ENODEVICE;
    lap("hipSetDevice");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, params->device) != hipSuccess) return IPX_ENODEVICE;
    lap("hipGetDeviceProperties");
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return IPX_ENODEVICE;   // code objects are gfx950 only
    if (params->sj_filter < 0 || params->sj_filter > 2 || params->strandness < 0 || params->strandness > 2 ||
        params->scope < 0 || params->scope > 1)
        return IPX_EINVAL;
    ipx_handle *h = new (std::nothrow) ipx_handle();
    if (!h) return IPX_ENOMEM;
    h->params = *params;
    {
        // The per-operation scan (ipx_scan_ops.hip) is correct but, in its round-1 form, slower than
        // the lane-per-alignment kernel (dependent loads per 64-word chunk): opt in with IPX_SCAN=ops.
        const char *e = getenv("IPX_SCAN");
        h->force_lane_scan = !(e && strcmp(e, "ops") == 0);
        // (round 4: that kernel lives in libipx_scanops.so, next to this library, and is loaded only when asked for)
        // a request for it that cannot be met is an error, not a quiet switch to the other kernel
        if (!h->force_lane_scan && !scanops_entry()) {
            fprintf(stderr, "ipx: IPX_SCAN=ops, but libipx_scanops.so is not beside libipx.so (build it: make -C intronprospector_amd/csrc)\n");
            delete h;
            return IPX_EINVAL;
        }
    }
    if (hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking) != hipSuccess) { delete h; return IPX_EDEVICE; }
    lap("stream");
    h->stream = h->own_stream;
    for (auto &e : h->ev)
        if (hipEventCreate(&e) != hipSuccess) { ipx_destroy(h); return IPX_EDEVICE; }
    if (hipHostMalloc(&h->pin_small, 4096, hipHostMallocDefault) != hipSuccess) { ipx_destroy(h); return IPX_ENOMEM; }
    {
        void *m = nullptr, *d = nullptr;
        if (hipHostMalloc(&m, 4096, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) { ipx_destroy(h); return IPX_ENOMEM; }
        memset(m, 0, 4096);
        h->mail_host = (uint32_t *)m;
        if (hipHostGetDevicePointer(&d, m, 0) != hipSuccess) { ipx_destroy(h); return IPX_EDEVICE; }
        h->mail_dev = (uint32_t *)d;
    }
    lap("events, pinned words");
    int rc = ensure(h, h->counters, IPX_COUNTERS_BYTES);      // layout: ipx_common.h
    if (!rc && hipMemsetAsync(h->counters.p, 0, IPX_COUNTERS_BYTES, h->stream) != hipSuccess) rc = IPX_EDEVICE;
    if (!rc) rc = ensure(h, h->close_ws, 2 * CLOSE_CAP * sizeof(unsigned long long));
    if (!rc) rc = ensure(h, h->d_small, 64 * sizeof(uint32_t));
    lap("device words");
    if (!rc) rc = reset_counters(h);
    if (laps) (void)hipStreamSynchronize(h->stream);
    lap("first kernel (code objects)");
    if (!rc) rc = rebuild_ranks(h);
    if (rc) { ipx_destroy(h); return rc; }
    lap("ranks");
    *out = h;
    return IPX_OK;
}

int ipx_destroy(ipx_handle *h) {
    if (!h) return IPX_OK;
    (void)hipSetDevice(h->params.device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    h->pending.active = false;
    for (DevBuf *b : {&h->d_rank_of_tid, &h->d_tid_of_rank, &h->d_twobit, &h->d_mask, &h->d_seq_len,
                      &h->d_seq_word2, &h->d_seq_wordm, &h->d_seq_of_rank, &h->d_rank_seq, &h->items_a, &h->items_b, &h->pays, &h->cnts, &h->dig_a, &h->dig_b,
                      &h->counters, &h->stage, &h->sort_ws, &h->scratch, &h->out_arena,
                      &h->rank_bitmap, &h->d_small, &h->out_arena2, &h->d_xchg, &h->close_ws, &h->defer, &h->bam_raw, &h->bs_ws,
                      &h->bz_comp, &h->bz_out, &h->bz_tab, &h->bz_soa, &h->bz_cigar, &h->bz_ws})
        release(*b);
    if (h->pin_small) (void)hipHostFree(h->pin_small);
    if (h->mail_host) (void)hipHostFree(h->mail_host);
    if (h->pin_out) (void)hipHostFree(h->pin_out);
    if (h->pin_out2) (void)hipHostFree(h->pin_out2);
    for (auto &e : h->ev)
        if (e) (void)hipEventDestroy(e);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
    return IPX_OK;
}

int ipx_set_targets(ipx_handle *h, uint32_t n, const char *const *names, const int64_t *lengths) {
    if (!h || (n && !names)) return IPX_EINVAL;
    { const int prc = push_complete(h); if (prc) return prc; }
    if (h->n_records) {
        // Records already carry key ranks.  In target scope rank == tid, so a list that only
        // APPENDS targets (the reference's save_targets on a further open()) keeps them valid.
        bool extends = h->params.scope == IPX_SCOPE_TARGET && n >= h->names.size();
        for (size_t i = 0; extends && i < h->names.size(); ++i) extends = names[i] && h->names[i] == names[i];
        if (!extends)
            return fail(h, IPX_EINVAL, "ipx_set_targets with records pending: only appending targets in target scope is possible%s%ld", "", 0);
    }
    (void)hipSetDevice(h->params.device);
    h->names.clear();
    h->target_len.clear();
    for (uint32_t i = 0; i < n; ++i) {
        h->names.emplace_back(names[i] ? names[i] : "");
        h->target_len.push_back(lengths ? lengths[i] : -1);
    }
    return rebuild_ranks(h);
}

int ipx_set_genome_2bit(ipx_handle *h, uint32_t n_seqs, const char *const *names, const int64_t *lengths,
                        const uint8_t *const *twobit, const uint8_t *const *mask) {
    if (!h) return IPX_EINVAL;
    if (n_seqs && (!names || !lengths || !twobit || !mask)) return IPX_EINVAL;
    { const int prc = push_complete(h); if (prc) return prc; }
    // pending records were combined (or not) according to the genome that was set when they were pushed
    if (h->n_records) return fail(h, IPX_EINVAL, "ipx_set_genome_2bit with records pending%s%ld", "", 0);
    (void)hipSetDevice(h->params.device);
    h->gnames.clear();
    h->glen.clear();
    h->have_genome = n_seqs > 0;
    if (!h->have_genome) return rebuild_ranks(h);
    std::vector<uint64_t> w2(n_seqs), wm(n_seqs);
    uint64_t tot2 = 0, totm = 0;
    for (uint32_t i = 0; i < n_seqs; ++i) {
        if (lengths[i] < 0) return IPX_EINVAL;
        h->gnames.emplace_back(names[i]);
        h->glen.push_back(lengths[i]);
        w2[i] = tot2; wm[i] = totm;
        tot2 += ((uint64_t)lengths[i] + 15) / 16;
        totm += ((uint64_t)lengths[i] + 31) / 32;
    }
    int rc;
    // (+ 2 spare words: ipx_motif_of reads pairs of words)
    if ((rc = ensure(h, h->d_twobit, (tot2 + 2) * 4))) return rc;
    if ((rc = ensure(h, h->d_mask, (totm + 2) * 4))) return rc;
    HIPCHK(hipMemsetAsync(h->d_twobit.p, 0, (tot2 + 2) * 4, h->stream));
    HIPCHK(hipMemsetAsync(h->d_mask.p, 0, (totm + 2) * 4, h->stream));
    for (uint32_t i = 0; i < n_seqs; ++i) {
        const uint64_t L = (uint64_t)lengths[i];
        if (!L) continue;
        HIPCHK(hipMemcpyAsync(h->d_twobit.as<uint32_t>() + w2[i], twobit[i], (L + 3) / 4, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->d_mask.as<uint32_t>() + wm[i], mask[i], (L + 7) / 8, hipMemcpyHostToDevice, h->stream));
    }
    if ((rc = upload(h, h->d_seq_len, h->glen.data(), n_seqs * sizeof(int64_t)))) return rc;
    if ((rc = upload(h, h->d_seq_word2, w2.data(), n_seqs * sizeof(uint64_t)))) return rc;
    if ((rc = upload(h, h->d_seq_wordm, wm.data(), n_seqs * sizeof(uint64_t)))) return rc;
    h->g_word2 = w2;
    h->g_wordm = wm;
    HIPCHK(ipx_stream_wait(h->stream));      // caller memory is borrowed until return
    return rebuild_ranks(h);
}

int ipx_reserve(ipx_handle *h, uint64_t n_alignments, uint64_t n_cigar_words, uint64_t n_records) {
    if (!h) return IPX_EINVAL;
    (void)hipSetDevice(h->params.device);
    int rc;
    if ((rc = ensure(h, h->stage, n_alignments * 20 + n_cigar_words * 4 + 8 * 256))) return rc;
    h->reserved_records = n_records;
    if ((rc = grow_records(h, h->n_records + n_records))) return rc;
    if ((rc = ensure(h, h->items_b, h->rec_cap * sizeof(uint4)))) return rc;
    if ((rc = ensure(h, h->dig_b, h->rec_cap + 64))) return rc;
    if ((rc = ensure(h, h->sort_ws, ipx_sort_workspace_bytes(h->rec_cap)))) return rc;
    return IPX_OK;
}

// ---- ipx_push_batch, split in two so that the caller can overlap its own work (decoding the next batch)
// with the copy and the scan: push_begin queues the H2D copies, the scan and the publication of its counters;
// push_complete waits for them, runs the second stage if the scan listed alignments for it, repeats the scan
// if the record arrays were too small, and books the batch.  ipx_push_batch = begin + complete;
// ipx_push_batch_async = complete(previous) + begin; every other entry point completes a pending push first.
static int launch_scan_attempt(ipx_handle *h) {
    PendingPush &pp = h->pending;
    ipx_prof_mark(h->stream, IPXK_BEGIN);
    HIPCHK(hipEventRecord(h->ev[0], h->stream));
    if (pp.use_ops) {
        HIPCHK(scanops_entry()(h->stream, &pp.sp, &pp.db, h->d_rank_of_tid.as<uint32_t>(), h->items_a.as<uint4>(),
                               h->pays.as<uint4>(), h->cnts.as<uint32_t>(), h->dig_a.as<uint8_t>(), h->rec_cap,
                               h->counters.as<Counters>()));
        ipx_prof_mark(h->stream, IPXK_SCAN);
    } else
        HIPCHK(ipx_launch_scan(h->stream, pp.sp, pp.db, h->d_rank_of_tid.as<uint32_t>(), h->items_a.as<uint4>(),
                               h->pays.as<uint4>(), h->cnts.as<uint32_t>(), h->dig_a.as<uint8_t>(), h->rec_cap, h->counters.as<Counters>(),
                               h->n_records, pp.use_stripes ? h->close_ws.as<unsigned long long>() : nullptr, CLOSE_CAP,
                               h->defer.as<uint32_t>(), h->rest_hint));
    HIPCHK(hipEventRecord(h->ev[1], h->stream));
    // [defer count, -, merged records (64 bit)]: IPX_CNT_DEFER and IPX_CNT_MERGED are neighbours
    const uint32_t *defer_n_dev = reinterpret_cast<const uint32_t *>(h->counters.as<char>() + IPX_CNT_DEFER);
    return read_back_begin(h, h->counters.p, sizeof(Counters) / 4, defer_n_dev, 4);
}

static int push_launch(ipx_handle *h);

static int push_begin(ipx_handle *h, const ipx_batch *b) {
    const uint64_t n = b->n_alignments, nw = b->n_cigar_words;
    if (!b->tid || !b->pos || !b->flag || !b->xs || !b->nh || !b->cigar_off || !b->n_cigar || (nw && !b->cigar))
        return fail(h, IPX_EINVAL, "ipx_push_batch: null array%s%ld", "", 0);
    if (b->base_read_index + n > IPX_ORD_MAX_READ)
        return fail(h, IPX_EINVAL, "read index beyond 2^34-1%s%ld", "", 0);
    if (nw >= (1ull << 32)) return fail(h, IPX_EINVAL, "more than 2^32-1 CIGAR words in one batch%s%ld", "", 0);
    int rc;
    PendingPush &pp = h->pending;
    pp.db = *b;
    pp.n = n;
    pp.nw = nw;
    if (b->memory == IPX_MEM_HOST) {
        // one staging buffer, 256-byte aligned sections
        auto al = [](uint64_t x) { return (x + 255) & ~255ull; };
        const uint64_t o_tid = 0, o_pos = al(o_tid + n * 4), o_flag = al(o_pos + n * 4), o_xs = al(o_flag + n * 2),
                       o_nh = al(o_xs + n), o_off = al(o_nh + n), o_nc = al(o_off + n * 4), o_cig = al(o_nc + n * 4),
                       total = al(o_cig + nw * 4);
        if ((rc = ensure(h, h->stage, total))) return rc;
        char *s = h->stage.as<char>();
        HIPCHK(hipMemcpyAsync(s + o_tid, b->tid, n * 4, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(s + o_pos, b->pos, n * 4, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(s + o_flag, b->flag, n * 2, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(s + o_xs, b->xs, n, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(s + o_nh, b->nh, n, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(s + o_off, b->cigar_off, n * 4, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(s + o_nc, b->n_cigar, n * 4, hipMemcpyHostToDevice, h->stream));
        if (nw) HIPCHK(hipMemcpyAsync(s + o_cig, b->cigar, nw * 4, hipMemcpyHostToDevice, h->stream));
        ipx_batch &db = pp.db;
        db.tid = (const int32_t *)(s + o_tid); db.pos = (const int32_t *)(s + o_pos);
        db.flag = (const uint16_t *)(s + o_flag); db.xs = (const uint8_t *)(s + o_xs);
        db.nh = (const uint8_t *)(s + o_nh); db.cigar_off = (const uint32_t *)(s + o_off);
        db.n_cigar = (const uint32_t *)(s + o_nc); db.cigar = (const uint32_t *)(s + o_cig);
    } else if (b->memory != IPX_MEM_DEVICE) {
        return IPX_EINVAL;
    }
    pp.is_bam = false;
    return push_launch(h);
}

// the batch is in device memory (pp.db): make room for its records, set the scan up and queue it
static int push_launch(ipx_handle *h) {
    PendingPush &pp = h->pending;
    const uint64_t n = pp.n, nw = pp.nw;
    int rc;
    // capacity: a record needs at least two CIGAR words (xM yN zM -> 1 record per extra 2 words)
    const uint64_t upper = nw / 2;
    uint64_t guess = 2 * n + 1024;
    if (h->seen_words) guess = std::max<uint64_t>(guess, (uint64_t)((double)h->seen_records / (double)h->seen_words * (double)nw * 1.25) + 1024);
    uint64_t want = h->n_records + std::min(upper, guess);
    if ((rc = grow_records(h, want))) return rc;

    ScanParams &sp = pp.sp;
    sp = ScanParams{};
    sp.min_anchor_length = h->params.min_anchor_length;
    sp.max_anchor_indel_size = h->params.max_anchor_indel_size;
    sp.min_intron_length = h->params.min_intron_length;
    sp.max_intron_length = h->params.max_intron_length;
    sp.strandness = h->params.strandness;
    sp.exclude_multi = (h->params.excludes & IPX_EXCLUDE_MULTI) ? 1u : 0u;
    sp.n_targets = (uint32_t)h->names.size();
    sp.key = h->key;
    if (h->n_records == 0) h->first_window = h->first_hint;
    sp.first = h->first_window;
    // Combining a tile's equal-key records is exact when every record of the group is counted.  Without a
    // genome, or with a filter other than "canon" (splice_juncs.cc:52-61), every record is.  With an active
    // filter the records before a junction's creator are dropped one by one (ipx_reduce.hip), and a merged
    // record must not straddle the creator: only records that pass the filter themselves are merged (the
    // creator is the first of those, so a group of them lies wholly at or after it), the others stay single.
    // IPX_COMBINE=0 in the environment switches it off (tests compare both).
    const char *combine_e = getenv("IPX_COMBINE");                 // read per push: a test switches it between runs
    const bool combine_env = !(combine_e && combine_e[0] == '0');
    const bool filter_active = h->have_genome && h->params.sj_filter == IPX_SJ_CANON;
    sp.combine = (h->opt_combine && combine_env) ? (filter_active ? 2u : 1u) : 0u;
    sp.skip_missing = h->params.skip_missing_targets;
    sp.rank_is_tid = h->params.scope == IPX_SCOPE_TARGET ? 1u : 0u;
    if (filter_active) sp.genome = genome_dev(h);

    // state to return to if a scan has to be repeated
    Counters &before = pp.before;
    before = Counters{};
    before.n_records = h->n_records;
    before.err_word = h->err_word_host;
    for (int k = 0; k < 3; ++k) { before.key_or[k] = h->key_or[k]; before.key_and[k] = h->key_and[k]; }
    pp.use_ops = !h->force_lane_scan;
    pp.use_stripes = true;
    pp.attempt = 0;
    // list of the alignments the short-read scan hands on to k_scan_rest (at most all of them)
    if ((rc = ensure(h, h->defer, ipx_scan_list_words(n) * sizeof(uint32_t)))) return rc;
    ProfScope prof_scope(&h->prof);
    if ((rc = launch_scan_attempt(h))) return rc;
    pp.active = true;
    return IPX_OK;
}

// ---- ipx_push_bam: the batch arrives as BAM records; ipx_bam.hip turns them into the SoA the scan reads
static int push_begin_bam(ipx_handle *h, const ipx_bam_batch *b) {
    const uint64_t n = b->n_alignments, nw = b->n_cigar_words;
    if (!b->bytes || !b->span_begin || !b->span_first || b->n_spans == 0 || (b->n_tid_map && !b->tid_map))
        return fail(h, IPX_EINVAL, "ipx_push_bam: null array%s%ld", "", 0);
    if (b->memory != IPX_MEM_HOST) return fail(h, IPX_EINVAL, "ipx_push_bam: the records must be in host memory%s%ld", "", 0);
    if (b->base_read_index + n > IPX_ORD_MAX_READ)
        return fail(h, IPX_EINVAL, "read index beyond 2^34-1%s%ld", "", 0);
    if (nw >= (1ull << 32) || n >= (1ull << 32) - 2)
        return fail(h, IPX_EINVAL, "more than 2^32-1 CIGAR words or alignments in one batch%s%ld", "", 0);
    // the span tables must describe [0, n_bytes) and [0, n): the device trusts nothing else about them
    if (b->span_begin[0] != 0 || b->span_begin[b->n_spans] != b->n_bytes || b->span_first[0] != 0 || b->span_first[b->n_spans] != n)
        return fail(h, IPX_EINVAL, "ipx_push_bam: span tables do not cover the batch%s%ld", "", 0);
    for (uint32_t k = 0; k < b->n_spans; ++k)
        if (b->span_begin[k] > b->span_begin[k + 1] || b->span_first[k] > b->span_first[k + 1])
            return fail(h, IPX_EINVAL, "ipx_push_bam: span tables are not ascending%s%ld", "", 0);
    int rc;
    PendingPush &pp = h->pending;
    pp.n = n;
    pp.nw = nw;
    auto al = [](uint64_t x) { return (x + 255) & ~255ull; };
    // raw side: bytes (+ 8: ld32 reads the aligned word pair around any offset), span tables, tid map, workspace
    const uint64_t r_bytes = 0, r_sb = al(r_bytes + b->n_bytes + 8), r_sf = al(r_sb + 8ull * (b->n_spans + 1)),
                   r_map = al(r_sf + 8ull * (b->n_spans + 1)), r_ws = al(r_map + 4ull * b->n_tid_map),
                   r_total = al(r_ws + ipx_bam_workspace_bytes(n));
    if ((rc = ensure(h, h->bam_raw, r_total))) return rc;
    char *r = h->bam_raw.as<char>();
    HIPCHK(hipMemcpyAsync(r + r_bytes, b->bytes, b->n_bytes, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemsetAsync(r + r_bytes + b->n_bytes, 0, 8, h->stream));
    HIPCHK(hipMemcpyAsync(r + r_sb, b->span_begin, 8ull * (b->n_spans + 1), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(r + r_sf, b->span_first, 8ull * (b->n_spans + 1), hipMemcpyHostToDevice, h->stream));
    if (b->n_tid_map) HIPCHK(hipMemcpyAsync(r + r_map, b->tid_map, 4ull * b->n_tid_map, hipMemcpyHostToDevice, h->stream));
    // SoA side: the sections ipx_push_batch stages a host batch into
    const uint64_t o_tid = 0, o_pos = al(o_tid + n * 4), o_flag = al(o_pos + n * 4), o_xs = al(o_flag + n * 2),
                   o_nh = al(o_xs + n), o_off = al(o_nh + n), o_nc = al(o_off + n * 4), o_cig = al(o_nc + n * 4),
                   total = al(o_cig + nw * 4 + 16);
    if ((rc = ensure(h, h->stage, total))) return rc;
    char *s = h->stage.as<char>();
    ipx_batch &db = pp.db;
    db = ipx_batch{};
    db.n_alignments = n;
    db.n_cigar_words = nw;
    db.base_read_index = b->base_read_index;
    db.memory = IPX_MEM_DEVICE;
    db.tid = (const int32_t *)(s + o_tid); db.pos = (const int32_t *)(s + o_pos);
    db.flag = (const uint16_t *)(s + o_flag); db.xs = (const uint8_t *)(s + o_xs);
    db.nh = (const uint8_t *)(s + o_nh); db.cigar_off = (const uint32_t *)(s + o_off);
    db.n_cigar = (const uint32_t *)(s + o_nc); db.cigar = (const uint32_t *)(s + o_cig);
    uint32_t *bam_err = reinterpret_cast<uint32_t *>(h->counters.as<char>() + IPX_CNT_DEFER) + 1;
    HIPCHK(ipx_launch_bam_decode(h->stream, (const uint8_t *)(r + r_bytes), (const uint64_t *)(r + r_sb), (const uint64_t *)(r + r_sf),
                                 b->n_spans, n, nw, b->n_tid_map ? (const int32_t *)(r + r_map) : nullptr, b->n_tid_map,
                                 (int32_t *)(s + o_tid), (int32_t *)(s + o_pos), (uint16_t *)(s + o_flag), (uint8_t *)(s + o_xs),
                                 (uint8_t *)(s + o_nh), (uint32_t *)(s + o_off), (uint32_t *)(s + o_nc), (uint32_t *)(s + o_cig),
                                 r + r_ws, bam_err));
    pp.is_bam = true;
    return push_launch(h);
}

static int push_complete(ipx_handle *h) {
    PendingPush &pp = h->pending;
    if (!pp.active) return IPX_OK;
    pp.active = false;
    (void)hipSetDevice(h->params.device);
    ProfScope prof_scope(&h->prof);
    int rc;
    for (;;) {
        if ((rc = read_back_end(h, h->pin_small))) return rc;
        h->prof.collect();
        const Counters *c = (const Counters *)h->pin_small;
        const uint32_t *tail4 = (const uint32_t *)h->pin_small + sizeof(Counters) / 4;
        const uint64_t merged = pp.use_ops ? 0ull : ((uint64_t)tail4[2] | ((uint64_t)tail4[3] << 32));
        const bool layout_bad = pp.use_ops && c->pad[0] != 0;
        const bool stripes_bad = c->pad[1] != 0;                 // more gaps between the stripes than the lists hold
        const uint32_t n_defer = pp.use_ops ? 0u : tail4[0];
        if (pp.is_bam && tail4[1] != 0xffffffffu) {
            // what the host parser throws "Error reader BAM record" for; the scan's records of this batch are dropped
            const uint32_t bad = tail4[1];
            const unsigned long long earlier = c->err_word;       // earliest data error so far, this batch's scan included
            hipLaunchKernelGGL(k_set_counters, dim3(1), dim3(64), 0, h->stream, h->counters.as<Counters>(), pp.before);
            HIPCHK(hipGetLastError());
            // The reference's loop throws at the FIRST defect in input order (junctions_extractor.cc:620-664): a data
            // error of an alignment before the record that does not parse is the one to report
            if (earlier != IPX_ERRW_NONE && ((earlier >> 8) & 3u) != IPX_ERRK_GENOME &&
                IPX_ORD_READ(earlier) < pp.db.base_read_index + (uint64_t)bad - 1)
                return report_data_error(h, earlier);
            return fail(h, IPX_EINVAL, "ipx_push_bam: malformed BAM record%s (alignment %ld of the batch)", "", (long)bad - 1);
        }
        if (getenv("IPX_DEBUG_PLAN"))
            fprintf(stderr, "ipx: scan of %llu alignments: %llu records, %llu merged away, %u alignments left to the second stage\n",
                    (unsigned long long)pp.n, (unsigned long long)(c->n_records - h->n_records), (unsigned long long)merged, n_defer);
        // (the second stage ran behind the first on the device, see ipx_launch_scan; the count sizes the next one's grid)
        h->rest_hint = n_defer;
        if (!layout_bad && !stripes_bad && c->n_records <= h->rec_cap) {
            for (int k = 0; k < 3; ++k) { h->key_or[k] = c->key_or[k]; h->key_and[k] = c->key_and[k]; }
            h->err_word_host = c->err_word;
            h->seen_words += pp.nw;
            h->seen_records += c->n_records - h->n_records;
            h->n_raw += c->n_records - h->n_records + merged;
            h->n_records = c->n_records;
            break;
        }
        if (pp.attempt == 4) return fail(h, IPX_EDEVICE, "record capacity still too small after regrow%s%ld", "", 0);
        // Either the per-operation scan cannot walk this CIGAR layout (words not back to back in
        // alignment order): repeat with the lane-per-alignment kernel.  Or the capacity estimate was
        // too small: records beyond it were counted but not stored; make room and scan again.
        const uint64_t need = c->n_records;
        hipLaunchKernelGGL(k_set_counters, dim3(1), dim3(64), 0, h->stream, h->counters.as<Counters>(), pp.before);
        HIPCHK(hipGetLastError());
        if (layout_bad) pp.use_ops = false;
        else if (stripes_bad) pp.use_stripes = false;
        else if ((rc = grow_records(h, need))) return rc;
        ++pp.attempt;
        if ((rc = launch_scan_attempt(h))) return rc;
    }
    (void)hipEventElapsedTime(&h->push_ms, h->ev[0], h->ev[1]);
    h->n_alignments += pp.n;
    return IPX_OK;
}

int ipx_push_batch(ipx_handle *h, const ipx_batch *b) {
    if (!h || !b) return IPX_EINVAL;
    (void)hipSetDevice(h->params.device);
    int rc = push_complete(h);
    if (rc) return rc;
    if (b->n_alignments == 0) return IPX_OK;
    if ((rc = push_begin(h, b))) return rc;
    return push_complete(h);
}

int ipx_push_batch_async(ipx_handle *h, const ipx_batch *b) {
    if (!h || !b) return IPX_EINVAL;
    (void)hipSetDevice(h->params.device);
    int rc = push_complete(h);              // at most one batch in flight
    if (rc) return rc;
    if (b->n_alignments == 0) return IPX_OK;
    return push_begin(h, b);
}

int ipx_push_bam(ipx_handle *h, const ipx_bam_batch *b) {
    if (!h || !b) return IPX_EINVAL;
    (void)hipSetDevice(h->params.device);
    int rc = push_complete(h);
    if (rc) return rc;
    if (b->n_alignments == 0) return IPX_OK;
    if ((rc = push_begin_bam(h, b))) return rc;
    return push_complete(h);
}

int ipx_push_bam_async(ipx_handle *h, const ipx_bam_batch *b) {
    if (!h || !b) return IPX_EINVAL;
    (void)hipSetDevice(h->params.device);
    int rc = push_complete(h);              // at most one batch in flight
    if (rc) return rc;
    if (b->n_alignments == 0) return IPX_OK;
    return push_begin_bam(h, b);
}

int ipx_sync(ipx_handle *h) {
    if (!h) return IPX_EINVAL;
    return push_complete(h);
}

int ipx_host_alloc(void **out, size_t bytes) {
    if (!out) return IPX_EINVAL;
    *out = nullptr;
    return hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault) == hipSuccess ? IPX_OK : IPX_ENOMEM;
}

int ipx_host_free(void *p) {
    if (p) (void)hipHostFree(p);
    return IPX_OK;
}

int ipx_clear(ipx_handle *h) {
    if (!h) return IPX_EINVAL;
    (void)hipSetDevice(h->params.device);
    (void)push_complete(h);                 // whatever the batch in flight held is dropped with the rest
    h->n_records = 0;
    h->n_raw = 0;
    h->n_alignments = 0;
    h->digs_stale = false;
    memset(&h->table, 0, sizeof h->table);
    return reset_counters(h);
}

int ipx_last_error(const ipx_handle *h, ipx_error *out) {
    if (!h || !out) return IPX_EINVAL;
    *out = h->err;
    return IPX_OK;
}

int ipx_last_timing(const ipx_handle *h, float *push_ms, float *finalize_ms) {
    if (!h) return IPX_EINVAL;
    if (push_ms) *push_ms = h->push_ms;
    if (finalize_ms) {
        ipx_handle *m = const_cast<ipx_handle *>(h);
        if (m->fin_timing_pending) {
            (void)hipEventSynchronize(m->ev[3]);
            (void)hipEventElapsedTime(&m->fin_ms, m->ev[2], m->ev[3]);
            m->fin_timing_pending = false;
        }
        *finalize_ms = h->fin_ms;
    }
    return IPX_OK;
}

// Shared by ipx_finalize and ipx_finalize_shard.
//   resort      : sort the records and count segment heads (false: reuse the state left by
//                 the previous call on the same records)
//   sj_filter   : filter the reduce applies (the shard's "all" table forces IPX_SJ_ALL)
//   out_arena   : device buffer that receives the table
static int finalize_core(ipx_handle *h, int32_t memory, bool resort, int32_t sj_filter, DevBuf &out_arena,
                         void **pin_out, size_t *pin_out_cap, ipx_table *out, unsigned long long *err_out) {
    memset(out, 0, sizeof *out);
    const uint64_t R = h->n_records;
    int rc;
    hipStream_t s = h->stream;
    uint32_t *small = (uint32_t *)h->pin_small;
    uint32_t J = 0, B = 0, n_valid = 0, n_hist = 0;
    bool ranked_early = false;      // the table's ijunc was queued before the last read-back, which therefore was the completion
    ReduceBuffers rb{};
    unsigned long long err_word = IPX_ERRW_NONE;

    ProfScope prof_scope(&h->prof);
    ipx_prof_mark(s, IPXK_BEGIN);
    if (R > 0) {
        // one workspace serves the sort (digit histograms) and, once the sort is done,
        // the per-chunk head counts of the reduce
        const uint64_t chunks = ipx_reduce_chunks(R);
        const uint64_t chunks_al = (chunks + 63) & ~63ull;
        const uint64_t sums1 = ipx_scan_sums_words(chunks, chunks);
        const size_t ws_bytes = std::max(ipx_sort_workspace_bytes(R),
                                         (size_t)((2 * chunks_al + sums1) * sizeof(uint32_t) + 1024));
        // one pass per (up to) 8 bits on which the keys differ (OR ^ AND).  The digit bytes that came
        // with the records serve the first pass if their window holds the lowest varying bits;
        // otherwise the first pass reads its digits from the records
        const uint32_t vary[3] = {h->key_or[0] ^ h->key_and[0], h->key_or[1] ^ h->key_and[1],
                                  h->key_or[2] ^ h->key_and[2]};
        const SortPlan ideal = ipx_plan_sort(vary, h->key.total_bits, nullptr);
        const bool bytes_ok = !h->digs_stale && ipx_window_is_first(vary, h->key.total_bits, h->first_window);
        const SortPlan with_bytes = ipx_plan_sort(vary, h->key.total_bits, &h->first_window);
        const bool use_bytes = bytes_ok && with_bytes.n <= ideal.n;
        const SortPlan plan = use_bytes ? with_bytes : ideal;
        h->first_hint = ideal.digits[0];
        if (getenv("IPX_DEBUG_PLAN")) {                      // the sort plan, for the curious
            fprintf(stderr, "ipx: %llu records, key bits %u, vary %08x %08x %08x, %u passes%s:", (unsigned long long)R,
                    h->key.total_bits, vary[2], vary[1], vary[0], plan.n, use_bytes ? " (first from the digit bytes)" : "");
            for (uint32_t k = 0; k < plan.n; ++k)
                fprintf(stderr, " [%u+%u,%u+%u]", plan.digits[k].lo1, plan.digits[k].w1, plan.digits[k].lo2, plan.digits[k].w2);
            fprintf(stderr, "\n");
        }
        uint32_t *tot = h->d_small.as<uint32_t>();      // [0] junctions, [1] bins, [2] the bucket sort's overflow flag
        // Which sort: the bucket sort (ipx_bsort.hip: every record moves through HBM three times whatever the key
        // width) from IPX_BSORT_MIN records on, the LSD radix sort below that, for keys without 16 spare bits, and
        // when a leaf of the bucket sort did not fit LDS.  Default 8 M records: measured on one MI355X the two are
        // level at 2 M records (S: 0.720 / 0.718 ms per step, fourteen launches against twenty-four), the bucket sort
        // is ahead at 10 M records in random order (S shuffled, round 5: sort 0.67 against 0.75 ms, step 1.27 against
        // 1.37 -- in round 3, before k_bs_assign ran four search chains a thread, it was 0.71 against 0.66 and the
        // default 16 M) and where the passes are bandwidth-bound (L, 40 M records: 7.22 / 7.37 ms per step).
        // IPX_SORT=lsd: always LSD; tests force either.
        const uint64_t bsort_min = [] {                   // read per call: tests switch between runs
            const char *m = getenv("IPX_SORT");
            if (m && m[0] == 'l') return ~0ull;
            const char *e = getenv("IPX_BSORT_MIN");
            return e ? (uint64_t)atoll(e) : (8ull << 20);
        }();
        bool by_buckets = resort && R >= bsort_min && ipx_bsort_applicable(R, h->key);
        if (by_buckets && h->bsort_skip) { --h->bsort_skip; by_buckets = false; }     // it overflowed recently on this handle
        auto lsd_sort = [&](bool bytes_valid, const SortPlan &pl) -> int {
            HIPCHK(ipx_launch_sort(s, h->items_a.as<uint4>(), h->items_b.as<uint4>(), h->dig_a.as<uint8_t>(),
                                   h->dig_b.as<uint8_t>(), bytes_valid, R, pl, h->sort_ws.p, h->sort_ws.cap));
            // the records now live (sorted) in one of the two buffers; keep that one as items_a
            // so that later pushes append to, and later sorts start from, a full copy
            if (pl.n & 1u) {
                std::swap(h->items_a, h->items_b);
                std::swap(h->dig_a, h->dig_b);
            }
            return IPX_OK;
        };
        if (resort) {
            if ((rc = ensure(h, h->items_b, h->rec_cap * sizeof(uint4)))) return rc;
            if ((rc = ensure(h, h->dig_b, h->rec_cap + 64))) return rc;
            if ((rc = ensure(h, h->sort_ws, ws_bytes))) return rc;
            // tot[2]: "the items are not sorted" -- raised by the bucket sort when a leaf did not fit LDS, and by k_count_heads
            // behind EVERY sort when two neighbouring items are out of order
            HIPCHK(hipMemsetAsync(tot + 2, 0, sizeof(uint32_t), s));
            if (by_buckets) {
                if ((rc = ensure(h, h->bs_ws, ipx_bsort_workspace_bytes()))) return rc;
                HIPCHK(ipx_launch_bsort(s, h->items_a.as<uint4>(), h->items_b.as<uint4>(), h->dig_a.as<uint8_t>(),
                                        h->dig_b.as<uint8_t>(), R, h->key, vary, h->sort_ws.p, h->sort_ws.cap, h->bs_ws.p, tot + 2));
            } else {
                if ((rc = lsd_sort(use_bytes, plan))) return rc;
            }
            // digit-0 bytes of the sorted order (a later re-sort starts with digit 0 again)
            h->digs_stale = true;
        }
        uint32_t *chunk_j = h->sort_ws.as<uint32_t>();
        uint32_t *chunk_b = chunk_j + chunks_al;
        uint32_t *sums = chunk_b + chunks_al;
        for (bool second_try = false;; second_try = true) {
            SortedItems si0{h->items_a.as<uint4>(), R};
            if (resort) {
                HIPCHK(ipx_launch_count_heads(s, si0, chunk_j, chunk_b, tot + 2));
                HIPCHK(ipx_launch_scan_u32_pair(s, chunk_j, chunk_j, chunks, tot + 0, chunk_b, chunk_b, chunks, tot + 1, sums));
            }
            if ((rc = read_back(h, small, tot, 3))) return rc;
            h->prof.collect();
            ipx_prof_mark(s, IPXK_BEGIN);
            if (!resort || small[2] == 0) break;
            // The items are not sorted: a leaf of the bucket sort did not fit LDS, or its ranks came out wrong (they rest on
            // the order in which the LDS atomics of one instruction return, ipx_bsort.hip) -- the records are all there,
            // keys intact, in some order -> the radix passes.  If THOSE leave them unsorted the device is not to be trusted.
            if (second_try || !by_buckets)
                return fail(h, IPX_EDEVICE, "ipx_finalize: the radix sort left the records unsorted%s%ld", "", 0);
            if (getenv("IPX_DEBUG_PLAN")) fprintf(stderr, "ipx: bucket sort overflowed a leaf, sorting %llu records by radix passes\n", (unsigned long long)R);
            by_buckets = false;
            h->bsort_skip = 15;
            HIPCHK(hipMemsetAsync(tot + 2, 0, sizeof(uint32_t), s));
            if ((rc = lsd_sort(false, ideal))) return rc;
        }
        SortedItems si{h->items_a.as<uint4>(), R};
        J = small[0];
        B = small[1];

        // stage 2: scratch and output sections, laid out twice (measure, then place)
        uint32_t *detail4 = nullptr;
        void *rank_ws = nullptr;
        auto layout = [&](Arena &tmp, Arena &outa) {
            rb.chunk_j = chunk_j; rb.chunk_b = chunk_b; rb.n_j = J; rb.n_b = B;
            rb.jrank = tmp.take<uint32_t>(J); rb.jstart = tmp.take<uint32_t>(J); rb.jend = tmp.take<uint32_t>(J);
            rb.jmotif = tmp.take<uint32_t>(J); rb.amin = tmp.take<uint32_t>(J); rb.amax = tmp.take<uint32_t>(J);
            rb.jvalid = tmp.take<uint32_t>(J); rb.jpos = tmp.take<uint32_t>(J); rb.jfirstbin = tmp.take<uint32_t>(J);
            rb.jstrand = tmp.take<uint8_t>(J);
            rb.first = tmp.take<unsigned long long>(J); rb.creator = tmp.take<unsigned long long>(J);
            rb.counts = tmp.take<unsigned long long>(3ull * J);
            rb.bin_off = tmp.take<uint32_t>(B); rb.bin_count = tmp.take<uint32_t>(B);
            rb.bin_flag = tmp.take<uint32_t>(B); rb.bin_pos = tmp.take<uint32_t>(B);
            rb.scan_sums = tmp.take<uint32_t>(ipx_scan_sums_words(J, B));
            rb.totals = tmp.take<uint32_t>(64);
            rb.rank_a = tmp.take<uint4>(J); rb.rank_b = tmp.take<uint4>(J);
            detail4 = tmp.take<uint32_t>(64);
            rank_ws = tmp.take<char>(ipx_sort_workspace_bytes(J));
            rb.out_target = outa.take<uint32_t>(J); rb.out_intron_start = outa.take<uint32_t>(J);
            rb.out_intron_end = outa.take<uint32_t>(J); rb.out_anchor_start = outa.take<uint32_t>(J);
            rb.out_anchor_end = outa.take<uint32_t>(J); rb.out_strand = outa.take<uint8_t>(J);
            rb.out_motif = outa.take<uint16_t>(J); rb.out_read_counts = outa.take<uint64_t>(3ull * J);
            rb.out_creator_order = outa.take<uint64_t>(J); rb.out_hist_begin = outa.take<uint64_t>((uint64_t)J + 1);
            rb.out_ijunc = outa.take<uint32_t>(J); rb.out_hist_value = outa.take<uint16_t>(B);
            rb.out_hist_count = outa.take<uint32_t>(B);
        };
        {
            Arena tmp, outa;
            layout(tmp, outa);
            if ((rc = ensure(h, h->scratch, tmp.off + 256))) return rc;
            if ((rc = ensure(h, out_arena, outa.off + 256))) return rc;
            Arena tmp2, outa2;
            tmp2.measuring = outa2.measuring = false;
            tmp2.base = h->scratch.as<char>();
            outa2.base = out_arena.as<char>();
            layout(tmp2, outa2);
        }
        const GenomeDev g = genome_dev(h);
        // no record stands for several (nothing was combined, nothing adopted with counts): the reduce takes a
        // record's count from its category bits and leaves the count words alone -- one random 128-byte line per
        // record less (40 M long-read records: the step 7.51 -> 7.35 ms)
        const uint32_t *cnts_or_null = h->n_raw == h->n_records ? nullptr : h->cnts.as<uint32_t>();
        HIPCHK(ipx_launch_reduce(s, si, h->pays.as<uint4>(), cnts_or_null, h->key, rb, g, sj_filter,
                                 h->params.skip_missing_targets, h->d_tid_of_rank.as<uint32_t>(),
                                 h->counters.as<Counters>()));
        // ijunc = rank of the creation order, through a bitmap over the bits on which the orders differ -- bits the host
        // learns from the read-back below, a round trip during which the GPU idles (~10 us of a 0.56 ms step).  Without a
        // genome nothing is dropped (n_valid == J), and a table's orders mostly differ on the bits the last table's did:
        // the rank kernels are queued BEFORE the read-back with the last finalize's runs, and stand if the read-back shows
        // that this table's varying bits lie inside them (bits that do not vary only widen the bitmap's index, the ranks
        // are the same); otherwise they run again, as before.
        const char *rank_env = getenv("IPX_RANK");                  // "sort": always the radix sort (tests)
        const bool force_sort = rank_env && rank_env[0] == 's';
        bool speculated = false;
        if (!h->have_genome && h->rank_hint_ok && J && !force_sort) {
            if ((rc = ensure(h, h->rank_bitmap, ipx_rank_bitmap_workspace_bytes(h->rank_hint.bits)))) return rc;
            HIPCHK(ipx_launch_rank_by_bitmap(s, J, rb.rank_a, h->rank_hint, h->rank_bitmap.as<uint32_t>(), rb.out_ijunc));
            speculated = true;
        }
        if ((rc = read_back(h, small, rb.totals, 8, h->counters.p, sizeof(Counters) / 4))) return rc;
        h->prof.collect();
        ipx_prof_mark(s, IPXK_BEGIN);
        n_valid = small[0];
        n_hist = small[1];
        err_word = ((const Counters *)(small + 8))->err_word;
        // ijunc: rank of the creation order (junctions_extractor.cc:459-460)
        if (n_valid) {
            // creation-order keys: only the bits on which they differ (OR ^ AND, small[3..6]) need passes
            const uint32_t rvary[3] = {small[3] ^ small[5], small[4] ^ small[6], 0u};
            // few varying bits in few runs: rank through a bitmap (5 small launches); else a radix sort
            RankBits rbits{};
            const uint64_t rv = (uint64_t)rvary[0] | ((uint64_t)rvary[1] << 32);
            bool by_bitmap = true;
            for (uint32_t b = 0; b < 64;) {
                if (!((rv >> b) & 1u)) { ++b; continue; }
                uint32_t w = 0;
                while (b + w < 64 && ((rv >> (b + w)) & 1u)) ++w;
                if (rbits.n_runs == 4) { by_bitmap = false; break; }
                rbits.lo[rbits.n_runs] = b;
                rbits.width[rbits.n_runs++] = w;
                rbits.bits += w;
                b += w;
            }
            uint64_t hint_mask = 0;
            for (uint32_t k = 0; k < h->rank_hint.n_runs; ++k)
                hint_mask |= (h->rank_hint.width[k] >= 64 ? ~0ull : ((1ull << h->rank_hint.width[k]) - 1ull)) << h->rank_hint.lo[k];
            const bool stands = speculated && n_valid == J && (rv & ~hint_mask) == 0;
            if (by_bitmap && rbits.bits <= 28) {
                // (the next table is ranked through the union of what was seen: a hint that only grows keeps standing)
                RankBits u{};
                const uint64_t um = h->rank_hint_ok ? (hint_mask | rv) : rv;
                bool fits = true;
                for (uint32_t b = 0; b < 64;) {
                    if (!((um >> b) & 1u)) { ++b; continue; }
                    uint32_t w = 0;
                    while (b + w < 64 && ((um >> (b + w)) & 1u)) ++w;
                    if (u.n_runs == 4) { fits = false; break; }
                    u.lo[u.n_runs] = b; u.width[u.n_runs++] = w; u.bits += w;
                    b += w;
                }
                if (fits && u.bits <= 28) { h->rank_hint = u; h->rank_hint_ok = true; }
                else { h->rank_hint = rbits; h->rank_hint_ok = true; }
            } else {
                h->rank_hint_ok = false;
            }
            if (stands) {
                ranked_early = true;
            } else if (by_bitmap && rbits.bits <= 28 && !force_sort) {
                if ((rc = ensure(h, h->rank_bitmap, ipx_rank_bitmap_workspace_bytes(rbits.bits)))) return rc;
                HIPCHK(ipx_launch_rank_by_bitmap(s, n_valid, rb.rank_a, rbits, h->rank_bitmap.as<uint32_t>(), rb.out_ijunc));
            } else {
                const SortPlan rp = ipx_plan_sort(rvary, 64, nullptr);
                HIPCHK(ipx_launch_sort(s, rb.rank_a, rb.rank_b, nullptr, nullptr, false, n_valid, rp, rank_ws,
                                       ipx_sort_workspace_bytes(J), IPXK_RANK_SORT));
                HIPCHK(ipx_launch_emit_ijunc(s, n_valid, (rp.n & 1u) ? rb.rank_b : rb.rank_a, rb.out_ijunc));
            }
        }
        if (err_word != IPX_ERRW_NONE && ((err_word >> 8) & 3u) == IPX_ERRK_GENOME) {
            HIPCHK(hipMemsetAsync(detail4, 0, 16, s));
            HIPCHK(ipx_launch_find_genome_error(s, J, rb.first, rb.jmotif, rb.jrank, rb.jstart, rb.jend,
                                                IPX_ORD_SEQ(err_word), detail4));
            HIPCHK(hipMemcpyAsync(small + 32, detail4, 16, hipMemcpyDeviceToHost, s));
        }
    } else {
        HIPCHK(hipMemcpyAsync(small + 4, h->counters.p, sizeof(Counters), hipMemcpyDeviceToHost, s));
        HIPCHK(ipx_stream_wait(s));
        err_word = ((const Counters *)(small + 4))->err_word;
    }

    ipx_table t{};
    t.n_junctions = n_valid;
    t.n_hist = n_hist;
    t.n_alignments = h->n_alignments;
    t.n_records = h->n_raw;
    t.memory = memory;
    if (memory == IPX_MEM_DEVICE || R == 0) {
        t.target = rb.out_target; t.intron_start = rb.out_intron_start; t.intron_end = rb.out_intron_end;
        t.anchor_start = rb.out_anchor_start; t.anchor_end = rb.out_anchor_end; t.strand = rb.out_strand;
        t.motif = rb.out_motif; t.read_counts = rb.out_read_counts; t.creator_order = rb.out_creator_order;
        t.ijunc = rb.out_ijunc; t.hist_begin = rb.out_hist_begin; t.hist_value = rb.out_hist_value;
        t.hist_count = rb.out_hist_count;
        if (!ranked_early && (rc = read_back(h, small + 48, nullptr, 0))) return rc;      // completion only
    } else {
        // pinned host copy, same section layout
        Arena ha;
        for (int pass = 0; pass < 2; ++pass) {
            ha.off = 0;
            ha.measuring = (pass == 0);
            ha.base = (char *)*pin_out;
            uint32_t *p_target = ha.take<uint32_t>(n_valid), *p_is = ha.take<uint32_t>(n_valid),
                     *p_ie = ha.take<uint32_t>(n_valid), *p_as = ha.take<uint32_t>(n_valid),
                     *p_ae = ha.take<uint32_t>(n_valid);
            uint8_t *p_strand = ha.take<uint8_t>(n_valid);
            uint16_t *p_motif = ha.take<uint16_t>(n_valid);
            uint64_t *p_counts = ha.take<uint64_t>(3ull * n_valid), *p_creator = ha.take<uint64_t>(n_valid),
                     *p_hb = ha.take<uint64_t>((uint64_t)n_valid + 1);
            uint32_t *p_ijunc = ha.take<uint32_t>(n_valid);
            uint16_t *p_hv = ha.take<uint16_t>(n_hist);
            uint32_t *p_hc = ha.take<uint32_t>(n_hist);
            if (pass == 0) {
                if (ha.off + 256 > *pin_out_cap) {
                    if (*pin_out) (void)hipHostFree(*pin_out);
                    *pin_out = nullptr;
                    *pin_out_cap = 0;
                    HIPCHK(hipHostMalloc(pin_out, ha.off + 256, hipHostMallocDefault));
                    *pin_out_cap = ha.off + 256;
                }
                continue;
            }
#define D2H(dst, src, count, T) \
    if (count) HIPCHK(hipMemcpyAsync(dst, src, (size_t)(count) * sizeof(T), hipMemcpyDeviceToHost, s))
            D2H(p_target, rb.out_target, n_valid, uint32_t);
            D2H(p_is, rb.out_intron_start, n_valid, uint32_t);
            D2H(p_ie, rb.out_intron_end, n_valid, uint32_t);
            D2H(p_as, rb.out_anchor_start, n_valid, uint32_t);
            D2H(p_ae, rb.out_anchor_end, n_valid, uint32_t);
            D2H(p_strand, rb.out_strand, n_valid, uint8_t);
            D2H(p_motif, rb.out_motif, n_valid, uint16_t);
            D2H(p_counts, rb.out_read_counts, 3ull * n_valid, uint64_t);
            D2H(p_creator, rb.out_creator_order, n_valid, uint64_t);
            D2H(p_hb, rb.out_hist_begin, (uint64_t)n_valid + 1, uint64_t);
            D2H(p_ijunc, rb.out_ijunc, n_valid, uint32_t);
            D2H(p_hv, rb.out_hist_value, n_hist, uint16_t);
            D2H(p_hc, rb.out_hist_count, n_hist, uint32_t);
#undef D2H
            HIPCHK(ipx_stream_wait(s));
            t.target = p_target; t.intron_start = p_is; t.intron_end = p_ie; t.anchor_start = p_as;
            t.anchor_end = p_ae; t.strand = p_strand; t.motif = p_motif; t.read_counts = p_counts;
            t.creator_order = p_creator; t.ijunc = p_ijunc; t.hist_begin = p_hb; t.hist_value = p_hv;
            t.hist_count = p_hc;
        }
    }
    h->prof.collect();
    *out = t;
    *err_out = err_word;
    return IPX_OK;
}

// turn the device's earliest-error word into ipx_error and the reference's message
static int report_data_error(ipx_handle *h, unsigned long long err_word) {
    uint32_t *small = (uint32_t *)h->pin_small;
    fill_data_error(h, err_word);
    if (h->err.kind == IPX_DATA_GENOME_MISSING || h->err.kind == IPX_DATA_GENOME_SHORT) {
        const uint32_t *d4 = small + 32;
        const uint32_t rank = d4[0];
        h->err.target = (rank < h->tid_of_rank.size()) ? h->tid_of_rank[rank] : 0xffffffffu;
        // the acceptor side [end-2,end) is fetched first (g++ evaluates the operands of
        // fetch(a) + "/" + fetch(b) right to left, junctions_extractor.cc:419-420)
        const int64_t e = (int64_t)(int32_t)d4[2], st = (int64_t)(int32_t)d4[1];
        int64_t L = -1;
        if (h->err.target != 0xffffffffu) {
            const std::string &nm = h->names[h->err.target];
            for (size_t g = 0; g < h->gnames.size(); ++g)
                if (h->gnames[g] == nm) { L = h->glen[g]; break; }
        }
        auto clamp_len = [&](int64_t beg, int64_t endi) -> int64_t {   // faidx_adjust_position
            if (endi < beg) beg = endi;
            if (beg < 0) beg = 0; else if (L <= beg) beg = L;
            if (endi < 0) endi = 0; else if (L <= endi) endi = L - 1;
            return endi + 1 - beg > 0 ? endi + 1 - beg : 0;
        };
        int64_t fs = e - 2, fe = e;
        int64_t got = (L >= 0) ? clamp_len(fs, fe - 1) : -2;
        if (L >= 0 && got == 2) { fs = st; fe = st + 2; got = clamp_len(fs, fe - 1); }
        h->err.start = (uint32_t)fs;
        h->err.end = (uint32_t)fe;
        h->err.retlen = (int32_t)got;
        const char *nm = (h->err.target != 0xffffffffu) ? h->names[h->err.target].c_str() : "?";
        if (h->err.kind == IPX_DATA_GENOME_MISSING)
            snprintf(h->err.detail, sizeof h->err.detail, "can't load genome sequence for: %s:%d-%d", nm, (int)fs, (int)fe);
        else
            snprintf(h->err.detail, sizeof h->err.detail, "unexpected length returned for: %s:%d-%d: %d", nm, (int)fs, (int)fe, (int)got);
    }
    return IPX_EDATA;
}

int ipx_finalize(ipx_handle *h, int32_t memory, ipx_table *out) {
    if (!h || !out || (memory != IPX_MEM_HOST && memory != IPX_MEM_DEVICE)) return IPX_EINVAL;
    (void)hipSetDevice(h->params.device);
    { const int prc = push_complete(h); if (prc) return prc; }
    memset(&h->err, 0, sizeof h->err);
    unsigned long long err_word = IPX_ERRW_NONE;
    (void)hipEventRecord(h->ev[2], h->stream);
    int rc = finalize_core(h, memory, true, h->params.sj_filter, h->out_arena, &h->pin_out, &h->pin_out_cap, out, &err_word);
    if (rc) return rc;
    (void)hipEventRecord(h->ev[3], h->stream);
    h->fin_timing_pending = true;          // read (and waited for) by ipx_last_timing: 8 us per call saved here
    h->table = *out;
    if (err_word != IPX_ERRW_NONE) return report_data_error(h, err_word);
    return IPX_OK;
}

static const char *const kKernelNames[IPXK_COUNT] = {
    "begin", "k_scan", "k_hist", "k_binscan", "k_skip", "k_scatter", "k_count_heads", "scan_u32",
    "k_write_heads", "k_motif", "memset", "k_reduce_a", "k_reduce", "k_finish_junctions", "k_bins",
    "k_emit_bins", "k_emit_junctions", "k_emit_ijunc", "copy", "k_partition", "k_adopt", "ijunc_rank_sort", "k_bs_sample",
    "k_bs_assign", "k_bs_bounds", "k_bs_leaf"};

int ipx_profile(ipx_handle *h, int enable) {
    if (!h) return IPX_EINVAL;
    h->prof.on = enable != 0;
    h->prof.only = 0;
    h->prof.reset();
    return IPX_OK;
}

int ipx_profile_only(ipx_handle *h, const char *kernel) {
    if (!h || !kernel) return IPX_EINVAL;
    for (int i = 1; i < IPXK_COUNT; ++i) {
        if (strcmp(kernel, kKernelNames[i]) != 0) continue;
        if (i != IPXK_SCAN && i != IPXK_SORT_SCATTER && i != IPXK_PHASE_B) break;     // the bracketed classes
        h->prof.on = true;
        h->prof.only = i;
        h->prof.reset();
        return IPX_OK;
    }
    return fail(h, IPX_EINVAL, "ipx_profile_only: %s is not one of k_scan, k_scatter, k_reduce%ld", kernel, 0);
}

int ipx_profile_read(const ipx_handle *h, ipx_kernel_stat *out, uint32_t cap, uint32_t *n) {
    if (!h || !n) return IPX_EINVAL;
    uint32_t k = 0;
    for (int i = 1; i < IPXK_COUNT; ++i) {
        if (!h->prof.launches[i]) continue;
        if (out && k < cap) {
            memset(&out[k], 0, sizeof out[k]);
            snprintf(out[k].name, sizeof out[k].name, "%s", kKernelNames[i]);
            out[k].launches = h->prof.launches[i];
            out[k].total_ms = h->prof.total_ms[i];
        }
        ++k;
    }
    *n = k;
    return IPX_OK;
}

int ipx_finalize_shard(ipx_handle *h, int32_t memory, ipx_table *out_from_creator, ipx_table *out_all) {
    if (!h || !out_from_creator || !out_all || (memory != IPX_MEM_HOST && memory != IPX_MEM_DEVICE)) return IPX_EINVAL;
    (void)hipSetDevice(h->params.device);
    { const int prc = push_complete(h); if (prc) return prc; }
    memset(&h->err, 0, sizeof h->err);
    unsigned long long err_word = IPX_ERRW_NONE, err2 = IPX_ERRW_NONE;
    (void)hipEventRecord(h->ev[2], h->stream);
    int rc = finalize_core(h, memory, true, h->params.sj_filter, h->out_arena, &h->pin_out, &h->pin_out_cap,
                           out_from_creator, &err_word);
    if (rc) return rc;
    // Every record passes when there is no motif to test or the filter is not "canon":
    // then the two tables are the same table.
    const bool filter_active = h->have_genome && h->params.sj_filter == IPX_SJ_CANON;
    if (filter_active) {
        rc = finalize_core(h, memory, false, IPX_SJ_ALL, h->out_arena2, &h->pin_out2, &h->pin_out2_cap, out_all, &err2);
        if (rc) return rc;
    } else {
        *out_all = *out_from_creator;
    }
    (void)hipEventRecord(h->ev[3], h->stream);
    h->fin_timing_pending = true;          // read (and waited for) by ipx_last_timing: 8 us per call saved here
    h->table = *out_from_creator;
    if (err_word != IPX_ERRW_NONE) return report_data_error(h, err_word);
    return IPX_OK;
}

// ---- multi-GPU record exchange (ipx_exchange.hip) ------------------------------------
int ipx_records_pending(const ipx_handle *h, uint64_t *n_records) {
    if (!h || !n_records) return IPX_EINVAL;
    { const int prc = push_complete(const_cast<ipx_handle *>(h)); if (prc) return prc; }
    *n_records = h->n_records;
    return IPX_OK;
}

// host-side twin of unpack_key (ipx_common.h)
static void unpack_key_host(const uint32_t it[4], const KeyLayout &kl, uint32_t &rank, uint32_t &start, uint32_t &end,
                            uint32_t &off16) {
    const uint64_t lo = (uint64_t)it[0] | ((uint64_t)it[1] << 32);
    const uint64_t hi = it[2];
    off16 = (uint32_t)lo & 0xffffu;
    const uint32_t len = (uint32_t)((lo >> 16) & ((kl.len_bits >= 32) ? 0xffffffffull : ((1ull << kl.len_bits) - 1)));
    const uint32_t s1 = 16 + kl.len_bits, s2 = s1 + 32;
    uint64_t st = lo >> s1;
    if (s1 > 32) st |= hi << (64 - s1);
    start = (uint32_t)st;
    const uint64_t rk = (s2 < 64) ? ((lo >> s2) | (hi << (64 - s2))) : (hi >> (s2 - 64));
    rank = (uint32_t)(rk & ((kl.rank_bits >= 32) ? 0xffffffffull : ((1ull << kl.rank_bits) - 1)));
    end = start + len;
}

int ipx_records_export(ipx_handle *h, ipx_record *dst, uint64_t capacity, uint64_t *n_records) {
    if (!h || !n_records) return IPX_EINVAL;
    (void)hipSetDevice(h->params.device);
    { const int prc = push_complete(h); if (prc) return prc; }
    const uint64_t n = h->n_records;
    *n_records = n;
    if (n == 0) return IPX_OK;
    if (!dst || capacity < n) return IPX_EINVAL;
    if (h->n_raw != n)
        return fail(h, IPX_EINVAL, "ipx_records_export: records were combined; set IPX_OPT_COMBINE to 0 before the first push%s%ld", "", 0);
    std::vector<uint32_t> items(4 * n), pays(4 * n);
    HIPCHK(hipMemcpyAsync(items.data(), h->items_a.p, n * 16, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(pays.data(), h->pays.p, n * 16, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(ipx_stream_wait(h->stream));
    for (uint64_t i = 0; i < n; ++i) {
        const uint32_t *it = &items[4 * i];
        const uint32_t *py = &pays[4 * (uint64_t)IPX_ITEM_INDEX(it[3])];         // item.w = index of the record's payload
        uint32_t rank;
        ipx_record &r = dst[i];
        unpack_key_host(it, h->key, rank, r.intron_start, r.intron_end, r.read_offset);
        r.target = rank < h->tid_of_rank.size() ? h->tid_of_rank[rank] : rank;
        r.order = (uint64_t)py[0] | ((uint64_t)py[1] << 32);
        r.anchor_start = py[2];
        r.anchor_end = py[3];
    }
    return IPX_OK;
}

int ipx_records_sample(ipx_handle *h, uint32_t n_samples, uint64_t *out_keys) {
    if (!h || (n_samples && !out_keys) || n_samples > 4096) return IPX_EINVAL;
    (void)hipSetDevice(h->params.device);
    { const int prc = push_complete(h); if (prc) return prc; }
    if (!n_samples) return IPX_OK;
    int rc = ensure(h, h->d_xchg, (size_t)4096 * 8 + 2 * 64 * 8);
    if (rc) return rc;
    HIPCHK(ipx_launch_sample(h->stream, h->items_a.as<uint4>(), h->n_records, h->key, n_samples,
                             h->d_xchg.as<unsigned long long>()));
    HIPCHK(hipMemcpyAsync(out_keys, h->d_xchg.p, (size_t)n_samples * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(ipx_stream_wait(h->stream));
    return IPX_OK;
}

int ipx_records_partition(ipx_handle *h, uint32_t n_parts, const uint64_t *splitters, uint64_t *counts,
                          void *dst_records) {
    if (!h || !n_parts || n_parts > 64 || (n_parts > 1 && !splitters) || !counts || (h->n_records && !dst_records))
        return IPX_EINVAL;
    (void)hipSetDevice(h->params.device);
    { const int prc = push_complete(h); if (prc) return prc; }
    // d_xchg: 4096 sample keys | 64 splitters | 64 part totals | 64 words per partition block
    const uint32_t blocks = ipx_partition_blocks(h->n_records);
    int rc = ensure(h, h->d_xchg, ((size_t)4096 + 2 * 64 + (size_t)blocks * 64) * 8);
    if (rc) return rc;
    unsigned long long *d_spl = h->d_xchg.as<unsigned long long>() + 4096;
    unsigned long long *d_cnt = d_spl + 64;
    unsigned long long *d_blk = d_cnt + 64;
    unsigned long long host[64];
    memset(host, 0, sizeof host);
    for (uint32_t i = 0; i + 1 < n_parts; ++i) host[i] = splitters[i];
    ProfScope prof_scope(&h->prof);
    ipx_prof_mark(h->stream, IPXK_BEGIN);
    HIPCHK(hipMemcpyAsync(d_spl, host, sizeof host, hipMemcpyHostToDevice, h->stream));
    HIPCHK(ipx_launch_partition(h->stream, h->items_a.as<uint4>(), h->pays.as<uint4>(), h->cnts.as<uint32_t>(), h->n_records, h->key, n_parts,
                                d_spl, d_blk, d_cnt, (uint4 *)dst_records));
    if ((rc = read_back(h, h->pin_small, d_cnt, 64 * 2))) return rc;
    h->prof.collect();
    const unsigned long long *c = (const unsigned long long *)h->pin_small;
    for (uint32_t i = 0; i < n_parts; ++i) counts[i] = c[i];
    return IPX_OK;
}

int ipx_records_adopt(ipx_handle *h, uint64_t n, const void *records, uint64_t n_alignments) {
    if (!h || (n && !records)) return IPX_EINVAL;
    (void)hipSetDevice(h->params.device);
    { const int prc = push_complete(h); if (prc) return prc; }
    // the handle's own records are replaced; what its scan found wrong with the input stays (err_word):
    // the records this rank scanned went to their key-range owners, their errors did not
    const unsigned long long scan_err = h->err_word_host;
    h->n_records = 0;
    h->n_raw = 0;
    h->digs_stale = false;
    h->first_window = h->first_hint;
    int rc = reset_counters(h);
    if (rc) return rc;
    if (scan_err != IPX_ERRW_NONE) {
        Counters c{};
        c.n_records = 0;
        c.err_word = scan_err;
        for (int k = 0; k < 3; ++k) { c.key_or[k] = 0; c.key_and[k] = ~0u; }
        hipLaunchKernelGGL(k_set_counters, dim3(1), dim3(64), 0, h->stream, h->counters.as<Counters>(), c);
        HIPCHK(hipGetLastError());
        h->err_word_host = scan_err;
    }
    if ((rc = grow_records(h, n))) return rc;
    ProfScope prof_scope(&h->prof);
    ipx_prof_mark(h->stream, IPXK_BEGIN);
    unsigned long long *merged_dev = reinterpret_cast<unsigned long long *>(h->counters.as<char>() + IPX_CNT_MERGED);
    HIPCHK(hipMemsetAsync(merged_dev, 0, sizeof(unsigned long long), h->stream));
    HIPCHK(ipx_launch_adopt(h->stream, (const uint4 *)records, n, 0, h->items_a.as<uint4>(), h->pays.as<uint4>(),
                            h->cnts.as<uint32_t>(), h->dig_a.as<uint8_t>(), h->first_window, h->counters.as<Counters>()));
    if ((rc = read_back(h, h->pin_small, h->counters.p, sizeof(Counters) / 4, merged_dev, 2))) return rc;
    h->prof.collect();
    const Counters *c = (const Counters *)h->pin_small;
    const uint32_t *m2 = (const uint32_t *)h->pin_small + sizeof(Counters) / 4;
    for (int k = 0; k < 3; ++k) { h->key_or[k] = c->key_or[k]; h->key_and[k] = c->key_and[k]; }
    h->n_records = n;
    h->n_raw = n + ((uint64_t)m2[0] | ((uint64_t)m2[1] << 32));
    h->n_alignments = n_alignments;
    return IPX_OK;
}

int ipx_use_stream(ipx_handle *h, void *hip_stream) {
    if (!h) return IPX_EINVAL;
    (void)hipSetDevice(h->params.device);
    { const int prc = push_complete(h); if (prc) return prc; }
    HIPCHK(hipStreamSynchronize(h->stream));
    h->stream = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
    return IPX_OK;
}

int ipx_copy_async(ipx_handle *h, void *dst, const void *src, uint64_t bytes) {
    if (!h || (bytes && (!dst || !src))) return IPX_EINVAL;
    (void)hipSetDevice(h->params.device);
    if (!bytes) return IPX_OK;
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, h->stream));
    return IPX_OK;
}

// ---- global sjN across key ranges (include/ipx.h, ABI 5) ----------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void k_sorted_orders(uint32_t n, const uint64_t *__restrict__ creator_order,
                                                       const uint32_t *__restrict__ ijunc, uint64_t *__restrict__ out, uint64_t capacity) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < capacity; i += (uint64_t)gridDim.x * 256) {
        if (i < n) out[ijunc[i]] = creator_order[i];          // the table's own ijunc is the rank of the order: a scatter sorts
        else out[i] = ~0ull;
    }
}
// the number of smaller orders in n_lists ascending lists (padded with ~0): one binary search per list
__global__ __launch_bounds__(256) void k_global_ijunc(uint32_t n, const uint64_t *__restrict__ creator_order,
                                                      const uint64_t *__restrict__ lists, uint32_t n_lists, uint64_t stride,
                                                      uint64_t length, uint32_t *__restrict__ out) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const uint64_t o = creator_order[i];
        uint64_t rank = 0;
        for (uint32_t l = 0; l < n_lists; ++l) {
            const uint64_t *lst = lists + (uint64_t)l * stride;
            uint64_t lo = 0, hi = length;                      // first element >= o
            while (lo < hi) {
                const uint64_t mid = (lo + hi) >> 1;
                if (lst[mid] < o) lo = mid + 1; else hi = mid;
            }
            rank += lo;
        }
        out[i] = (uint32_t)rank;
    }
}
}  // namespace

int ipx_table_sorted_orders(ipx_handle *h, const ipx_table *table, uint64_t *out, uint64_t capacity) {
    if (!h || !table || (capacity && !out)) return IPX_EINVAL;
    if (table->memory != IPX_MEM_DEVICE) return fail(h, IPX_EINVAL, "ipx_table_sorted_orders: the table must be in device memory%s%ld", "", 0);
    if (table->n_junctions > capacity) return fail(h, IPX_EINVAL, "ipx_table_sorted_orders: %s%ld junctions do not fit the list", "", (long)table->n_junctions);
    (void)hipSetDevice(h->params.device);
    if (!capacity) return IPX_OK;
    const uint32_t blocks = (uint32_t)std::min<uint64_t>(ipx_div_up(capacity, 256), 4096);
    hipLaunchKernelGGL(k_sorted_orders, dim3(blocks), dim3(256), 0, h->stream, (uint32_t)table->n_junctions, table->creator_order,
                       table->ijunc, out, capacity);
    HIPCHK(hipGetLastError());
    return IPX_OK;
}

int ipx_table_global_ijunc(ipx_handle *h, const ipx_table *table, const uint64_t *lists, uint32_t n_lists, uint64_t stride,
                           uint64_t length, uint32_t *out_ijunc) {
    if (!h || !table || length > stride || (table->n_junctions && (!lists || !out_ijunc || !n_lists))) return IPX_EINVAL;
    if (table->memory != IPX_MEM_DEVICE) return fail(h, IPX_EINVAL, "ipx_table_global_ijunc: the table must be in device memory%s%ld", "", 0);
    (void)hipSetDevice(h->params.device);
    if (!table->n_junctions) return IPX_OK;
    const uint32_t blocks = (uint32_t)std::min<uint64_t>(ipx_div_up(table->n_junctions, 256), 4096);
    hipLaunchKernelGGL(k_global_ijunc, dim3(blocks), dim3(256), 0, h->stream, (uint32_t)table->n_junctions, table->creator_order, lists,
                       n_lists, stride, length, out_ijunc);
    HIPCHK(hipGetLastError());
    return IPX_OK;
}

// the splitters as the partition kernels want them (device copy, refreshed when they change)
static int stage_splitters(ipx_handle *h, uint32_t n_parts, const uint64_t *splitters, unsigned long long **d_spl,
                           unsigned long long **d_cnt, unsigned long long **d_blk, uint32_t blocks) {
    int rc = ensure(h, h->d_xchg, ((size_t)4096 + 2 * 64 + (size_t)blocks * 64) * 8);
    if (rc) return rc;
    *d_spl = h->d_xchg.as<unsigned long long>() + 4096;
    *d_cnt = *d_spl + 64;
    *d_blk = *d_cnt + 64;
    unsigned long long host[64];
    memset(host, 0, sizeof host);
    for (uint32_t i = 0; i + 1 < n_parts; ++i) host[i] = splitters[i];
    if (h->spl_valid && h->d_xchg.p == h->spl_where && memcmp(host, h->spl_host, sizeof host) == 0) return IPX_OK;
    // (a copy from this stack frame: staged by the runtime before the call returns)
    HIPCHK(hipMemcpyAsync(*d_spl, host, sizeof host, hipMemcpyHostToDevice, h->stream));
    memcpy(h->spl_host, host, sizeof host);
    h->spl_valid = true;
    h->spl_where = h->d_xchg.p;
    return IPX_OK;
}

int ipx_records_partition_slots(ipx_handle *h, uint32_t n_parts, const uint64_t *splitters, uint64_t slot_records,
                                uint64_t error_order, int32_t error_status, void *dst_slots) {
    return ipx_records_partition_slots_spill(h, n_parts, splitters, slot_records, error_order, error_status, dst_slots, nullptr, 0);
}

int ipx_records_partition_slots_spill(ipx_handle *h, uint32_t n_parts, const uint64_t *splitters, uint64_t slot_records,
                                      uint64_t error_order, int32_t error_status, void *dst_slots, void *dst_spill,
                                      uint64_t spill_records) {
    if (!h || !n_parts || n_parts > 64 || (n_parts > 1 && !splitters) || slot_records < 2 + (n_parts + 7) / 8 || !dst_slots) return IPX_EINVAL;
    if (!dst_spill) spill_records = 0;
    (void)hipSetDevice(h->params.device);
    { const int prc = push_complete(h); if (prc) return prc; }
    const uint32_t blocks = ipx_partition_blocks(h->n_records);
    unsigned long long *d_spl, *d_cnt, *d_blk;
    int rc = stage_splitters(h, n_parts, splitters, &d_spl, &d_cnt, &d_blk, blocks);
    if (rc) return rc;
    ProfScope prof_scope(&h->prof);
    ipx_prof_mark(h->stream, IPXK_BEGIN);
    HIPCHK(ipx_launch_partition_slots(h->stream, h->items_a.as<uint4>(), h->pays.as<uint4>(), h->cnts.as<uint32_t>(), h->n_records,
                                      h->key, n_parts, d_spl, d_blk, d_cnt, slot_records, error_order, (uint32_t)error_status,
                                      (uint4 *)dst_slots, spill_records ? (uint4 *)dst_spill : nullptr, spill_records));
    return IPX_OK;
}

int ipx_records_adopt_slots(ipx_handle *h, uint32_t n_slots, uint64_t slot_records, const void *slots, uint64_t n_alignments,
                            ipx_slots_info *info) {
    if (!h || !n_slots || n_slots > 64 || slot_records < 2 + (n_slots + 7) / 8 || !slots || !info) return IPX_EINVAL;
    (void)hipSetDevice(h->params.device);
    { const int prc = push_complete(h); if (prc) return prc; }
    const unsigned long long scan_err = h->err_word_host;       // what this rank's scan found wrong stays (see ipx_records_adopt)
    h->n_records = 0;
    h->n_raw = 0;
    h->digs_stale = false;
    h->first_window = h->first_hint;
    for (int k = 0; k < 3; ++k) { h->key_or[k] = 0; h->key_and[k] = ~0u; }
    Counters c{};
    c.n_records = 0;
    c.err_word = scan_err;
    for (int k = 0; k < 3; ++k) { c.key_or[k] = 0; c.key_and[k] = ~0u; }
    hipLaunchKernelGGL(k_set_counters, dim3(1), dim3(64), 0, h->stream, h->counters.as<Counters>(), c);
    HIPCHK(hipGetLastError());
    int rc;
    if ((rc = grow_records(h, (uint64_t)n_slots * slot_records + 1))) return rc;
    ProfScope prof_scope(&h->prof);
    ipx_prof_mark(h->stream, IPXK_BEGIN);
    unsigned long long *merged_dev = reinterpret_cast<unsigned long long *>(h->counters.as<char>() + IPX_CNT_MERGED);
    HIPCHK(hipMemsetAsync(merged_dev, 0, sizeof(unsigned long long), h->stream));
    // the headers' summary lands right behind the merged-records counter: one publication brings all of it back
    uint32_t *d_info = reinterpret_cast<uint32_t *>(h->counters.as<char>() + IPX_CNT_MERGED + 8);
    HIPCHK(ipx_launch_adopt_slots(h->stream, (const uint4 *)slots, n_slots, slot_records, 0, h->items_a.as<uint4>(), h->pays.as<uint4>(),
                                  h->cnts.as<uint32_t>(), h->dig_a.as<uint8_t>(), h->first_window, h->counters.as<Counters>(), d_info));
    if ((rc = read_back(h, h->pin_small, h->counters.p, sizeof(Counters) / 4, merged_dev, 2 + 13))) return rc;
    h->prof.collect();
    const Counters *cc = (const Counters *)h->pin_small;
    const uint32_t *m2 = (const uint32_t *)h->pin_small + sizeof(Counters) / 4;
    const unsigned long long merged = (unsigned long long)m2[0] | ((unsigned long long)m2[1] << 32);
    const uint32_t *w = m2 + 2;
    for (int k = 0; k < 3; ++k) { h->key_or[k] = cc->key_or[k]; h->key_and[k] = cc->key_and[k]; }
    info->n_records = (uint64_t)w[0] | ((uint64_t)w[1] << 32);
    info->max_part = (uint64_t)w[2] | ((uint64_t)w[3] << 32);
    info->overflow = w[4];
    info->error_slot = w[5];
    info->error_order = (uint64_t)w[6] | ((uint64_t)w[7] << 32);
    info->error_status = (int32_t)w[8];
    info->reserved = 0;
    info->max_received = (uint64_t)w[9] | ((uint64_t)w[10] << 32);
    info->all_received = (uint64_t)w[11] | ((uint64_t)w[12] << 32);
    h->n_records = info->n_records;
    h->n_raw = info->n_records + merged;
    h->n_alignments = n_alignments;
    return IPX_OK;
}

int ipx_records_adopt_more(ipx_handle *h, uint64_t n, const void *records) {
    if (!h || (n && !records)) return IPX_EINVAL;
    if (!n) return IPX_OK;
    (void)hipSetDevice(h->params.device);
    { const int prc = push_complete(h); if (prc) return prc; }
    int rc;
    if ((rc = grow_records(h, h->n_records + n))) return rc;
    ProfScope prof_scope(&h->prof);
    ipx_prof_mark(h->stream, IPXK_BEGIN);
    // the counters go on from where the adoption before left them: key masks accumulate, the merged-reads count too
    unsigned long long *merged_dev = reinterpret_cast<unsigned long long *>(h->counters.as<char>() + IPX_CNT_MERGED);
    HIPCHK(ipx_launch_adopt(h->stream, (const uint4 *)records, n, h->n_records, h->items_a.as<uint4>(), h->pays.as<uint4>(),
                            h->cnts.as<uint32_t>(), h->dig_a.as<uint8_t>(), h->first_window, h->counters.as<Counters>()));
    if ((rc = read_back(h, h->pin_small, h->counters.p, sizeof(Counters) / 4, merged_dev, 2))) return rc;
    h->prof.collect();
    const Counters *c = (const Counters *)h->pin_small;
    const uint32_t *m2 = (const uint32_t *)h->pin_small + sizeof(Counters) / 4;
    for (int k = 0; k < 3; ++k) { h->key_or[k] = c->key_or[k]; h->key_and[k] = c->key_and[k]; }
    h->n_records += n;
    h->n_raw = h->n_records + ((uint64_t)m2[0] | ((uint64_t)m2[1] << 32));
    return IPX_OK;
}

int ipx_decode_bgzf(ipx_handle *h, const ipx_bgzf_file *f, ipx_batch *out) {
    if (!h || !f || !out) return IPX_EINVAL;
    memset(out, 0, sizeof *out);
    if (!f->bytes || !f->n_blocks || !f->payload_begin || !f->payload_end || !f->isize || !f->crc32 || (f->n_tid_map && !f->tid_map))
        return fail(h, IPX_EINVAL, "ipx_decode_bgzf: null array%s%ld", "", 0);
    (void)hipSetDevice(h->params.device);
    { const int prc = push_complete(h); if (prc) return prc; }
    const uint64_t nb = f->n_blocks;
    if (nb >= (1ull << 31)) return fail(h, IPX_EINVAL, "ipx_decode_bgzf: too many blocks%s%ld", "", 0);
    std::vector<uint64_t> out_off(nb + 1);
    uint64_t T = 0;
    for (uint64_t b = 0; b < nb; ++b) {
        if (f->payload_begin[b] > f->payload_end[b] || f->payload_end[b] > f->n_bytes || f->isize[b] > 65536 ||
            (b && f->payload_begin[b] < f->payload_end[b - 1]))
            return fail(h, IPX_EINVAL, "ipx_decode_bgzf: block table out of order or out of the file (block %s%ld)", "", (long)b);
        out_off[b] = T;
        T += f->isize[b];
    }
    out_off[nb] = T;
    if (f->records_begin > T) return fail(h, IPX_EINVAL, "ipx_decode_bgzf: records_begin beyond the inflated stream%s%ld", "", 0);
    auto al = [](uint64_t x) { return (x + 255) & ~255ull; };
    int rc;
    // tables: payload_begin, payload_end, out_off (u64), isize, crc, status, count, prefix (u32), span_begin, span_first (u64), flags
    const uint64_t t_pb = 0, t_pe = al(t_pb + 8 * nb), t_oo = al(t_pe + 8 * nb), t_is = al(t_oo + 8 * (nb + 1)), t_crc = al(t_is + 4 * nb),
                   t_st = al(t_crc + 4 * nb), t_cnt = al(t_st + 4 * nb), t_pre = al(t_cnt + 4 * (nb + 1)), t_sb = al(t_pre + 4 * (nb + 1)),
                   t_sf = al(t_sb + 8 * (nb + 1)), t_sums = al(t_sf + 8 * (nb + 1)), t_flags = al(t_sums + 4 * (ipx_div_up(nb, 4096) + 8)),
                   t_map = al(t_flags + 64), t_ord = al(t_map + 4ull * f->n_tid_map), t_total = al(t_ord + 4 * nb);
    // Does it fit?  The file's bytes, the inflated stream, the decoded batch (20 bytes per alignment and the CIGAR words:
    // at most half of the stream again) and the tables all lie on the device at once.  Asked BEFORE anything is copied: a
    // caller that hears IPX_ENOMEM reads the file through its ordinary batches.
    {
        size_t free_b = 0, total_b = 0;
        const uint64_t have = h->bz_comp.cap + h->bz_out.cap + h->bz_soa.cap + h->bz_cigar.cap;      // (kept from an earlier file)
        const uint64_t need = f->n_bytes + T + T / 2 + t_total + ipx_bgzf_tables_bytes(nb) + (64ull << 20);
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && need > have && need - have > (uint64_t)((double)free_b * 0.9))
            return fail(h, IPX_ENOMEM, "ipx_decode_bgzf: the file needs about %s%ld MB of device memory", "", (long)(need >> 20));
    }
    // IPX_DEBUG_PLAN: where the call's time goes (every lap waits for the stream: the sums are longer than the call)
    const bool laps = getenv("IPX_DEBUG_PLAN") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!laps) return;
        (void)hipStreamSynchronize(h->stream);
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "ipx_decode_bgzf: %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    if ((rc = ensure(h, h->bz_comp, f->n_bytes + 16))) return rc;
    if ((rc = ensure(h, h->bz_out, T + 64))) return rc;
    if ((rc = ensure(h, h->bz_tab, t_total + 256))) return rc;
    if ((rc = ensure(h, h->bz_ws, ipx_bgzf_tables_bytes(nb)))) return rc;
    char *tb = h->bz_tab.as<char>();
    hipStream_t s = h->stream;
    // The file's bytes are the one large copy of this call.  Pageable memory goes through the runtime's staging
    // buffers at a fraction of the link's rate: pin the caller's pages for the duration
    // (On every way out -- the early error returns included -- the stream is waited for before the caller's buffers are
    // let go: copies of the file's bytes and of the caller's block tables may still be under way, and the caller frees both
    // right after a failure: advisor, round 3.)
    // (`order` is the source of an asynchronous copy from pageable memory further down: it is declared BEFORE `reg`, whose
    // destructor waits for the stream, so that on every way out it outlives that wait -- advisor, round 4)
    std::vector<uint32_t> order;
    struct Registered {
        hipStream_t s;
        void *p = nullptr;
        ~Registered() {
            (void)hipStreamSynchronize(s);
            if (p) (void)hipHostUnregister(p);
        }
    } reg{s};
    {
        if (f->n_bytes >= (1u << 20) && hipHostRegister((void *)f->bytes, f->n_bytes, hipHostRegisterDefault) == hipSuccess)
            reg.p = (void *)f->bytes;
        else
            (void)hipGetLastError();
    }
    lap("device memory, pinning");
    HIPCHK(hipMemcpyAsync(h->bz_comp.p, f->bytes, f->n_bytes, hipMemcpyHostToDevice, s));
    lap("the file's bytes to the device");
    HIPCHK(hipMemsetAsync(h->bz_comp.as<char>() + f->n_bytes, 0, 16, s));
    HIPCHK(hipMemsetAsync(h->bz_out.as<char>() + T, 0, 64, s));
    HIPCHK(hipMemcpyAsync(tb + t_pb, f->payload_begin, 8 * nb, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(tb + t_pe, f->payload_end, 8 * nb, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(tb + t_oo, out_off.data(), 8 * (nb + 1), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(tb + t_is, f->isize, 4 * nb, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(tb + t_crc, f->crc32, 4 * nb, hipMemcpyHostToDevice, s));
    if (f->n_tid_map) HIPCHK(hipMemcpyAsync(tb + t_map, f->tid_map, 4ull * f->n_tid_map, hipMemcpyHostToDevice, s));
    // the blocks longest first (by their compressed size): the lanes of a wave then take blocks of about one length and
    // the waves that start last are the short ones
    order.resize(nb);
    for (uint64_t b = 0; b < nb; ++b) order[b] = (uint32_t)b;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
        return f->payload_end[x] - f->payload_begin[x] > f->payload_end[y] - f->payload_begin[y];
    });
    HIPCHK(hipMemcpyAsync(tb + t_ord, order.data(), 4 * nb, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemsetAsync(tb + t_flags, 0, 64, s));
    uint32_t *status = (uint32_t *)(tb + t_st), *flags = (uint32_t *)(tb + t_flags);
    HIPCHK(ipx_launch_bgzf_inflate(s, h->bz_comp.as<uint8_t>(), (const uint64_t *)(tb + t_pb), (const uint64_t *)(tb + t_pe),
                                   (const uint32_t *)(tb + t_is), (const uint64_t *)(tb + t_oo), (const uint32_t *)(tb + t_crc), nb,
                                   h->bz_out.as<uint8_t>(), h->bz_ws.as<uint32_t>(), status, (const uint32_t *)(tb + t_ord)));
    lap("tables, inflate, CRC-32");
    // flags[0] = unaligned, flags[1] = first bad block + 1 (inflate status or CRC), flags[2] = records in all
    HIPCHK(ipx_launch_bam_count_blocks(s, h->bz_out.as<uint8_t>(), (const uint64_t *)(tb + t_oo), (const uint32_t *)(tb + t_is), nb,
                                       f->records_begin, (uint32_t *)(tb + t_cnt), (uint32_t *)(tb + t_pre), (uint32_t *)(tb + t_sums),
                                       flags + 2, (uint64_t *)(tb + t_sb), (uint64_t *)(tb + t_sf), flags));
    hipLaunchKernelGGL(k_first_bad_block, dim3((uint32_t)ipx_div_up(nb, 256)), dim3(256), 0, s, status, (uint32_t)nb, flags + 1);
    HIPCHK(hipGetLastError());
    if ((rc = read_back(h, h->pin_small, flags, 4))) return rc;
    lap("records per block");
    const uint32_t *fl = (const uint32_t *)h->pin_small;
    if (fl[1]) {
        uint32_t st = 0;
        HIPCHK(hipMemcpyAsync(&st, status + (fl[1] - 1), 4, hipMemcpyDeviceToHost, s));
        HIPCHK(ipx_stream_wait(s));
        return fail(h, IPX_EINVAL, (st & 0x100u) ? "ipx_decode_bgzf: CRC-32 mismatch%s in block %ld" : "ipx_decode_bgzf: a block does not inflate to its ISIZE%s (block %ld)",
                    "", (long)fl[1] - 1);
    }
    if (fl[0]) return fail(h, IPX_EINVAL, "ipx_decode_bgzf: records straddle block ends%s%ld", "", 0);
    const uint64_t n = fl[2];
    // the batch: seven arrays of n entries, then the CIGAR words
    const uint64_t o_tid = 0, o_pos = al(o_tid + n * 4), o_flag = al(o_pos + n * 4), o_xs = al(o_flag + n * 2), o_nh = al(o_xs + n),
                   o_off = al(o_nh + n), o_nc = al(o_off + n * 4), soa_total = al(o_nc + n * 4);
    if ((rc = ensure(h, h->bz_soa, soa_total + 256))) return rc;
    if ((rc = ensure(h, h->bam_raw, ipx_bam_workspace_bytes(n) + 256))) return rc;
    char *sa = h->bz_soa.as<char>();
    uint32_t *bam_err = flags + 3, *total_dev = nullptr;
    if (n) {
        HIPCHK(ipx_launch_bam_decode_fields(s, h->bz_out.as<uint8_t>(), (const uint64_t *)(tb + t_sb), (const uint64_t *)(tb + t_sf),
                                            (uint32_t)nb, n, f->n_tid_map ? (const int32_t *)(tb + t_map) : nullptr, f->n_tid_map,
                                            (int32_t *)(sa + o_tid), (int32_t *)(sa + o_pos), (uint16_t *)(sa + o_flag), (uint8_t *)(sa + o_xs),
                                            (uint8_t *)(sa + o_nh), (uint32_t *)(sa + o_off), (uint32_t *)(sa + o_nc), h->bam_raw.p, bam_err,
                                            &total_dev));
        if ((rc = read_back(h, h->pin_small, bam_err, 1, total_dev, 1))) return rc;
        lap("record fields");
        const uint32_t *r2 = (const uint32_t *)h->pin_small;
        if (r2[0] != 0xffffffffu)
            return fail(h, IPX_EINVAL, "ipx_decode_bgzf: malformed BAM record%s (alignment %ld)", "", (long)r2[0] - 1);
        const uint64_t nw = r2[1];
        if ((rc = ensure(h, h->bz_cigar, (nw + 4) * 4 + 256))) return rc;
        HIPCHK(ipx_launch_bam_decode_cigar(s, h->bz_out.as<uint8_t>(), n, nw, (const uint32_t *)(sa + o_off), (const uint32_t *)(sa + o_nc),
                                           h->bz_cigar.as<uint32_t>(), h->bam_raw.p, bam_err));
        if ((rc = read_back(h, h->pin_small, bam_err, 1))) return rc;
        lap("CIGAR words");
        if (((const uint32_t *)h->pin_small)[0] != 0xffffffffu)
            return fail(h, IPX_EINVAL, "ipx_decode_bgzf: CIGAR lengths and records disagree%s%ld", "", 0);
        out->n_cigar_words = nw;
    }
    out->n_alignments = n;
    out->base_read_index = 0;
    out->tid = (const int32_t *)(sa + o_tid); out->pos = (const int32_t *)(sa + o_pos); out->flag = (const uint16_t *)(sa + o_flag);
    out->xs = (const uint8_t *)(sa + o_xs); out->nh = (const uint8_t *)(sa + o_nh); out->cigar_off = (const uint32_t *)(sa + o_off);
    out->n_cigar = (const uint32_t *)(sa + o_nc); out->cigar = h->bz_cigar.as<uint32_t>();
    out->memory = IPX_MEM_DEVICE;
    return IPX_OK;
}

int ipx_set_option(ipx_handle *h, int32_t option, int64_t value) {
    if (!h) return IPX_EINVAL;
    switch (option) {
        case IPX_OPT_COMBINE:
            if (h->n_records) return fail(h, IPX_EINVAL, "ipx_set_option(IPX_OPT_COMBINE) with records pending%s%ld", "", 0);
            h->opt_combine = value != 0;
            return IPX_OK;
        default:
            return fail(h, IPX_EINVAL, "ipx_set_option: unknown option %s%ld", "", (long)option);
    }
}

int ipx_copy(ipx_handle *h, void *dst, const void *src, uint64_t bytes) {
    if (!h || (bytes && (!dst || !src))) return IPX_EINVAL;
    (void)hipSetDevice(h->params.device);
    if (!bytes) return IPX_OK;
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, h->stream));
    HIPCHK(ipx_stream_wait(h->stream));
    return IPX_OK;
}

}  // extern "C"
