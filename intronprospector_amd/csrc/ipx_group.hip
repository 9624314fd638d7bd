// ipx_group.hip -- several GPUs of one node behind one object: include/ipx.h "ipx_group".
//
// No reference counterpart (the reference is one thread on one CPU); SURVEY.md 8e.  The group is built on
// the single-device C ABI only (one ipx_handle per member) plus RCCL for the one exchange step:
//
//   push      the batch is cut into contiguous read-index ranges, member i scans range i on its GPU
//             (one host thread per member, so the members' scans overlap)
//   finalize  every member samples its record keys; common key-range splitters; ipx_records_partition
//             on every member; ONE all-to-all of the 32-byte records: ncclGroupStart .. ncclSend/ncclRecv
//             per pair .. ncclGroupEnd over RCCL/xGMI (peer copies when members share a GPU);
//             ipx_records_adopt + ipx_finalize per member; tables concatenated on the host in member order
//             (= key order), ijunc = rank of the creation order over all of them
//   errors    the data error with the smallest ipx_error.order of all members is the group's error
//             (junctions_extractor.cc:73-74,312-313: the reference throws at the first one)
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ipx.h"

namespace {

// librccl at run time: a process that only ever uses one GPU does not need it
struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool load() {
        if (lib) return true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
        }
        if (!lib) return false;
#define IPX_SYM(field, sym) field = reinterpret_cast<decltype(field)>(dlsym(lib, sym))
        IPX_SYM(CommInitAll, "ncclCommInitAll"); IPX_SYM(CommDestroy, "ncclCommDestroy");
        IPX_SYM(GroupStart, "ncclGroupStart"); IPX_SYM(GroupEnd, "ncclGroupEnd");
        IPX_SYM(Send, "ncclSend"); IPX_SYM(Recv, "ncclRecv"); IPX_SYM(GetErrorString, "ncclGetErrorString");
#undef IPX_SYM
        return CommInitAll && CommDestroy && GroupStart && GroupEnd && Send && Recv && GetErrorString;
    }
};

struct Member {
    int device = 0;
    ipx_handle *h = nullptr;
    hipStream_t stream = nullptr;
    ncclComm_t comm = nullptr;
    void *send = nullptr, *recv = nullptr;
    size_t send_cap = 0, recv_cap = 0;
    uint64_t n_alignments = 0;
    std::vector<uint32_t> off_scratch;        // rebased cigar offsets of the member's range
};

constexpr uint32_t N_SAMPLES = 512, REC_BYTES = 32;

}  // namespace

struct ipx_group {
    std::vector<Member> m;
    bool use_rccl = false;
    Rccl rccl;
    ipx_error err{};
    // the merged table (host)
    std::vector<uint32_t> target, intron_start, intron_end, anchor_start, anchor_end, ijunc, hist_count;
    std::vector<uint8_t> strand;
    std::vector<uint16_t> motif, hist_value;
    std::vector<uint64_t> read_counts, creator_order, hist_begin;
    std::vector<uint64_t> splitters;
    bool have_splitters = false;
};

namespace {

int gfail(ipx_group *g, int status, const char *what, const char *detail = "") {
    memset(&g->err, 0, sizeof g->err);
    g->err.status = status;
    g->err.order = ~0ull;
    snprintf(g->err.detail, sizeof g->err.detail, "%s%s%s", what, detail[0] ? ": " : "", detail);
    return status;
}

// the member's own error becomes the group's
int mfail(ipx_group *g, Member &mb, int status) {
    ipx_last_error(mb.h, &g->err);
    if (!g->err.status) g->err.status = status;
    return status;
}

template <typename F> void for_members(ipx_group *g, std::vector<int> &rc, F fn) {
    const size_t n = g->m.size();
    rc.assign(n, IPX_OK);
    if (n == 1) { rc[0] = fn(0); return; }
    std::vector<std::thread> th;
    for (size_t i = 0; i < n; ++i) th.emplace_back([&, i] { rc[i] = fn(i); });
    for (auto &t : th) t.join();
}

int ensure_dev(Member &mb, void **p, size_t *cap, size_t bytes) {
    if (bytes <= *cap) return IPX_OK;
    if (hipSetDevice(mb.device) != hipSuccess) return IPX_EDEVICE;
    if (*p) (void)hipFree(*p);
    *p = nullptr; *cap = 0;
    const size_t want = bytes + bytes / 4 + 4096;
    if (hipMalloc(p, want) != hipSuccess) return IPX_ENOMEM;
    *cap = want;
    return IPX_OK;
}

}  // namespace

extern "C" {

int ipx_group_create(const ipx_params *params, uint32_t n_devices, const int32_t *devices, ipx_group **out) {
    if (!params || !out || n_devices == 0 || n_devices > 64) return IPX_EINVAL;
    *out = nullptr;
    ipx_group *g = new (std::nothrow) ipx_group();
    if (!g) return IPX_ENOMEM;
    g->m.resize(n_devices);
    bool distinct = true;
    for (uint32_t i = 0; i < n_devices; ++i) {
        g->m[i].device = devices ? devices[i] : (int)i;
        for (uint32_t k = 0; k < i; ++k) distinct = distinct && g->m[k].device != g->m[i].device;
    }
    int rc = IPX_OK;
    for (uint32_t i = 0; i < n_devices && rc == IPX_OK; ++i) {
        ipx_params p = *params;
        p.device = g->m[i].device;
        rc = ipx_create(&p, &g->m[i].h);
        if (rc == IPX_OK && (hipSetDevice(p.device) != hipSuccess ||
                             hipStreamCreateWithFlags(&g->m[i].stream, hipStreamNonBlocking) != hipSuccess))
            rc = IPX_EDEVICE;
    }
    // RCCL needs one communicator per DISTINCT device; members that share a GPU exchange by peer copies
    if (rc == IPX_OK && distinct) {
        if (!g->rccl.load()) rc = IPX_EDEVICE;
        if (rc == IPX_OK) {
            std::vector<int> devs(n_devices);
            std::vector<ncclComm_t> comms(n_devices, nullptr);
            for (uint32_t i = 0; i < n_devices; ++i) devs[i] = g->m[i].device;
            if (g->rccl.CommInitAll(comms.data(), (int)n_devices, devs.data()) != ncclSuccess) rc = IPX_EDEVICE;
            else {
                for (uint32_t i = 0; i < n_devices; ++i) g->m[i].comm = comms[i];
                g->use_rccl = true;
            }
        }
    } else if (rc == IPX_OK) {
        for (uint32_t i = 0; i < n_devices; ++i)
            for (uint32_t k = 0; k < n_devices; ++k) {
                if (g->m[i].device == g->m[k].device) continue;
                // members on different devices without RCCL exchange by peer copies: the devices must see each other
                // ("already enabled" is fine; anything else is not -- round 3 threw the answer away, and a pair of GPUs
                // without peer access would have failed at the first exchange, or copied through the host)
                int can = 0;
                if (hipSetDevice(g->m[i].device) != hipSuccess ||
                    hipDeviceCanAccessPeer(&can, g->m[i].device, g->m[k].device) != hipSuccess || !can) {
                    rc = IPX_EDEVICE;
                    continue;
                }
                const hipError_t pe = hipDeviceEnablePeerAccess(g->m[k].device, 0);
                if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) rc = IPX_EDEVICE;
                (void)hipGetLastError();
            }
    }
    if (rc != IPX_OK) { ipx_group_destroy(g); return rc; }
    *out = g;
    return IPX_OK;
}

int ipx_group_destroy(ipx_group *g) {
    if (!g) return IPX_OK;
    for (Member &mb : g->m) {
        (void)hipSetDevice(mb.device);
        if (mb.stream) (void)hipStreamSynchronize(mb.stream);
        if (mb.comm && g->rccl.CommDestroy) (void)g->rccl.CommDestroy(mb.comm);
        if (mb.send) (void)hipFree(mb.send);
        if (mb.recv) (void)hipFree(mb.recv);
        if (mb.stream) (void)hipStreamDestroy(mb.stream);
        if (mb.h) ipx_destroy(mb.h);
    }
    delete g;
    return IPX_OK;
}

int ipx_group_uses_rccl(const ipx_group *g) { return g && g->use_rccl ? 1 : 0; }

int ipx_group_set_targets(ipx_group *g, uint32_t n, const char *const *names, const int64_t *lengths) {
    if (!g) return IPX_EINVAL;
    for (Member &mb : g->m) {
        const int rc = ipx_set_targets(mb.h, n, names, lengths);
        if (rc) return mfail(g, mb, rc);
    }
    return IPX_OK;
}

int ipx_group_set_genome_2bit(ipx_group *g, uint32_t n_seqs, const char *const *names, const int64_t *lengths,
                              const uint8_t *const *twobit, const uint8_t *const *mask) {
    if (!g) return IPX_EINVAL;
    for (Member &mb : g->m) {          // replicated: the motif of a junction is looked up where its records meet
        const int rc = ipx_set_genome_2bit(mb.h, n_seqs, names, lengths, twobit, mask);
        if (rc) return mfail(g, mb, rc);
    }
    return IPX_OK;
}

int ipx_group_set_option(ipx_group *g, int32_t option, int64_t value) {
    if (!g) return IPX_EINVAL;
    for (Member &mb : g->m) {
        const int rc = ipx_set_option(mb.h, option, value);
        if (rc) return mfail(g, mb, rc);
    }
    return IPX_OK;
}

int ipx_group_push_batch(ipx_group *g, const ipx_batch *b) {
    if (!g || !b) return IPX_EINVAL;
    if (b->memory != IPX_MEM_HOST) return gfail(g, IPX_EINVAL, "ipx_group_push_batch takes batches in host memory");
    const uint64_t n = b->n_alignments, parts = g->m.size();
    if (n == 0) return IPX_OK;
    // CIGAR words in alignment order let every member take just its slice of the array
    bool monotonic = true;
    for (uint64_t k = 1; k <= parts && monotonic; ++k) {
        const uint64_t i = n * k / parts;
        if (i > 0 && i < n) monotonic = b->cigar_off[i] >= b->cigar_off[i - 1];
    }
    std::vector<int> rc;
    for_members(g, rc, [&](size_t i) -> int {
        Member &mb = g->m[i];
        const uint64_t lo = n * i / parts, hi = n * (i + 1) / parts;
        if (hi == lo) return IPX_OK;
        ipx_batch s = *b;
        s.n_alignments = hi - lo;
        s.base_read_index = b->base_read_index + lo;
        s.tid = b->tid + lo; s.pos = b->pos + lo; s.flag = b->flag + lo; s.xs = b->xs + lo; s.nh = b->nh + lo;
        s.n_cigar = b->n_cigar + lo; s.cigar_off = b->cigar_off + lo;
        if (monotonic) {
            const uint64_t w0 = b->cigar_off[lo];
            uint64_t w1 = hi < n ? b->cigar_off[hi] : b->n_cigar_words;
            bool ok = w1 >= w0 && w1 <= b->n_cigar_words;
            mb.off_scratch.resize(hi - lo);
            for (uint64_t k = lo; k < hi && ok; ++k) {
                const uint64_t o = b->cigar_off[k];
                ok = o >= w0 && o + b->n_cigar[k] <= w1;
                mb.off_scratch[k - lo] = (uint32_t)(o - w0);
            }
            if (ok) {
                s.cigar = b->cigar + w0;
                s.n_cigar_words = w1 - w0;
                s.cigar_off = mb.off_scratch.data();
            }
        }
        const int r = ipx_push_batch(mb.h, &s);
        if (r == IPX_OK) mb.n_alignments += hi - lo;
        return r;
    });
    for (size_t i = 0; i < parts; ++i)
        if (rc[i]) return mfail(g, g->m[i], rc[i]);
    return IPX_OK;
}

// One member's share, sharded by the caller: the batch may lie in that member's DEVICE memory (an input that is
// already resident, as in bench.py's steps) or in host memory.  base_read_index places it in the run's input order.
int ipx_group_push_member(ipx_group *g, uint32_t member, const ipx_batch *b) {
    if (!g || !b) return IPX_EINVAL;
    if (member >= g->m.size()) return gfail(g, IPX_EINVAL, "ipx_group_push_member: no such member");
    if (b->n_alignments == 0) return IPX_OK;
    Member &mb = g->m[member];
    const int r = ipx_push_batch(mb.h, b);
    if (r) return mfail(g, mb, r);
    mb.n_alignments += b->n_alignments;
    return IPX_OK;
}

int ipx_group_clear(ipx_group *g) {
    if (!g) return IPX_EINVAL;
    for (Member &mb : g->m) {
        mb.n_alignments = 0;
        const int rc = ipx_clear(mb.h);
        if (rc) return mfail(g, mb, rc);
    }
    return IPX_OK;
}

int ipx_group_last_error(const ipx_group *g, ipx_error *out) {
    if (!g || !out) return IPX_EINVAL;
    *out = g->err;
    return IPX_OK;
}

int ipx_group_finalize(ipx_group *g, ipx_table *out) {
    if (!g || !out) return IPX_EINVAL;
    memset(out, 0, sizeof *out);
    memset(&g->err, 0, sizeof g->err);
    const size_t P = g->m.size();
    std::vector<int> rc;
    std::vector<ipx_table> tabs(P);

    // IPX_GROUP_FORCE_EXCHANGE=1: a group of ONE runs the exchange too (the RCCL calls on a one-GPU box)
    static const bool force = [] { const char *e = getenv("IPX_GROUP_FORCE_EXCHANGE"); return e && e[0] == '1'; }();
    if (P > 1 || force) {
        // 1. common splitters from everybody's samples; kept while the exchange stays balanced
        std::vector<uint64_t> pending(P, 0);
        for (size_t i = 0; i < P; ++i) (void)ipx_records_pending(g->m[i].h, &pending[i]);
        if (!g->have_splitters) {
            std::vector<uint64_t> samples(P * N_SAMPLES);
            for_members(g, rc, [&](size_t i) { return ipx_records_sample(g->m[i].h, N_SAMPLES, samples.data() + i * N_SAMPLES); });
            for (size_t i = 0; i < P; ++i)
                if (rc[i]) return mfail(g, g->m[i], rc[i]);
            std::vector<uint64_t> all;
            for (size_t i = 0; i < P; ++i)
                if (pending[i]) all.insert(all.end(), samples.begin() + i * N_SAMPLES, samples.begin() + (i + 1) * N_SAMPLES);
            std::sort(all.begin(), all.end());
            g->splitters.assign(P - 1, ~0ull >> 1);
            for (size_t k = 1; k < P && !all.empty(); ++k) g->splitters[k - 1] = all[k * all.size() / P];
            g->have_splitters = !all.empty();
        }
        // 2. every member buckets its records by key range
        std::vector<uint64_t> counts(P * P, 0);
        for_members(g, rc, [&](size_t i) -> int {
            Member &mb = g->m[i];
            int r = ensure_dev(mb, &mb.send, &mb.send_cap, (size_t)pending[i] * REC_BYTES + 256);
            if (r) return r;
            return ipx_records_partition(mb.h, (uint32_t)P, g->splitters.data(), counts.data() + i * P, mb.send);
        });
        for (size_t i = 0; i < P; ++i)
            if (rc[i]) return rc[i] == IPX_ENOMEM ? gfail(g, rc[i], "send buffer") : mfail(g, g->m[i], rc[i]);
        // 3. the all-to-all
        std::vector<uint64_t> n_recv(P, 0);
        for (size_t i = 0; i < P; ++i)
            for (size_t p = 0; p < P; ++p) n_recv[p] += counts[i * P + p];
        for (size_t i = 0; i < P; ++i) {
            const int r = ensure_dev(g->m[i], &g->m[i].recv, &g->m[i].recv_cap, (size_t)n_recv[i] * REC_BYTES + 256);
            if (r) return gfail(g, r, "receive buffer");
        }
        if (g->use_rccl) {
            ncclResult_t nr = g->rccl.GroupStart();
            for (size_t i = 0; i < P && nr == ncclSuccess; ++i) {
                Member &mb = g->m[i];
                (void)hipSetDevice(mb.device);
                uint64_t soff = 0, roff = 0;
                for (size_t p = 0; p < P && nr == ncclSuccess; ++p) {
                    const uint64_t ns = counts[i * P + p], nv = counts[p * P + i];
                    if (ns) nr = g->rccl.Send((const char *)mb.send + soff * REC_BYTES, ns * REC_BYTES, ncclUint8, (int)p, mb.comm, mb.stream);
                    if (nv && nr == ncclSuccess)
                        nr = g->rccl.Recv((char *)mb.recv + roff * REC_BYTES, nv * REC_BYTES, ncclUint8, (int)p, mb.comm, mb.stream);
                    soff += ns;
                    roff += nv;
                }
            }
            const ncclResult_t ne = g->rccl.GroupEnd();
            if (nr == ncclSuccess) nr = ne;
            if (nr != ncclSuccess) return gfail(g, IPX_EDEVICE, "RCCL all-to-all", g->rccl.GetErrorString(nr));
        } else {
            for (size_t i = 0; i < P; ++i) {
                Member &mb = g->m[i];
                (void)hipSetDevice(mb.device);
                uint64_t soff = 0;
                for (size_t p = 0; p < P; ++p) {
                    const uint64_t ns = counts[i * P + p];
                    uint64_t roff = 0;                       // part i of member p's receive buffer
                    for (size_t q = 0; q < i; ++q) roff += counts[q * P + p];
                    if (ns && hipMemcpyPeerAsync((char *)g->m[p].recv + roff * REC_BYTES, g->m[p].device,
                                                 (const char *)mb.send + soff * REC_BYTES, mb.device, ns * REC_BYTES,
                                                 mb.stream) != hipSuccess)
                        return gfail(g, IPX_EDEVICE, "peer copy of the record exchange");
                    soff += ns;
                }
            }
        }
        for (Member &mb : g->m) {
            (void)hipSetDevice(mb.device);
            if (hipStreamSynchronize(mb.stream) != hipSuccess) return gfail(g, IPX_EDEVICE, "record exchange", hipGetErrorString(hipGetLastError()));
        }
        // a skewed exchange draws new splitters next time (they decide the balance only, never the result)
        const uint64_t total = std::accumulate(n_recv.begin(), n_recv.end(), (uint64_t)0);
        const uint64_t most = *std::max_element(n_recv.begin(), n_recv.end());
        if (total && (double)most > 1.25 * (double)total / (double)P) g->have_splitters = false;
        // 4. every member owns a key range now
        for_members(g, rc, [&](size_t i) -> int {
            Member &mb = g->m[i];
            const int r = ipx_records_adopt(mb.h, n_recv[i], mb.recv, mb.n_alignments);
            if (r) return r;
            return ipx_finalize(mb.h, IPX_MEM_HOST, &tabs[i]);
        });
    } else {
        rc.assign(1, ipx_finalize(g->m[0].h, IPX_MEM_HOST, &tabs[0]));
    }
    // errors: a failure wins; else the earliest data error of the run
    int worst = IPX_OK;
    size_t who = 0;
    uint64_t best_order = ~0ull;
    for (size_t i = 0; i < P; ++i) {
        if (rc[i] == IPX_OK) continue;
        ipx_error e;
        ipx_last_error(g->m[i].h, &e);
        const uint64_t order = rc[i] == IPX_EDATA ? e.order : 0;
        if (worst == IPX_OK || order < best_order) { worst = rc[i]; who = i; best_order = order; }
    }
    if (worst != IPX_OK && worst != IPX_EDATA) return mfail(g, g->m[who], worst);

    // 5. one table: member order is key order
    uint64_t J = 0, B = 0, n_aln = 0, n_rec = 0;
    for (const ipx_table &t : tabs) { J += t.n_junctions; B += t.n_hist; n_aln += t.n_alignments; n_rec += t.n_records; }
    g->target.resize(J); g->intron_start.resize(J); g->intron_end.resize(J); g->anchor_start.resize(J);
    g->anchor_end.resize(J); g->ijunc.resize(J); g->strand.resize(J); g->motif.resize(J);
    g->read_counts.resize(3 * J); g->creator_order.resize(J); g->hist_begin.resize(J + 1);
    g->hist_value.resize(B); g->hist_count.resize(B);
    // (every member's rows are copied by a thread of its own: at millions of junctions the copies are the step)
    std::vector<uint64_t> j_at(P + 1, 0), b_at(P + 1, 0);
    for (size_t i = 0; i < P; ++i) { j_at[i + 1] = j_at[i] + tabs[i].n_junctions; b_at[i + 1] = b_at[i] + tabs[i].n_hist; }
    {
        std::vector<int> crc;
        for_members(g, crc, [&](size_t i) -> int {
            const ipx_table &t = tabs[i];
            const uint64_t n = t.n_junctions, j0 = j_at[i], b0 = b_at[i];
            if (n) {
                memcpy(&g->target[j0], t.target, n * 4); memcpy(&g->intron_start[j0], t.intron_start, n * 4);
                memcpy(&g->intron_end[j0], t.intron_end, n * 4); memcpy(&g->anchor_start[j0], t.anchor_start, n * 4);
                memcpy(&g->anchor_end[j0], t.anchor_end, n * 4); memcpy(&g->strand[j0], t.strand, n);
                memcpy(&g->motif[j0], t.motif, n * 2); memcpy(&g->read_counts[3 * j0], t.read_counts, 3 * n * 8);
                memcpy(&g->creator_order[j0], t.creator_order, n * 8);
                for (uint64_t k = 0; k < n; ++k) g->hist_begin[j0 + k] = t.hist_begin[k] + b0;
            }
            if (t.n_hist) {
                memcpy(&g->hist_value[b0], t.hist_value, t.n_hist * 2);
                memcpy(&g->hist_count[b0], t.hist_count, t.n_hist * 4);
            }
            return IPX_OK;
        });
    }
    g->hist_begin[J] = B;
    // ijunc = rank of the creation order (junctions_extractor.cc:459-460), over the whole run.  Every member's table
    // carries the ranks WITHIN the member (its own ijunc): inverted, they are the member's rows in creation order, and
    // the run's order is a merge of those P sorted lists -- O(J P) in streams (round 4 sorted J indices through their
    // 64-bit orders on one thread: 0.2 s of a 0.22 s step at 2.7 M junctions)
    if (P > 1) {
        // (round 5, second step: the merge itself is cut into T ranges of the order axis, a thread each -- creation
        // orders are distinct, so a pivot order splits every member's list at one place and the ranks of a range start
        // at the sum of those places.  One thread took 35 ms for 2.7 M junctions of two members)
        std::vector<std::vector<uint32_t>> by_order(P);
        std::vector<std::vector<uint64_t>> ord(P);           // member i's creation orders, ascending
        {
            std::vector<int> crc;
            for_members(g, crc, [&](size_t i) -> int {
                const uint64_t n = tabs[i].n_junctions;
                try { by_order[i].resize(n); ord[i].resize(n); } catch (const std::bad_alloc &) { return IPX_ENOMEM; }
                for (uint64_t k = 0; k < n; ++k) by_order[i][tabs[i].ijunc[k]] = (uint32_t)k;
                for (uint64_t r = 0; r < n; ++r) ord[i][r] = g->creator_order[j_at[i] + by_order[i][r]];
                return IPX_OK;
            });
            for (size_t i = 0; i < P; ++i)
                if (crc[i]) return gfail(g, crc[i], "global sjN");
        }
        static const unsigned merge_threads = [] {
            const char *e = getenv("IPX_GROUP_MERGE_THREADS");
            const unsigned hw = std::thread::hardware_concurrency();
            const unsigned t = e ? (unsigned)atoi(e) : (hw ? hw : 4u);
            return std::min(std::max(t, 1u), 32u);
        }();
        const unsigned T = J >= 4096 ? merge_threads : 1u;
        size_t longest = 0;
        for (size_t i = 1; i < P; ++i)
            if (ord[i].size() > ord[longest].size()) longest = i;
        // cut[t][i]: where range t begins in member i's list (pivots: quantiles of the longest list)
        std::vector<std::vector<uint64_t>> cut(T + 1, std::vector<uint64_t>(P, 0));
        for (size_t i = 0; i < P; ++i) cut[T][i] = ord[i].size();
        for (unsigned t = 1; t < T; ++t) {
            const uint64_t pivot = ord[longest][(uint64_t)t * ord[longest].size() / T];
            for (size_t i = 0; i < P; ++i)
                cut[t][i] = (uint64_t)(std::lower_bound(ord[i].begin(), ord[i].end(), pivot) - ord[i].begin());
        }
        auto merge_range = [&](unsigned t) {
            std::vector<uint64_t> head(cut[t]);
            const std::vector<uint64_t> &end = cut[t + 1];
            uint64_t r = std::accumulate(head.begin(), head.end(), (uint64_t)0);
            for (;; ++r) {
                size_t best = P;
                uint64_t best_o = 0;
                for (size_t i = 0; i < P; ++i) {
                    if (head[i] == end[i]) continue;
                    const uint64_t o = ord[i][head[i]];
                    if (best == P || o < best_o) { best = i; best_o = o; }
                }
                if (best == P) break;
                g->ijunc[j_at[best] + by_order[best][head[best]]] = (uint32_t)r;
                ++head[best];
            }
        };
        if (T == 1) {
            merge_range(0);
        } else {
            std::vector<std::thread> th;
            for (unsigned t = 0; t < T; ++t) th.emplace_back(merge_range, t);
            for (auto &x : th) x.join();
        }
    } else if (J) {
        memcpy(g->ijunc.data(), tabs[0].ijunc, J * 4);
    }
    ipx_table t{};
    t.n_junctions = J; t.n_hist = B; t.n_alignments = n_aln; t.n_records = n_rec; t.memory = IPX_MEM_HOST;
    t.target = g->target.data(); t.intron_start = g->intron_start.data(); t.intron_end = g->intron_end.data();
    t.anchor_start = g->anchor_start.data(); t.anchor_end = g->anchor_end.data(); t.strand = g->strand.data();
    t.motif = g->motif.data(); t.read_counts = g->read_counts.data(); t.creator_order = g->creator_order.data();
    t.ijunc = g->ijunc.data(); t.hist_begin = g->hist_begin.data(); t.hist_value = g->hist_value.data();
    t.hist_count = g->hist_count.data();
    *out = t;
    if (worst == IPX_EDATA) {
        ipx_last_error(g->m[who].h, &g->err);
        return IPX_EDATA;
    }
    return IPX_OK;
}

}  // extern "C"
