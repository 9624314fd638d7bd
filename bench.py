#!/usr/bin/env python3
"""Benchmark of the junction-extraction hot path on MI355X.

A step is one pass of the path over one batch of synthetic alignments that is
already resident in HBM: ipx_clear -> ipx_push_batch (CIGAR scan -> junction
records) -> ipx_finalize (radix sort + segmented reduce -> junction table in
HBM).  With N > 1 every rank holds its own contiguous shard of N x reads
alignments (weak scaling) and the partial results are combined over RCCL.

Timing: W warm-up steps, then exactly K timed steps between barrier + synchronize.  Inside the timed
region HIP events bracket the launches of the dominant kernel class only (roofline.kernel_avg_ms); the
full per-kernel table (roofline.kernels_ms_per_step) comes from three further, untimed steps with
events around every launch.

What the JSON line says (rank 0 prints ONE line; DESIGN.md "Measurement"):
  value / ms_per_step   whole-job alignments per second over the timed region (input resident in HBM)
  roofline.frac         SURVEY 8(d): algorithmic bytes of the WHOLE step (20 + 4c + 64j per alignment)
                        / timed step / 8 TB/s -- sort passes, heads and launch gaps all count as time
  roofline.kernel_*     the dominant kernel alone against ITS minimum traffic (flat keys)
  roofline.traffic      HBM bytes per step by PMC (profiles/traffic.json), waste_ratio = traffic / algorithmic
  cpu_baseline          the oracle (a scalar port of the reference) on this host: 1 core (value) and all
                        cores the cgroup grants (all_cores_value), host_cores / cpu_model stated
  configs               N == 1 only: BASELINE.json's other single-GPU configurations in the same run
                        (L: 2 M long reads, G: 10 M reads + device genome), each with ms_per_step,
                        frac and a parity check of a prefix against the oracle
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
KEYS = ("tid", "pos", "flag", "xs", "nh", "cigar_off", "n_cigar", "cigar")
DEFAULT_READS = {"S": 10_000_000, "G": 10_000_000, "L": 2_000_000}     # BASELINE.json configs[1..3]



class LookAhead:
    """A scan of the NEXT batch on a helper thread (the exchange path's two handles take turns).  What the thread raises
    is kept and raised again where it is joined: a scan that failed must not leave a cleared or half-filled handle to the
    next step's exchange (advisor, round 4)."""
    def __init__(self, fn, arg):
        import threading
        self.exc = None
        def body():
            try:
                fn(arg)
            except BaseException as e:      # noqa: BLE001 -- re-raised by join()
                self.exc = e
        self.th = threading.Thread(target=body)
        self.th.start()

    def join(self):
        self.th.join()
        if self.exc is not None:
            exc, self.exc = self.exc, None
            raise exc

def make_workload(kind, reads, seed):
    from intronprospector_amd import synth
    if kind == "S":
        return synth.short_reads(reads, seed=seed)
    if kind == "G":     # S with a synthetic genome: strand / motif inference, --sj-filter=canon
        return synth.short_reads(reads, seed=seed, genome=True)
    if kind == "L":
        return synth.long_reads(reads, seed=seed)
    raise SystemExit("unknown workload %r" % kind)


# minimum bytes each kernel class has to move per launch (DESIGN.md "Kernels"):
# n alignments, w CIGAR words, r junction records, m records after the scan's tile combiner
KERNEL_BYTES = {
    "k_scan": lambda n, w, r, m: 20 * n + 4 * w + 36 * m,    # SoA + CIGAR in, 36-byte records out
    "k_hist": lambda n, w, r, m: 1 * m,                      # one digit byte per record
    "k_scatter": lambda n, w, r, m: 33 * m,                  # item in, item + next digit byte out
    "k_count_heads": lambda n, w, r, m: 16 * m,
    "k_write_heads": lambda n, w, r, m: 16 * m,
    "k_reduce_a": lambda n, w, r, m: 36 * m,                 # item + payload + count word of every record
    "k_reduce": lambda n, w, r, m: 36 * m,
}



# An event pair around a launch adds ~3 us to what it measures (the pair's own records; measured: eight bracketed radix
# passes make a 0.57 ms step 0.60 ms).  The probe step that picks the dominant kernel class brackets EVERY launch, so a
# class of eight 17 us launches reads 24 us longer than rocprofv3 sees it; the choice discounts that -- without it the
# timed region of a step whose scan (one pair) and radix passes (eight pairs) are level brackets the eight and pays for it.
EVENT_PAIR_MS = 0.003


def pick_dominant(warm, fallback):
    cand = {k: v for k, v in warm.items() if k in ("k_scan", "k_scatter", "k_reduce")}
    if not cand:
        return fallback
    return max(cand.items(), key=lambda kv: kv[1][1] - EVENT_PAIR_MS * kv[1][0])[0]

def host_info():
    """cores this process may use (affinity, cgroup quota) and the CPU model"""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(p)
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / p
        except Exception:
            pass
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    usable = max(1, min(cores, int(quota)) if quota else cores)
    return {"host_cores": os.cpu_count(), "affinity_cores": cores, "cgroup_cpu_quota": quota, "usable_cores": usable,
            "cpu_model": model}


def _oracle_params(kind):
    return {"max_anchor_indel_size": 36} if kind == "L" else {}


def _cpu_worker(path, lo, hi, names, params, budget_s, barrier, out_q):
    """all-cores leg: one process per core, each over its own contiguous range of the sample"""
    from intronprospector_amd import synth
    from tests.oracle_binding import Oracle
    batch = {k: np.load(os.path.join(path, k + ".npy"), mmap_mode="r") for k in KEYS}
    shard = synth.slice_batch(batch, lo, hi)
    barrier.wait()
    passes, spent = 0, 0.0
    t_start = time.perf_counter()
    while spent < budget_s and passes < 2000:
        ora = Oracle(targets=names, **params)
        t0 = time.perf_counter()
        ora.process(shard)
        spent += time.perf_counter() - t0
        ora.close()
        passes += 1
    out_q.put(((hi - lo) * passes, time.perf_counter() - t_start))


def cpu_baseline(batch, names, kind, budget_reads, budget_s=10.0, all_cores=True):
    """The oracle (a scalar port of the reference, oracle/) timed on this host: whole passes over the first
    budget_reads alignments of the same batch (a fresh table each pass) until about budget_s seconds of CPU
    work are done -- on ONE core (the reference is single-threaded), then on all cores the cgroup grants,
    one process per core over contiguous read ranges (per-shard tables, not merged: an upper bound for a
    threaded port).  Reported next to the GPU number only."""
    from intronprospector_amd import synth
    from tests.oracle_binding import Oracle
    n = min(budget_reads, len(batch["tid"]))
    sample = synth.slice_batch(batch, 0, n)
    params = _oracle_params(kind)
    passes, spent, nj = 0, 0.0, 0
    while spent < budget_s and passes < 2000:
        ora = Oracle(targets=names, **params)
        t0 = time.perf_counter()
        ora.process(sample)
        nj = ora.lib.ipxo_num_junctions(ora.h)
        spent += time.perf_counter() - t0
        ora.close()
        passes += 1
    info = host_info()
    out = {"value": n * passes / spent, "unit": "alignments/s", "cores": 1, "kind": "port",
           # the survey's probe of the REFERENCE itself (unmodified sources behind a throw-away SAM shim,
           # 2 M synthetic 100 bp reads, one thread, another machine): the port keeps junctions in a hash table where the
           # reference has a std::map with std::string compares -- `value` is the port's, not the reference's
           "reference_probe": {"value": 0.65e6, "unit": "alignments/s", "cores": 1,
                               "source": "BASELINE.md section 2 / SURVEY.md section 6 (indicative, measured once in the survey's container)"},
           "sample": "%d passes over %s %d alignments of the %s workload, oracle/libipxo.so (scalar C restatement), "
                     "%d junctions, %.1f s of CPU work" % (passes, "all" if n == len(batch["tid"]) else "the first",
                                                           n, kind, nj, spent)}
    out.update(info)
    workers = info["usable_cores"]
    if all_cores and workers > 1:
        import multiprocessing as mp
        import shutil
        import tempfile
        tmp = tempfile.mkdtemp(prefix="ipx_cpu_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
        try:
            for k in KEYS:
                np.save(os.path.join(tmp, k + ".npy"), sample[k])
            ctx = mp.get_context("spawn")
            barrier, q = ctx.Barrier(workers), ctx.Queue()
            cuts = np.linspace(0, n, workers + 1).astype(int)
            procs = [ctx.Process(target=_cpu_worker, args=(tmp, int(cuts[i]), int(cuts[i + 1]), list(names), params,
                                                           budget_s, barrier, q)) for i in range(workers)]
            for p in procs:
                p.start()
            got, deadline = [], time.time() + 20 * budget_s + 120
            while len(got) < workers:
                try:
                    got.append(q.get(timeout=1.0))
                except Exception:
                    if any(p.exitcode not in (None, 0) for p in procs) or time.time() > deadline:
                        for p in procs:
                            if p.is_alive():
                                p.terminate()
                        raise RuntimeError("a CPU-baseline worker died or timed out")
            for p in procs:
                p.join(timeout=60)
            out["all_cores"] = workers
            out["all_cores_value"] = sum(g[0] for g in got) / max(g[1] for g in got)
            out["all_cores_sample"] = "%d processes, one contiguous read range each, %.1f s wall, shard tables not merged" % (
                workers, max(g[1] for g in got))
        except Exception as e:                      # the single-core figure stands on its own
            out["all_cores_error"] = repr(e)[:200]
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    return out


def table_digest(tab):
    """order-free digest of a table as numpy dicts (device: extractor.get_junctions; oracle: Oracle.table)"""
    import hashlib
    h = hashlib.sha256()
    order = np.lexsort((tab["intron_end"], tab["intron_start"], tab["target"].astype(np.int64)))
    for k in ("target", "intron_start", "intron_end", "anchor_start", "anchor_end", "strand", "ijunc"):
        h.update(np.ascontiguousarray(tab[k][order].astype(np.int64)).tobytes())
    h.update(np.ascontiguousarray(tab["counts"][order].astype(np.uint64)).tobytes())
    h.update(np.ascontiguousarray(tab["offsets_flat"].astype(np.uint16)).tobytes())
    return h.hexdigest()


def parity_prefix(kind, batch, names, genome_pack, fasta_genome, n_prefix, device):
    """the device path against the oracle on the first n_prefix alignments of the benchmark's own batch:
    SHA-256 over keys, anchors, strand, ijunc, counts and the per-junction read-offset multisets"""
    import tempfile
    import intronprospector_amd as ipa
    from intronprospector_amd import synth, _lib as L
    from intronprospector_amd.extractor import expand_histogram, offsets_per_junction
    from tests.oracle_binding import Oracle
    n = min(n_prefix, len(batch["tid"]))
    sample = synth.slice_batch(batch, 0, n)
    params = _oracle_params(kind)
    dparams, oparams, tmpdir = dict(params), dict(params), None
    if genome_pack is not None:
        dparams.update(genome=genome_pack, junction_filter=L.IPX_SJ_CANON)
        tmpdir = tempfile.mkdtemp(prefix="ipx_fa_")
        fasta = os.path.join(tmpdir, "g.fa")
        fasta_genome.write_fasta(fasta)
        oparams.update(genome_fasta=fasta, junction_filter=2)
    ex = ipa.JunctionsExtractor(targets=names, device=device, **dparams)
    ex.push(sample)
    dev = ex.get_junctions()
    ex.close()
    ora = Oracle(targets=names, **oparams)
    ora.process(sample)
    ref = ora.table()
    ora.close()
    if tmpdir:
        import shutil
        shutil.rmtree(tmpdir, ignore_errors=True)

    def flat_offsets(order, counts, flat):
        starts = np.concatenate([[0], np.cumsum(counts)])[:-1]
        return np.concatenate([flat[starts[i]:starts[i] + counts[i]] for i in order]) if len(order) else flat[:0]
    od = np.lexsort((dev["intron_end"], dev["intron_start"], dev["target"].astype(np.int64)))
    dev["offsets_flat"] = flat_offsets(od, offsets_per_junction(dev).astype(np.int64), expand_histogram(dev))
    oo = np.lexsort((ref["intron_end"], ref["intron_start"], ref["target"].astype(np.int64)))
    ref["offsets_flat"] = flat_offsets(oo, ref["n_offsets"].astype(np.int64), ref["offsets"])
    a, b = table_digest(dev), table_digest(ref)
    return {"alignments": n, "junctions": int(len(ref["intron_start"])), "sha256": a[:16], "equals_oracle": a == b}


def upload(batch, dev):
    import torch
    arrays = [torch.from_numpy(batch[k]).to(dev) for k in KEYS]
    torch.cuda.synchronize()
    return arrays, [t.data_ptr() for t in arrays]


def kernel_source_hash():
    """sha256 over the device sources (csrc/*.hip, *.h, *.inc): what profiles/traffic.json was measured on is stamped with it"""
    import glob
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "intronprospector_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h")) + glob.glob(os.path.join(d, "*.inc"))):
        h.update(os.path.basename(f).encode() + b"\0" + open(f, "rb").read())
    return h.hexdigest()[:16]


def load_traffic(kind):
    """PMC traffic of the workload from profiles/traffic.json -- only if it was measured on THESE kernel sources
    (its `kernel_sources_sha16` stamp); a stale file yields nothing and the line says traffic: null"""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        doc = json.load(open(tpath))
    except Exception:
        return {}
    if doc.get("kernel_sources_sha16") != kernel_source_hash():
        return {"stale": True, "measured_on": doc.get("kernel_sources_sha16")}
    return doc.get(kind, {})


def roofline_block(kind, n, n_words, r, m, ms_per_step, dominant, timed, stats, breakdown_steps, steps):
    c, j = n_words / n, r / n
    b_alg = 20 + 4 * c + 64 * j                      # SURVEY.md 8(d), bytes per alignment
    alg_step = b_alg * n
    dev_ms = sum(v[1] for v in stats.values()) / breakdown_steps if stats else None
    tr = load_traffic(kind)
    traffic_step = tr.get("step_bytes")
    traffic_lower = tr.get("step_bytes_lower")
    out = {
        "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "basis": "whole step: SURVEY 8(d) algorithmic bytes (20 + 4c + 64j per alignment) / timed step",
        "achieved": alg_step / (ms_per_step * 1e-3) / 1e9,
        "frac": alg_step / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "traffic": traffic_step,
        "bytes_per_alignment": b_alg, "algorithmic_bytes_per_step": alg_step,
        "traffic_step_bytes": traffic_step,
        "waste_ratio": (traffic_step / alg_step) if traffic_step else None,
        # FETCH_SIZE is doubled only for the classes whose loads are 16-byte-per-lane streams; the others are a range
        # (profiles/pmc_traffic.py): `traffic` is its upper end, this the lower
        "traffic_lower": traffic_lower,
        "waste_ratio_lower": (traffic_lower / alg_step) if traffic_lower else None,
        "launches_per_step": tr.get("launches_per_step"),
        "traffic_source": ("profiles/traffic.json is stale (measured on kernel sources %s, these are %s): traffic not reported"
                           % (tr.get("measured_on"), kernel_source_hash())) if tr.get("stale") else
                          "profiles/traffic.json, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE on these kernel sources",
        "device_ms_per_step": dev_ms,
        "records_after_combine": m,
    }
    if dominant in timed:
        launches, total_ms = timed[dominant]
        avg_ms = total_ms / launches
        kb = KERNEL_BYTES.get(dominant, lambda *a: 0)(n, n_words, r, m)
        out.update({
            "kernel": dominant, "kernel_launches_per_step": launches / steps, "kernel_avg_ms": avg_ms,
            "kernel_bytes_per_launch": kb, "kernel_achieved": kb / (avg_ms * 1e-3) / 1e9,
            # what the kernel class must move per launch (the same figure, under the name the review asked for) and what the
            # PMC counters saw it move (profiles/traffic.json; upper end of the range for mixed access patterns)
            "kernel_necessary_bytes": kb,
            "kernel_traffic": (tr.get(dominant) / (launches / steps)) if tr.get(dominant) else None,
            "kernel_frac": kb / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            # PMC bytes of this kernel class per STEP (all its launches; profiles/traffic.json)
            "kernel_traffic_step_bytes": tr.get(dominant),
        })
    if stats:
        out["kernels_ms_per_step"] = {k: v[1] / breakdown_steps for k, v in sorted(stats.items(), key=lambda kv: -kv[1][1])}
    return out


def run_single(kind, batch, names, genome, dev, local_rank, steps, warmup, prefix, extra_batches=()):
    """one workload on one GPU, no collectives: the `configs` entries.  extra_batches: further batches of the same
    shape (other seeds); step k then takes batch k mod their number, so that nothing a handle carries from one batch
    to the next (first-pass digit window, second-stage grid, combiner mode, capacity estimates) is a guaranteed hit"""
    import torch
    import intronprospector_amd as ipa
    from intronprospector_amd import _lib as L
    batches = [batch] + list(extra_batches)
    n = int(round(sum(len(b["tid"]) for b in batches) / len(batches)))
    n_words = int(round(sum(len(b["cigar"]) for b in batches) / len(batches)))
    ups = [upload(b, dev) for b in batches]
    arrays = [u[0] for u in ups]
    turn = [0]
    params = _oracle_params(kind)
    pack = None
    if genome is not None:
        pack = (genome.names, [(t, m, len(q)) for (t, m), q in zip(genome.twobit(), genome.seqs)])
        params.update(genome=pack, junction_filter=L.IPX_SJ_CANON)
    ex = ipa.JunctionsExtractor(targets=names, device=local_rank, **params)
    ex.reserve(0, 0, max(len(b["cigar"]) for b in batches) // 2 + 1024)
    table = None

    def step():
        k = turn[0] % len(batches)
        turn[0] += 1
        ex.clear()
        ex.push_device(len(batches[k]["tid"]), len(batches[k]["cigar"]), ups[k][1], base_read_index=0)
        m = ex.records_pending()
        rc, t = ex.finalize_raw(L.IPX_MEM_DEVICE)
        if rc:
            ex._check(rc)
        return t, m
    for _ in range(warmup):
        table, m = step()
    ex.profile(True)
    table, m = step()
    torch.cuda.synchronize()
    warm = ex.profile_read()
    dominant = pick_dominant(warm, "k_scan")
    ex.profile_only(dominant)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        table, m = step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    timed = ex.profile_read()
    ex.profile(True)
    for _ in range(2):
        table, m = step()
    torch.cuda.synchronize()
    stats = ex.profile_read()
    ex.profile(False)
    r, nj = int(table.n_records), int(table.n_junctions)
    ex.close()
    del arrays
    ms = elapsed / steps * 1e3
    roof = roofline_block(kind, n, n_words, r, m, ms, dominant, timed, stats, 2, steps)
    out = {"workload": kind, "alignments": n, "cigar_words": n_words, "records": r, "junctions": nj, "steps": steps,
           "ms_per_step": ms, "value": n * steps / elapsed, "unit": "alignments/s",
           "frac": roof["frac"], "achieved": roof["achieved"], "kernel": roof.get("kernel"),
           "kernel_frac": roof.get("kernel_frac"), "kernel_avg_ms": roof.get("kernel_avg_ms"),
           "records_after_combine": m, "kernels_ms_per_step": roof.get("kernels_ms_per_step")}
    if len(batches) > 1:
        out["rotating"] = "%d batches of other seeds take turns (alignments, CIGAR words: their mean; records: the last step's)" % len(batches)
    if prefix:
        out["parity_prefix"] = parity_prefix(kind, batch, names, pack, genome, prefix, local_rank)
    return out


def run_m1(args, local_rank):
    """One rank's share of config 5 (M = 100 chunks of 10 M alignments, seeds SEED_M0 + i, a contiguous
    eighth per GPU): chunks 0..12 (the thirteenth half) resident in HBM, a step = ipx_clear + thirteen pushes into
    ONE table + ipx_finalize.  The whole table is compared with the oracle's of the same stream."""
    import torch
    import intronprospector_amd as ipa
    from intronprospector_amd import synth, _lib as L
    from tests import helpers
    from tests.oracle_binding import Oracle
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    t_gen = time.perf_counter()
    ups, sizes, names, ora, t_cpu = [], [], None, None, 0.0
    for i in range(13):
        b, names, _ = synth.short_reads(10_000_000 if i < 12 else 5_000_000, seed=synth.SEED_M0 + i)
        if ora is None:
            ora = Oracle(targets=names)
        t0 = time.perf_counter()
        ora.process(b)
        t_cpu += time.perf_counter() - t0
        ups.append(upload(b, dev))
        sizes.append((len(b["tid"]), len(b["cigar"])))
        del b
    t_gen = time.perf_counter() - t_gen
    n, n_words = sum(a for a, _ in sizes), sum(w for _, w in sizes)
    ex = ipa.JunctionsExtractor(targets=names, device=local_rank)

    def step():
        ex.clear()
        base = 0
        for (na, nw), (_, ptrs) in zip(sizes, ups):
            ex.push_device(na, nw, ptrs, base_read_index=base)
            base += na
        m = ex.records_pending()
        rc, t = ex.finalize_raw(L.IPX_MEM_DEVICE)
        if rc:
            ex._check(rc)
        return t, m
    for _ in range(max(args.warmup, 1)):
        table, m = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        table, m = step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ex.profile(True)
    table, m = step()
    torch.cuda.synchronize()
    stats = ex.profile_read()
    ex.profile(False)
    r, nj = int(table.n_records), int(table.n_junctions)
    dev_tab = ex.get_junctions()
    ex.close()
    ref = ora.table()
    ora.close()
    try:
        helpers.compare_tables(dev_tab, ref)
        parity = {"equals_oracle": True, "junctions": int(len(ref["intron_start"])), "alignments": n}
    except AssertionError as e:
        parity = {"equals_oracle": False, "error": str(e)[:300]}
    ms = elapsed / args.steps * 1e3
    b_alg = 20 + 4 * n_words / n + 64 * r / n
    return {"metric": "alignments/sec junction extraction", "value": n * args.steps / elapsed, "unit": "alignments/s",
            "n_gpus": 1, "steps": args.steps, "warmup": max(args.warmup, 1), "ms_per_step": ms, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u32/u64 integer", "data": "synthetic",
            "config": {"workload": "M1: one rank's share of 1 G short reads over 8 GPUs = %d alignments in 13 pushes (chunks of "
                                   "seeds %d..%d), one table of %d junctions" % (n, synth.SEED_M0, synth.SEED_M0 + 12, nj),
                       "alignments_per_gpu": n, "cigar_words": n_words, "records": r, "records_after_combine": int(m),
                       "junctions": nj, "generate_s": t_gen, "parity": parity,
                       "input": "SoA chunks resident in HBM before the timed region"},
            "roofline": {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "achieved": b_alg * n / (ms * 1e-3) / 1e9,
                         "frac": b_alg * n / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                         "kernels_ms_per_step": {k: v[1] for k, v in sorted(stats.items(), key=lambda kv: -kv[1][1])}},
            "cpu_baseline": {"value": n / t_cpu, "unit": "alignments/s", "cores": 1, "kind": "port",
                             "sample": "the oracle over the same 125 M alignments, one pass, %.1f s" % t_cpu}}


def run_group(args):
    """--multi group: BASELINE's config 5 through the PRODUCT's multi-GPU object, ipx_group (include/ipx.h, csrc/ipx_group.hip;
    what `intronProspector --devices=0-7` runs): ONE process, N members, member m on GPU m (or, on a box with fewer GPUs,
    several members sharing GPU 0 -- a rehearsal, and the line says so).  Member m holds rank m's contiguous 12.5 chunks of
    the M stream in ITS device's memory; a step = ipx_group_clear, every member's pieces pushed from device memory
    (ipx_group_push_member), ipx_group_finalize (scan per member, junction records to their key-range owners over RCCL --
    or device copies between members of one GPU --, every owner's sort + reduce, ONE table).  The same checks as the
    torch.distributed path: conservation, sjN a permutation, rows in key order, and -- on a down-scaled stream -- the
    whole table against the oracle's."""
    import torch
    import intronprospector_amd as ipa
    from intronprospector_amd import synth, config5
    from tests import helpers
    from tests.oracle_binding import Oracle
    members = args.gpus
    n_dev = torch.cuda.device_count()
    devices = list(range(members)) if n_dev >= members else [0] * members
    chunk = args.chunk or 10_000_000

    class Stream:
        def __init__(self, chunk_alignments, keep_host):
            self.ups, self.sizes, self.host, self.bases, self.names = [], [], [], [], None
            t0 = time.perf_counter()
            for m in range(members):
                dev = torch.device("cuda", devices[m])
                ups, sizes = [], []
                for c, lo, hi in config5.rank_pieces(m, chunk_alignments):
                    b, self.names = config5.make_piece(c, lo, hi, chunk_alignments)
                    ups.append(upload(b, dev))
                    sizes.append((len(b["tid"]), len(b["cigar"])))
                    if keep_host:
                        self.host.append(b)
                self.ups.append(ups)
                self.sizes.append(sizes)
                self.bases.append(config5.rank_base(m, chunk_alignments))
            self.n = sum(a for sz in self.sizes for a, _ in sz)
            self.n_words = sum(w for sz in self.sizes for _, w in sz)
            self.generate_s = time.perf_counter() - t0
            self.grp = ipa.JunctionsExtractorGroup(devices, targets=self.names)

        def step(self):
            self.grp.clear()
            for m in range(members):
                base = self.bases[m]
                for (na, nw), (_, ptrs) in zip(self.sizes[m], self.ups[m]):
                    self.grp.push_member_device(m, na, nw, ptrs, base)
                    base += na
            return self.grp.get_junctions()

    def checks_of(tab):
        J = len(tab["intron_start"])
        key = np.stack([tab["target"].astype(np.int64), tab["intron_start"].astype(np.int64), tab["intron_end"].astype(np.int64)], 1)
        asc = True
        if J > 1:
            a, b = key[:-1], key[1:]
            asc = bool(((a[:, 0] < b[:, 0]) | ((a[:, 0] == b[:, 0]) & ((a[:, 1] < b[:, 1]) | ((a[:, 1] == b[:, 1]) & (a[:, 2] < b[:, 2]))))).all())
        counted = int(tab["counts"].astype(np.uint64).sum())
        return {"junctions": J, "reads_counted": counted, "records_made": int(tab["n_records"]),
                "conservation": counted == int(tab["n_records"]),
                "ijunc_is_a_permutation": bool(np.array_equal(np.sort(tab["ijunc"].astype(np.int64)), np.arange(J))),
                "rows_in_key_order": asc}

    main_s = Stream(chunk, keep_host=False)
    for _ in range(max(args.warmup, 2)):
        tab = main_s.step()
    for d in set(devices):
        torch.cuda.synchronize(d)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tab = main_s.step()
    for d in set(devices):
        torch.cuda.synchronize(d)
    elapsed = time.perf_counter() - t0
    if "error" in tab:
        raise tab["error"]
    checks = checks_of(tab)
    uses_rccl = main_s.grp.uses_rccl()
    n, nw, gen_s = main_s.n, main_s.n_words, main_s.generate_s
    main_s.grp.close()
    del main_s

    # parity: the same code over a down-scaled stream, the whole table against the oracle's
    small_chunk = int(min(400_000, 20_000_000 // (12.5 * members))) // 2 * 2
    sm = Stream(small_chunk, keep_host=True)
    for _ in range(2):
        tab_s = sm.step()
    t0 = time.perf_counter()
    ora = Oracle(targets=sm.names)
    for b in sm.host:
        ora.process(b)
    ref = ora.table()
    ora.close()
    try:
        helpers.compare_tables(tab_s, ref)
        parity = {"equals_oracle": True}
    except AssertionError as e:
        parity = {"equals_oracle": False, "error": str(e)[:300]}
    parity.update({"alignments": sm.n, "junctions": int(len(ref["intron_start"])), "chunk_alignments": small_chunk,
                   "oracle_s": time.perf_counter() - t0, "checks": checks_of(tab_s)})
    sm.grp.close()
    ms = elapsed / args.steps * 1e3
    r = checks["records_made"]
    b_alg = 20 + 4 * nw / n + 64 * r / n
    per_gpu = len(set(devices))
    return {"metric": "alignments/sec junction extraction", "value": n * args.steps / elapsed, "unit": "alignments/s",
            "n_gpus": per_gpu, "steps": args.steps, "warmup": max(args.warmup, 2), "ms_per_step": ms, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u32/u64 integer", "data": "synthetic",
            "config": {"workload": "M through ipx_group: %d member(s) x 12.5 chunks of %d alignments (chunk i from seed %d + i) = %d short "
                                   "reads, one process, devices %s%s, %d junctions" % (
                                       members, chunk, synth.SEED_M0, n, devices,
                                       "" if per_gpu == members else " (REHEARSAL: members share a GPU)", checks["junctions"]),
                       "multi": "group", "members": members, "devices": devices, "uses_rccl": uses_rccl,
                       "alignments": n, "cigar_words": nw, "generate_s": gen_s, "checks": checks, "parity": parity,
                       "input": "SoA chunks resident in each member's HBM before the timed region; the table comes back to host memory "
                                "inside the timed step (ipx_group_finalize)"},
            "roofline": {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "achieved": b_alg * n / per_gpu / (ms * 1e-3) / 1e9, "frac": b_alg * n / per_gpu / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "traffic": None, "note": "per GPU: algorithmic bytes (20 + 4c + 64j per alignment) of the GPU's members / the step"}}


def run_m(args, rank, world, local_rank):
    """N > 1 (and --workload M): BASELINE.json's config 5.  Every rank holds its contiguous 12.5 chunks of the M stream
    (intronprospector_amd/config5.py: 125 M alignments at full size) resident in HBM; a step = ipx_clear + its pushes
    into ONE table + the record exchange over RCCL (all_to_all by key range) + ipx_finalize of the rank's key range +
    the global sjN numbers.  The line carries what can be checked at this size -- conservation, sjN a permutation, key
    ranges in order, one digest per range -- and `parity`: the SAME code over a down-scaled stream (at most 20 M
    alignments in all), every rank's rows gathered and compared with the oracle's table."""
    import torch
    import torch.distributed as dist
    import intronprospector_amd as ipa
    from intronprospector_amd import config5, distributed, synth, _lib as L
    from intronprospector_amd.distributed import RecordExchanger, table_to_torch, _all_gather_var
    if args.backend == "gloo":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if args.backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # RCCL over xGMI
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    comm_dev = dev if args.backend == "nccl" else torch.device("cpu")
    cpu_group = dist.new_group(backend="gloo")          # the checks after the timed region talk over CPU tensors

    class Stream:
        """one rank's share of a stream of the given chunk size (or explicit pieces), resident in HBM, and the step over it"""
        def __init__(self, chunk_alignments, pieces=None, base=None):
            t0 = time.perf_counter()
            self.pieces = pieces if pieces is not None else config5.rank_pieces(rank, chunk_alignments)
            self.base = base if base is not None else config5.rank_base(rank, chunk_alignments)
            self.ups, self.sizes, self.host = [], [], []
            for c, lo, hi in self.pieces:
                b, self.names = config5.make_piece(c, lo, hi, chunk_alignments)
                self.ups.append(upload(b, dev))
                self.sizes.append((len(b["tid"]), len(b["cigar"])))
                if chunk_alignments <= 1_000_000:
                    self.host.append(b)              # (the small stream: rank 0's oracle wants every rank's pieces)
            self.n = sum(a for a, _ in self.sizes)
            self.n_words = sum(w for _, w in self.sizes)
            self.generate_s = time.perf_counter() - t0
            self.ex = ipa.JunctionsExtractor(targets=self.names, device=local_rank)
            # (two handles that take turns: the scan of step k + 1 runs on a helper thread under the exchange of step k)
            self.ex2 = None if args.no_overlap else ipa.JunctionsExtractor(targets=self.names, device=local_rank)
            self.ahead = None
            self.cur = self.ex
            self.xch = RecordExchanger(dev, comm_device=comm_dev)
            self.xch.strict = False
            self.n_items = 0

        def scan(self, h):
            h.clear()
            base = self.base
            for (na, nw), (_, ptrs) in zip(self.sizes, self.ups):
                h.push_device(na, nw, ptrs, base_read_index=base)
                base += na

        def step(self):
            if self.ex2 is None:
                self.scan(self.ex)
            else:
                if self.ahead is None:
                    self.scan(self.ex)
                    self.cur = self.ex
                else:
                    self.cur, th = self.ahead
                    th.join()
                nxt = self.ex2 if self.cur is self.ex else self.ex
                self.ahead = (nxt, LookAhead(self.scan, nxt))
            self.n_items = self.cur.records_pending()
            # (a batch that outgrows the slots sized from the one before is completed by a second all_to_all inside run():
            # xch.slot_overflows counts those; nothing is scanned twice)
            return self.xch.run(self.cur, self.n)

        def close(self):
            if self.ahead is not None:
                self.ahead[1].join()
                self.ahead = None
            self.ex.close()
            if self.ex2 is not None:
                self.ex2.close()

        def timed(self, steps, warmup):
            for _ in range(max(warmup, 2)):           # (the first step draws splitters: the exact exchange; slots from the second on)
                self.step()
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                out = self.step()
            if self.ahead is not None:                 # the scan the last step started belongs to the timed region: steps scans, steps exchanges
                self.ahead[1].join()
            torch.cuda.synchronize()
            dist.barrier()
            elapsed = time.perf_counter() - t0
            self.xch.check_last()
            tmax = torch.tensor([elapsed], dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX, group=cpu_group)
            return float(tmax.item()), out

    chunk = args.chunk or 10_000_000
    main_s = Stream(chunk)
    elapsed, (table, ijunc) = main_s.timed(args.steps, args.warmup)
    t = table_to_torch(main_s.cur, table, "cpu")
    checks = config5.cross_rank_checks(t, ijunc.cpu(), int(table.n_records), group=cpu_group)
    checks["slot_overflows_completed_by_second_all_to_all"] = main_s.xch.slot_overflows
    per_rank = {"alignments": main_s.n, "cigar_words": main_s.n_words, "records_after_combine": int(main_s.n_items),
                "pushes": len(main_s.pieces), "generate_s": main_s.generate_s}
    # kernel classes of one more step (of the handle whose turn it is)
    if main_s.ahead is not None:
        main_s.ahead[1].join()
    prof_h = main_s.ahead[0] if main_s.ahead is not None else main_s.ex
    other_h = main_s.cur if main_s.ahead is not None else None
    if other_h is not None:
        other_h.profile(True)                      # (its scan of the step after runs under this step's exchange)
    prof_h.profile(True)
    main_s.step()
    if main_s.ahead is not None:
        main_s.ahead[1].join()
    torch.cuda.synchronize()
    stats = prof_h.profile_read()
    if other_h is not None:
        for k, v in other_h.profile_read().items():
            stats[k] = (stats[k][0] + v[0], stats[k][1] + v[1]) if k in stats else v
    main_s.close()
    del main_s.ups

    # ---- parity: the same code over a down-scaled stream, all ranks' rows against the oracle's table
    small_chunk = int(min(400_000, 20_000_000 // (12.5 * world))) // 2 * 2
    sm = Stream(small_chunk)
    for _ in range(3):                                 # (exact exchange, then twice in slots)
        tab_s, ij_s = sm.step()
    sm.xch.check_last()
    ts = table_to_torch(sm.cur, tab_s, "cpu")
    ts["ijunc"] = ij_s.cpu()
    small_checks = config5.cross_rank_checks(ts, ij_s.cpu(), int(tab_s.n_records), group=cpu_group)
    gathered = {k: torch.cat(_all_gather_var(ts[k], group=cpu_group)) for k in
                ("target", "intron_start", "intron_end", "anchor_start", "anchor_end", "strand", "ijunc", "counts",
                 "hist_value", "hist_count")}
    all_host = [None] * world
    dist.all_gather_object(all_host, sm.host, group=cpu_group)
    parity = None
    if rank == 0:
        from tests.oracle_binding import Oracle
        t0 = time.perf_counter()
        ora = Oracle(targets=sm.names)
        n_small = 0
        for pieces in all_host:
            for b in pieces:
                ora.process(b)
                n_small += len(b["tid"])
        ref = ora.table()
        ora.close()
        oo = np.lexsort((ref["intron_end"], ref["intron_start"], ref["target"]))
        bad = [k for k in ("target", "intron_start", "intron_end", "anchor_start", "anchor_end", "strand", "ijunc")
               if not np.array_equal(gathered[k].numpy().astype(np.int64), ref[k][oo].astype(np.int64))]
        if not np.array_equal(gathered["counts"].numpy().astype(np.uint64), ref["counts"][oo]):
            bad.append("counts")
        starts = np.concatenate([[0], np.cumsum(ref["n_offsets"].astype(np.int64))])
        flat = np.concatenate([ref["offsets"][starts[i]:starts[i + 1]] for i in oo]) if len(oo) else np.zeros(0, np.int64)
        got = np.repeat(gathered["hist_value"].numpy(), gathered["hist_count"].numpy())
        if not np.array_equal(flat.astype(np.int64), got.astype(np.int64)):
            bad.append("offsets")
        parity = {"equals_oracle": not bad, "differs_in": bad, "alignments": n_small, "junctions": int(len(oo)),
                  "chunk_alignments": small_chunk, "oracle_s": time.perf_counter() - t0, "checks": small_checks}
    sm.close()
    del sm

    # ---- round 3's N > 1 line, kept as a sub-record: 10 M alignments per rank (seeds SEED_M0 + rank), one push per step
    weak = None
    if not args.no_configs:
        ws = Stream(10_000_000, pieces=[(rank, 0, 10_000_000)], base=rank * 10_000_000)
        w_el, _ = ws.timed(max(args.steps // 2, 3), 2)
        weak = {"ms_per_step": w_el / max(args.steps // 2, 3) * 1e3, "alignments_per_gpu": ws.n,
                "value": world * ws.n * max(args.steps // 2, 3) / w_el,
                "workload": "S-sized: 10 M alignments per GPU (seed SEED_M0 + rank), one push, exchange, finalize"}
        ws.close()
    dist.destroy_process_group()
    if rank != 0:
        return None
    ms = elapsed / args.steps * 1e3
    n, nw, r = per_rank["alignments"], per_rank["cigar_words"], checks["records_made"] / world
    b_alg = 20 + 4 * nw / n + 64 * r / n
    out = {"metric": "alignments/sec junction extraction", "value": world * n * args.steps / elapsed, "unit": "alignments/s",
           "n_gpus": world, "steps": args.steps, "warmup": max(args.warmup, 2), "ms_per_step": ms, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "u32/u64 integer", "data": "synthetic",
           "config": {"workload": "M: BASELINE config 5 -- %d GPU(s) x 12.5 chunks of %d alignments (chunk i from seed %d + i) = %d "
                                  "short reads in all, every rank's share pushed into one table, junction records exchanged by key "
                                  "range over %s, %d junctions" % (world, chunk, synth.SEED_M0, world * n,
                                                                  "RCCL" if args.backend == "nccl" else "gloo (rehearsal)",
                                                                  checks["junctions"]),
                      "alignments_per_gpu": n, "per_rank": per_rank, "multi": "exchange", "checks": checks, "parity": parity,
                      "overlap": (None if args.no_overlap else "the scan of step k + 1 (second handle, helper thread) runs under the "
                                  "exchange and reduce of step k"),
                      "input": "SoA chunks resident in HBM before the timed region (decode and H2D excluded)"},
           "roofline": {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "achieved": b_alg * n / (ms * 1e-3) / 1e9,
                        "frac": b_alg * n / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                        "note": "per GPU: the rank's algorithmic bytes (20 + 4c + 64j per alignment) / the step, exchange included",
                        "kernels_ms_per_step": {k: v[1] for k, v in sorted(stats.items(), key=lambda kv: -kv[1][1])}}}
    if weak is not None:
        out["configs"] = {"weak_S": weak}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=None, choices=["S", "L", "G", "M1", "M"],
                    help="default: S on one GPU, M on several.  M: BASELINE's config 5 (1 G short reads over 8 GPUs): every rank "
                         "its contiguous 12.5 chunks of 10 M alignments in one table, record exchange over RCCL.  M1: ONE rank's "
                         "share of it without an exchange, the whole table against the oracle")
    ap.add_argument("--chunk", type=int, default=0, help="workload M: alignments per chunk (default 10 M; smaller: rehearsals)")
    ap.add_argument("--reads", type=int, default=0, help="alignments per GPU (default: 10 M for S and G, 2 M for L)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="alignments for the CPU baseline (0 = auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the device-vs-oracle check of a prefix (profiling runs)")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the `configs` sub-records (L at 2 M, G at 10 M) of the default single-GPU S run")
    ap.add_argument("--shuffle", action="store_true",
                    help="alignments in random order instead of coordinate order (the reference's whole-file mode "
                         "accepts unsorted input): what the step costs without any locality")
    ap.add_argument("--multi", default="exchange", choices=["exchange", "group"],
                    help="N>1.  exchange (default; what the driver's torch.distributed launch measures): one process per GPU, this "
                         "package's harness over libipx's slot exchange entry points and torch.distributed collectives.  group: ONE "
                         "process drives --gpus members through ipx_group (the product's --devices path; run WITHOUT "
                         "torch.distributed.run; members share GPU 0 on a box with fewer GPUs)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: rehearsal of the N>1 path on a one-GPU box (tables merged on the CPU)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="N>1 / rehearsal: one handle, scan and exchange one after the other (round 3's step)")
    ap.add_argument("--rehearse-collectives", action="store_true",
                    help="N=1 only: run the N>1 code path (process group of one rank) to exercise the RCCL calls")
    args = ap.parse_args()

    # stdout carries ONE line, the JSON: libraries that print there on their own (RCCL's version banner when the
    # first communicator is made) are sent to stderr until the line goes out
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.multi == "group":
        if world != 1:
            raise SystemExit("--multi group is one process: run it without torch.distributed.run")
        import torch            # before the package: libipx.so must bind to the HIP runtime torch brings along
        out = run_group(args)
        sys.stdout.flush()
        os.dup2(json_fd, 1)
        print(json.dumps(out), flush=True)
        return
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)

    import torch            # before the package: libipx.so must bind to the HIP runtime torch brings along
    from intronprospector_amd import synth
    kind = args.workload or ("M" if world > 1 else "S")
    if kind == "M":
        out = run_m(args, rank, world, local_rank)
        sys.stdout.flush()
        os.dup2(json_fd, 1)
        if rank == 0:
            print(json.dumps(out), flush=True)
        return
    if kind == "M1":
        out = run_m1(args, local_rank)
        sys.stdout.flush()
        os.dup2(json_fd, 1)
        print(json.dumps(out), flush=True)
        return
    reads = args.reads or DEFAULT_READS[kind]
    seed = (synth.SEED_L if kind == "L" else synth.SEED_S) if world == 1 else synth.SEED_M0 + rank
    t_gen = time.perf_counter()
    batch, names, genome = make_workload(kind, reads, seed)
    if args.shuffle:
        batch = synth.permute_batch(batch, np.random.default_rng(seed + 7).permutation(len(batch["tid"])))
    t_gen = time.perf_counter() - t_gen
    n = len(batch["tid"])
    n_words = len(batch["cigar"])

    # CPU baseline first: its worker processes start before this process touches the GPU
    cpu = None
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        cpu = cpu_baseline(batch, names, kind, args.cpu_sample or n)

    import torch.distributed as dist
    if args.backend == "gloo":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    multi = world > 1 or args.rehearse_collectives
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # RCCL over xGMI
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    import intronprospector_amd as ipa
    from intronprospector_amd import _lib as L

    dev_arrays, ptrs = upload(batch, dev)
    params = _oracle_params(kind)
    genome_pack = None
    if genome is not None:      # device 2-bit genome; filter as the command line defaults to with a genome
        genome_pack = (genome.names, [(t, m, len(q)) for (t, m), q in zip(genome.twobit(), genome.seqs)])
        params["genome"] = genome_pack
        params["junction_filter"] = L.IPX_SJ_CANON
    ex = ipa.JunctionsExtractor(targets=names, device=local_rank, **params)
    ex.reserve(0, 0, n_words // 2 + 1024)

    # (the table-merge formulation of the exchange step -- all_gather of key lists, all_reduce of dense tables: SURVEY 8e
    # steps 2-4 -- was a bench option through round 4: 440 ms against the record exchange's 57 ms on two ranks.  Retired
    # from the bench in round 5; distributed.merge_shard_tables stays as the tests' second formulation)
    exchanger = None
    if multi:
        from intronprospector_amd import distributed
        comm_dev = dev if args.backend == "nccl" else torch.device("cpu")
        exchanger = distributed.RecordExchanger(dev, comm_device=comm_dev)
        exchanger.strict = False              # the ranks' finalize errors are looked at once, after the timed steps

    n_records = [0]
    n_merged = [0]
    n_items = [0]
    # The exchange path runs on TWO handles that take turns (round 4): while the records of batch k travel, are adopted and
    # reduced on handle A -- a chain of small kernels, a collective and three read-backs on the exchanger's stream --,
    # a helper thread scans batch k + 1 on handle B's own stream.  (ctypes releases the interpreter lock during a call;
    # different handles are independent, include/ipx.h.)  --no-overlap: one handle, one thing after the other, as in round 3.
    overlap = exchanger is not None and not args.no_overlap
    ex2 = None
    scanned = [None]          # the handle whose scan of the NEXT batch is under way (or done), and the thread doing it
    if overlap:
        ex2 = ipa.JunctionsExtractor(targets=names, device=local_rank, **params)
        ex2.reserve(0, 0, n_words // 2 + 1024)

    def scan(h):
        h.clear()
        h.push_device(n, n_words, ptrs, base_read_index=rank * n)

    def step():
        if overlap:
            if scanned[0] is None:                # the first step scans its own batch
                scan(ex)
                cur = ex
            else:
                cur, th = scanned[0]
                th.join()
            nxt = ex2 if cur is ex else ex
            scanned[0] = (nxt, LookAhead(scan, nxt))
        else:
            cur = ex
            scan(cur)
        n_items[0] = cur.records_pending()
        if exchanger is not None:
            # junction records -> key-range owners over RCCL (all_to_all), then the local finalize
            table, ijunc = exchanger.run(cur, n)
            n_records[0] = table.n_records
            n_merged[0] = int(table.n_junctions)          # this rank's key range; summed after the timed loop
            return table
        rc, table = ex.finalize_raw(L.IPX_MEM_DEVICE)
        if rc:
            ex._check(rc)
        n_records[0] = table.n_records
        return table

    # A probe step with event pairs around every launch tells which kernel class dominates.  The timed region
    # then brackets the launches of that class alone (an event before and after each): pairs around all
    # launches of a step cost a 10 M-alignment step about 10 % of its time.
    def both(f):                  # (the exchange path alternates between two handles: their kernel timers are read together)
        a = f(ex)
        if ex2 is not None:
            b = f(ex2)
            if isinstance(a, dict):
                for k, v in b.items():
                    a[k] = (a[k][0] + v[0], a[k][1] + v[1]) if k in a else v
        return a
    for _ in range(args.warmup):
        table = step()
    both(lambda h: h.profile(True))   # one more untimed step, all launches bracketed (the first steps also allocate)
    table = step()
    if ex2 is not None:
        table = step()
    torch.cuda.synchronize()
    warm = both(lambda h: h.profile_read())
    if ex2 is not None:
        warm = {k: (v[0], v[1] / 2) for k, v in warm.items()}
    dominant = pick_dominant(warm, "k_scan" if kind == "L" else "k_scatter")
    both(lambda h: h.profile_only(dominant))
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        table = step()
    if scanned[0] is not None:           # the scan the last step started belongs to the timed region: steps scans, steps exchanges
        scanned[0][1].join()
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if exchanger is not None:
        exchanger.check_last()
    timed = both(lambda h: h.profile_read())
    # per-kernel breakdown: a few more steps, untimed, with event pairs around every launch
    breakdown_steps = 4 if ex2 is not None else 3
    both(lambda h: h.profile(True))
    for _ in range(breakdown_steps):
        table = step()
    if scanned[0] is not None:
        scanned[0][1].join()
    torch.cuda.synchronize()
    stats = both(lambda h: h.profile_read())
    both(lambda h: h.profile(False))
    if multi:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    if exchanger is not None:
        tot = torch.tensor([n_merged[0], int(n_records[0])], dtype=torch.int64, device=comm_dev)
        dist.all_reduce(tot)
        n_merged[0] = int(tot[0].item())
        n_records[0] = int(tot[1].item()) // world       # records per GPU (after the exchange a rank holds a key range)
    n_junctions = n_merged[0] if multi else int(table.n_junctions)
    r = int(n_records[0])
    ex.close()
    if ex2 is not None:
        ex2.close()

    out = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * n * args.steps / elapsed
        c = n_words / n
        j = r / n
        roofline = roofline_block(kind, n, n_words, r, n_items[0], ms_per_step, dominant, timed, stats, breakdown_steps,
                                  args.steps)
        out = {
            "metric": "alignments/sec junction extraction", "value": value, "unit": "alignments/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32/u64 integer",
            "data": "synthetic",
            "config": {"workload": "%s: %d synthetic %s per GPU, %.2f CIGAR ops and %.2f junction records per "
                                   "alignment, %d junctions in the table" %
                                   (kind, n, "ONT-style long reads" if kind == "L" else "100 bp short reads (1-2 N ops)" +
                                    (", device 2-bit genome, --sj-filter=canon" if kind == "G" else ""),
                                    c, j, n_junctions),
                       "alignments_per_gpu": n, "cigar_words": n_words, "records": r, "junctions": n_junctions,
                       "input": "SoA batch resident in HBM before the timed region (decode and H2D excluded)",
                       "order": "shuffled" if args.shuffle else "coordinate-sorted", "seed": seed, "generate_s": t_gen,
                       "multi": (args.multi if multi else None),
                       "overlap": ("the scan of batch k + 1 (second handle, helper thread) runs under the exchange and reduce of "
                                   "batch k" if overlap else None)},
            "roofline": roofline,
        }
        if cpu is not None:
            out["cpu_baseline"] = cpu
        if world == 1 and not multi and not args.shuffle and not args.no_parity:
            try:        # the benchmark's own batch, a prefix of it, device against oracle
                out["config"]["parity_prefix"] = parity_prefix(kind, batch, names, genome_pack, genome,
                                                               20_000 if kind == "L" else 200_000, local_rank)
            except Exception as e:
                out["config"]["parity_prefix"] = {"error": repr(e)[:300]}
    # BASELINE.json's other single-GPU configurations, in the same run
    if rank == 0 and world == 1 and not multi and kind == "S" and not args.no_configs and not args.shuffle and not args.reads:
        del dev_arrays, batch
        configs = {}
        for k2, steps2, prefix in (("G", 10, 200_000), ("L", 5, 20_000)):
            t1 = time.perf_counter()
            try:
                b2, names2, g2 = make_workload(k2, DEFAULT_READS[k2], synth.SEED_L if k2 == "L" else synth.SEED_S)
                gen_s = time.perf_counter() - t1
                rec = run_single(k2, b2, names2, g2, dev, local_rank, steps2, 2, prefix)
                rec["generate_s"] = gen_s
                configs[k2] = rec
                del b2, g2
            except Exception as e:                   # the headline line must still come out
                configs[k2] = {"error": repr(e)[:300]}
        # the headline workload again, three batches of different seeds taking turns: what the per-handle hints cost
        # when they miss (the headline replays ONE batch, so every hint is a hit)
        t1 = time.perf_counter()
        try:
            bs = [make_workload("S", DEFAULT_READS["S"], synth.SEED_S + k)[0] for k in range(3)]
            gen_s = time.perf_counter() - t1
            rec = run_single("S", bs[0], names, None, dev, local_rank, 12, 3, 0, extra_batches=bs[1:])
            rec["generate_s"] = gen_s
            rec["seeds"] = [synth.SEED_S + k for k in range(3)]
            configs["S_rotating"] = rec
            del bs
        except Exception as e:
            configs["S_rotating"] = {"error": repr(e)[:300]}
        out["configs"] = configs
    if multi:
        dist.destroy_process_group()
    sys.stdout.flush()
    os.dup2(json_fd, 1)
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
