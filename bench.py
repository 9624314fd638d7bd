#!/usr/bin/env python3
"""Benchmark of the junction-extraction hot path on MI355X.

A step is one pass of the path over one batch of synthetic alignments that is
already resident in HBM: ipx_clear -> ipx_push_batch (CIGAR scan -> junction
records) -> ipx_finalize (radix sort + segmented reduce -> junction table in
HBM).  With N > 1 every rank holds its own contiguous shard of N x reads
alignments (weak scaling) and the partial tables are combined over RCCL.

Timing: W warm-up steps, then exactly K timed steps between barrier + synchronize.  Inside the timed
region HIP events bracket the launches of the dominant kernel class only (roofline.avg_ms); the full
per-kernel table (roofline.kernels_ms_per_step) comes from three further, untimed steps with events
around every launch.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement").
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


def make_workload(kind, reads, seed):
    from intronprospector_amd import synth
    if kind == "S":
        return synth.short_reads(reads, seed=seed)
    if kind == "G":     # S with a synthetic genome: strand / motif inference, --sj-filter=canon
        return synth.short_reads(reads, seed=seed, genome=True)
    if kind == "L":
        return synth.long_reads(reads, seed=seed)
    raise SystemExit("unknown workload %r" % kind)


# algorithmic bytes each kernel has to move per launch (DESIGN.md "Kernels"):
# n alignments, w CIGAR words, r records
KERNEL_BYTES = {
    "k_scan": lambda n, w, r: 20 * n + 4 * w + 32 * r,     # SoA + CIGAR in, 32-byte records out
    "k_hist": lambda n, w, r: 1 * r,                        # one digit byte per record
    "k_scatter": lambda n, w, r: 33 * r,                    # item in, item + next digit byte out
    "k_count_heads": lambda n, w, r: 16 * r,
    "k_write_heads": lambda n, w, r: 16 * r,
    "k_reduce_a": lambda n, w, r: 32 * r,                   # item + payload of every record
    "k_reduce": lambda n, w, r: 32 * r,
}


def cpu_baseline(batch, names, kind, budget_reads, budget_s=10.0):
    """The oracle (a scalar port of the reference, oracle/) timed on this host, one core: whole
    passes over the first budget_reads alignments of the same batch (a fresh table each pass) until
    about budget_s seconds of CPU work are done.  Reported next to the GPU number only."""
    from intronprospector_amd import synth
    from tests.oracle_binding import Oracle
    n = min(budget_reads, len(batch["tid"]))
    sample = synth.slice_batch(batch, 0, n)
    params = {"max_anchor_indel_size": 36} if kind == "L" else {}
    passes, spent, nj = 0, 0.0, 0
    while spent < budget_s and passes < 200:
        ora = Oracle(targets=names, **params)
        t0 = time.perf_counter()
        ora.process(sample)
        nj = ora.lib.ipxo_num_junctions(ora.h)
        spent += time.perf_counter() - t0
        ora.close()
        passes += 1
    return {"value": n * passes / spent, "unit": "alignments/s", "cores": 1, "kind": "port",
            "sample": "%d passes over %s %d alignments of the %s workload, oracle/libipxo.so (scalar C restatement), "
                      "%d junctions, %.1f s of CPU work" % (passes, "all" if n == len(batch["tid"]) else "the first",
                                                            n, kind, nj, spent)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="S", choices=["S", "L", "G"])
    ap.add_argument("--reads", type=int, default=0, help="alignments per GPU (default: 10M for S, 200k for L)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="alignments for the CPU baseline (0 = auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--shuffle", action="store_true",
                    help="alignments in random order instead of coordinate order (the reference's whole-file mode "
                         "accepts unsorted input): what the step costs without any locality")
    ap.add_argument("--multi", default="exchange", choices=["exchange", "tables"],
                    help="N>1: exchange junction records by key range (default) or merge shard tables")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: rehearsal of the N>1 path on a one-GPU box (tables merged on the CPU)")
    ap.add_argument("--rehearse-collectives", action="store_true",
                    help="N=1 only: run the N>1 code path (process group of one rank) to exercise the RCCL calls")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    import torch
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
    from intronprospector_amd import synth
    from intronprospector_amd import _lib as L

    kind = args.workload
    reads = args.reads or (200_000 if kind == "L" else 10_000_000)
    seed = (synth.SEED_L if kind == "L" else synth.SEED_S) if world == 1 else synth.SEED_M0 + rank
    t_gen = time.perf_counter()
    batch, names, genome = make_workload(kind, reads, seed)
    if args.shuffle:
        import numpy as np
        batch = synth.permute_batch(batch, np.random.default_rng(seed + 7).permutation(len(batch["tid"])))
    t_gen = time.perf_counter() - t_gen
    n = len(batch["tid"])
    n_words = len(batch["cigar"])

    keys = ("tid", "pos", "flag", "xs", "nh", "cigar_off", "n_cigar", "cigar")
    dev_arrays = [torch.from_numpy(batch[k]).to(dev) for k in keys]
    ptrs = [t.data_ptr() for t in dev_arrays]
    torch.cuda.synchronize()

    params = {"max_anchor_indel_size": 36} if kind == "L" else {}
    if genome is not None:      # device 2-bit genome; filter as the command line defaults to with a genome
        params["genome"] = (genome.names, [(t, m, len(q)) for (t, m), q in zip(genome.twobit(), genome.seqs)])
        params["junction_filter"] = L.IPX_SJ_CANON
        genome = None
    ex = ipa.JunctionsExtractor(targets=names, device=local_rank, **params)
    ex.reserve(0, 0, n_words // 2 + 1024)

    merger = None
    exchanger = None
    if multi:
        from intronprospector_amd import distributed
        comm_dev = dev if args.backend == "nccl" else torch.device("cpu")
        if args.multi == "tables":
            merger = distributed.TableMerger(comm_dev)
        else:
            exchanger = distributed.RecordExchanger(dev, comm_device=comm_dev)

    n_records = [0]
    n_merged = [0]

    def step():
        ex.clear()
        ex.push_device(n, n_words, ptrs, base_read_index=rank * n)
        if exchanger is not None:
            # junction records -> key-range owners over RCCL (all_to_all), then the local finalize
            table, ijunc = exchanger.run(ex, n)
            n_records[0] = table.n_records
            n_merged[0] = int(table.n_junctions)          # this rank's key range; summed after the timed loop
            return table
        if merger is not None:
            # partial tables of this rank's shard -> RCCL merge of the junction tables
            merged, table = merger.merge(ex, rank * n)
            n_records[0] = table.n_records
            n_merged[0] = int(merged["target"].shape[0])
            return table
        rc, table = ex.finalize_raw(L.IPX_MEM_DEVICE)
        if rc:
            ex._check(rc)
        n_records[0] = table.n_records
        return table

    # A probe step with event pairs around every launch tells which kernel class dominates.  The timed region
    # then brackets the launches of that class alone (an event before and after each): pairs around all
    # ~70 launches of a step cost a 10 M-alignment step about 10 % of its time.
    for _ in range(args.warmup):
        table = step()
    ex.profile(True)              # one more untimed step, all launches bracketed (the first steps also allocate)
    table = step()
    torch.cuda.synchronize()
    warm = ex.profile_read()
    bracketed = {"k_scan", "k_scatter", "k_reduce"}
    cand = {k: v for k, v in warm.items() if k in bracketed}
    dominant = max(cand.items(), key=lambda kv: kv[1][1])[0] if cand else ("k_scan" if kind == "L" else "k_scatter")
    ex.profile_only(dominant)
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        table = step()
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    timed = ex.profile_read()
    # per-kernel breakdown: a few more steps, untimed, with event pairs around every launch
    breakdown_steps = 3
    ex.profile(True)
    for _ in range(breakdown_steps):
        table = step()
    torch.cuda.synchronize()
    stats = ex.profile_read()
    ex.profile(False)
    if multi:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    if exchanger is not None:
        tot = torch.tensor([n_merged[0]], dtype=torch.int64, device=comm_dev)
        dist.all_reduce(tot)
        n_merged[0] = int(tot.item())
    n_junctions = n_merged[0] if multi else int(table.n_junctions)
    r = int(n_records[0])

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * n * args.steps / elapsed
        c = n_words / n
        j = r / n
        b_alg = 20 + 4 * c + 64 * j                      # SURVEY.md 8(d), bytes per alignment
        dev_ms = sum(v[1] for v in stats.values()) / breakdown_steps
        dom = (dominant, timed[dominant]) if dominant in timed else None
        roofline = None
        if dom:
            name, (launches, total_ms) = dom
            avg_ms = total_ms / launches
            kb = KERNEL_BYTES.get(name, lambda n_, w_, r_: 0)(n, n_words, r)
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tpath):
                try:
                    traffic = json.load(open(tpath)).get(kind, {}).get(name)
                except Exception:
                    traffic = None
            roofline = {
                "bound": "hbm", "kernel": name, "launches_per_step": launches / args.steps,
                "avg_ms": avg_ms, "algorithmic_bytes_per_launch": kb,
                "achieved": kb / (avg_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": kb / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
                # whole path: SURVEY 8(d) algorithmic bytes over ALL device time of a step
                "pipeline": {"bytes_per_alignment": b_alg, "device_ms_per_step": dev_ms,
                             "achieved": b_alg * n / (dev_ms * 1e-3) / 1e9 if dev_ms else None,
                             "frac": b_alg * n / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if dev_ms else None},
                # from breakdown_steps extra steps after the timed region (every launch bracketed there)
                "kernels_ms_per_step": {k: v[1] / breakdown_steps for k, v in sorted(stats.items(), key=lambda kv: -kv[1][1])},
            }
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
                       "order": "shuffled" if args.shuffle else "coordinate-sorted", "seed": seed, "generate_s": t_gen},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            sample = args.cpu_sample or n
            out["cpu_baseline"] = cpu_baseline(batch, names, kind, sample)
        print(json.dumps(out))
    ex.close()
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
