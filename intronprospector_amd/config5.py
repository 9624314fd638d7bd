"""BASELINE.json's config 5 -- "1 B synthetic short reads sharded across 8 x MI355X" -- as bench.py and the tests run it.

The stream (workload M): 100 chunks of `chunk_alignments` coordinate-sorted short reads, chunk i
generated from seed SEED_M0 + i (intronprospector_amd/synth.py); the full size has 10 M alignments per chunk = 1 G in
all.  Rank r of N takes a CONTIGUOUS share of 12.5 chunks -- global alignments [r, r + 1) x 12.5 x chunk_alignments --
so N = 8 is the whole stream and N = 1, 2, 4 are its first eighth, quarter, half (weak scaling: the per-GPU work is
the same at every N).  A rank pushes its pieces into ONE table, the junction records go to their key-range owners
(distributed.RecordExchanger) and every rank finalizes its range.

Nothing here touches the GPU: the arithmetic of the shares and the cross-rank checks that the bench line carries
(conservation, global sjN a permutation, one digest per key range) run under gloo on CPU tensors as well
(tests/test_distributed_gloo.py).  Test and bench harness only, like the rest of this package.
"""
import hashlib

import numpy as np
import torch
import torch.distributed as dist

HALF_CHUNKS_PER_RANK = 25          # 12.5 chunks
N_CHUNKS = 100


def rank_pieces(rank, chunk_alignments):
    """[(chunk index, first alignment of the chunk taken, one past the last)] of rank's share, in stream order."""
    half = chunk_alignments // 2
    out = []
    for h in range(HALF_CHUNKS_PER_RANK * rank, HALF_CHUNKS_PER_RANK * (rank + 1)):
        c, second = h // 2, h % 2
        lo, hi = (half, chunk_alignments) if second else (0, half)
        if out and out[-1][0] == c and out[-1][2] == lo:
            out[-1] = (c, out[-1][1], hi)
        else:
            out.append((c, lo, hi))
    return out


def rank_base(rank, chunk_alignments):
    """global read index of the first alignment of rank's share"""
    return HALF_CHUNKS_PER_RANK * rank * (chunk_alignments // 2)


def share_alignments(chunk_alignments):
    return HALF_CHUNKS_PER_RANK * (chunk_alignments // 2)


def make_piece(chunk, lo, hi, chunk_alignments):
    """the alignments [lo, hi) of chunk `chunk` as a self-contained SoA batch (+ target names)"""
    from . import synth
    batch, names, _ = synth.short_reads(chunk_alignments, seed=synth.SEED_M0 + chunk)
    if lo != 0 or hi != chunk_alignments:
        batch = synth.slice_batch(batch, lo, hi)
    return batch, names


def range_digest(t, ijunc):
    """SHA-256 over one rank's key range as table_to_torch gives it (CPU tensors) and its global sjN numbers"""
    h = hashlib.sha256()
    for k in ("target", "intron_start", "intron_end", "anchor_start", "anchor_end", "strand"):
        h.update(np.ascontiguousarray(t[k].cpu().numpy().astype(np.int64)).tobytes())
    h.update(np.ascontiguousarray(ijunc.cpu().numpy().astype(np.int64)).tobytes())
    h.update(np.ascontiguousarray(t["counts"].cpu().numpy().astype(np.uint64)).tobytes())
    h.update(np.ascontiguousarray(t["hist_value"].cpu().numpy().astype(np.int64)).tobytes())
    h.update(np.ascontiguousarray(t["hist_count"].cpu().numpy().astype(np.int64)).tobytes())
    return h.hexdigest()


def cross_rank_checks(t, ijunc, n_raw_local, filter_active=False, group=None):
    """What holds for ANY input of a run that exchanged records by key range, whatever its size (every rank calls; CPU
    tensors; the collectives go over `group`, gloo or RCCL-with-CPU-staging as the caller set it up):
      conservation   the reads the tables count = the junction records the scans made, summed over the ranks
                     (junctions_extractor.cc:466-485: every record is counted once; only without an active junction
                     filter, which drops the records before a junction's creator)
      sjN            the global ijunc values of all ranks are a permutation of 0 .. J - 1 (junctions_extractor.cc:459-460)
      key order      every rank's rows ascend by (target, start, end) and rank r's last row lies below rank r + 1's first
      digests        one SHA-256 per rank over its rows (a reader can compare two runs range by range)
    -> dict, the same on every rank."""
    from .distributed import _all_gather_var
    world = dist.get_world_size(group)
    counted = int(t["counts"].sum().item())
    tot = torch.tensor([counted, int(n_raw_local), int(t["target"].shape[0])], dtype=torch.int64)
    dist.all_reduce(tot, group=group)
    J = int(tot[2].item())
    all_ij = torch.cat(_all_gather_var(ijunc.to(torch.int64).cpu(), group=group))
    perm = bool(all_ij.numel() == J and (J == 0 or bool(torch.equal(torch.sort(all_ij).values, torch.arange(J, dtype=torch.int64)))))
    n = int(t["target"].shape[0])
    asc = True
    if n > 1:
        a = torch.stack([t["target"][:-1], t["intron_start"][:-1], t["intron_end"][:-1]], 1).cpu().numpy().astype(np.int64)
        b = torch.stack([t["target"][1:], t["intron_start"][1:], t["intron_end"][1:]], 1).cpu().numpy().astype(np.int64)
        lt = (a[:, 0] < b[:, 0]) | ((a[:, 0] == b[:, 0]) & ((a[:, 1] < b[:, 1]) | ((a[:, 1] == b[:, 1]) & (a[:, 2] < b[:, 2]))))
        asc = bool(lt.all())
    ends = torch.full((2, 3), -1, dtype=torch.int64)
    if n:
        for j, k in enumerate(("target", "intron_start", "intron_end")):
            ends[0, j] = int(t[k][0].item())
            ends[1, j] = int(t[k][-1].item())
    all_ends = [torch.empty_like(ends) for _ in range(world)]
    dist.all_gather(all_ends, ends, group=group)
    flags = torch.tensor([1 if asc else 0], dtype=torch.int64)
    dist.all_reduce(flags, op=dist.ReduceOp.MIN, group=group)
    ordered, last = bool(flags.item()), None
    for e in all_ends:
        if int(e[0, 0].item()) < 0:
            continue                                   # a rank without rows
        first = tuple(int(x) for x in e[0])
        if last is not None and not last < first:
            ordered = False
        last = tuple(int(x) for x in e[1])
    dig = range_digest(t, ijunc)
    digs = [None] * world
    dist.all_gather_object(digs, dig, group=group)
    out = {"junctions": J, "reads_counted": int(tot[0].item()), "records_made": int(tot[1].item()),
           "ijunc_is_a_permutation": perm, "key_ranges_in_order": ordered, "range_digests": digs}
    out["conservation"] = None if filter_active else bool(tot[0].item() == tot[1].item())
    return out
