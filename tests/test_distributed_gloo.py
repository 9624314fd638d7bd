"""Multi-rank table merge on CPU (gloo, world_size 2 and 3): shard tables produced by
the oracle, merged by the product's protocol, must equal a single oracle run."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_table(batch, names, params, fasta):
    from tests.oracle_binding import Oracle
    ora = Oracle(targets=names, genome_fasta=fasta, **params)
    ora.process(batch)
    t = ora.table()
    ora.close()
    return t


def _to_torch(t, rank):
    """oracle table -> the dict merge_shard_tables takes; creation order surrogate
    (rank << 32 | local ijunc) orders creations like global read indices do for
    contiguous shards."""
    n = len(t["intron_start"])
    row = np.repeat(np.arange(n), t["n_offsets"].astype(np.int64))
    pair = (row.astype(np.int64) << 16) | t["offsets"].astype(np.int64)
    ub, cnt = np.unique(pair, return_counts=True)
    hb = np.zeros(n + 1, np.int64)
    np.cumsum(np.bincount(ub >> 16, minlength=n), out=hb[1:])
    i64 = lambda a: torch.from_numpy(np.ascontiguousarray(a).astype(np.int64))
    return {
        "target": i64(t["target"]), "intron_start": i64(t["intron_start"]), "intron_end": i64(t["intron_end"]),
        "anchor_start": i64(t["anchor_start"]), "anchor_end": i64(t["anchor_end"]), "strand": i64(t["strand"]),
        "motif": torch.zeros(n, dtype=torch.int64), "counts": i64(t["counts"]),
        "creator_order": (rank << 32) | i64(t["ijunc"]),
        "hist_begin": torch.from_numpy(hb), "hist_value": i64(ub & 0xffff), "hist_count": i64(cnt),
    }


def _worker(rank, world, port, genome, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from intronprospector_amd import synth
        from intronprospector_amd.distributed import merge_shard_tables
        import tempfile
        params, fasta = {}, None
        if genome:
            batch, names, gen = synth.short_reads(30_000, seed=21, n_targets=3, target_len=2_800_000, n_sites=400, genome=True)
            d = tempfile.mkdtemp()
            fasta = os.path.join(d, "g%d.fa" % rank)
            gen.write_fasta(fasta)
            params = {"junction_filter": 2}
        else:
            batch, names, _ = synth.short_reads(30_000, seed=22, n_targets=3, target_len=2_800_000, n_sites=400)
        n = len(batch["tid"])
        cuts = np.linspace(0, n, world + 1).astype(int)
        shard = synth.slice_batch(batch, int(cuts[rank]), int(cuts[rank + 1]))
        fc = _to_torch(_oracle_table(shard, names, params, fasta), rank)
        p_all = dict(params)
        if genome:
            p_all["junction_filter"] = 1
        al = _to_torch(_oracle_table(shard, names, p_all, fasta), rank) if genome else fc
        merged = merge_shard_tables(fc, al, rank << 32)
        if rank == 0:
            ref = _oracle_table(batch, names, params, fasta)
            oo = np.lexsort((ref["intron_end"], ref["intron_start"], ref["target"]))
            ok = True
            msg = []
            for k in ("target", "intron_start", "intron_end", "anchor_start", "anchor_end", "strand", "ijunc"):
                a = merged[k].numpy().astype(np.int64)
                b = ref[k][oo].astype(np.int64)
                if a.shape != b.shape or not np.array_equal(a, b):
                    ok = False
                    msg.append(k)
            if ok and not np.array_equal(merged["counts"].numpy().astype(np.uint64), ref["counts"][oo]):
                ok = False
                msg.append("counts")
            if ok:
                # offsets multisets
                starts = np.concatenate([[0], np.cumsum(ref["n_offsets"].astype(np.int64))])
                flat = np.concatenate([ref["offsets"][starts[i]:starts[i + 1]] for i in oo]) if len(oo) else np.zeros(0)
                got = np.repeat(merged["hist_value"].numpy(), merged["hist_count"].numpy())
                if not np.array_equal(flat.astype(np.int64), got.astype(np.int64)):
                    ok = False
                    msg.append("offsets")
                hb = merged["hist_begin"].numpy()
                csum = np.concatenate([[0], np.cumsum(merged["hist_count"].numpy())])
                if not np.array_equal(csum[hb[1:]] - csum[hb[:-1]], ref["n_offsets"][oo].astype(np.int64)):
                    ok = False
                    msg.append("hist_begin")
            q.put((ok, msg, len(oo)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("genome", [False, True])
def test_shard_merge_equals_single_run(world, genome, oracle_lib):
    from tests.mp_util import run_workers
    ok, msg, n = run_workers(_worker, world, (_free_port(), genome), timeout=300)
    assert ok, "merged table differs from the single run in: %s" % msg
    assert n > 100


def _error_worker(rank, world, port, case, q):
    """Every rank must raise the SAME error -- the earliest in input order -- and nobody may hang."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import ctypes as C
        from intronprospector_amd import _lib as L
        from intronprospector_amd.distributed import ShardError, merge_shard_tables, empty_table, agree_on_error
        from intronprospector_amd.extractor import IpxError

        def err(status, kind, read_index, ordinal, errk, detail, text):
            e = L.Error()
            e.status, e.kind, e.read_index = status, kind, read_index
            e.detail = text.encode()
            e.order = (read_index << 30) | (ordinal << 10) | (errk << 8) | detail
            return ShardError(e.order if status == L.IPX_EDATA else 0,
                              bytes((C.c_char * C.sizeof(e)).from_buffer_copy(e)), status)
        mine = None
        if case == "two":            # a 'P' op on the last rank's shard, a FUNMAP spliced read on rank 0's
            if rank == world - 1:
                mine = err(L.IPX_EDATA, L.IPX_DATA_CIGAR, 500 + rank, 0, 0, ord("P"), "Unknown cigar operation 'P'")
            if rank == 0:
                mine = err(L.IPX_EDATA, L.IPX_DATA_CATEGORY, 7, 1, 2, 0, "should never look at FQCAIL, FUNMAP, FDUP reads")
        elif case == "one":          # only the last rank has something to report
            if rank == world - 1:
                mine = err(L.IPX_EDATA, L.IPX_DATA_CIGAR, 500 + rank, 0, 0, ord("B"), "Unknown cigar operation 'B'")
        elif case == "hard":         # a device failure on rank 1 beats any data error
            if rank == 1:
                mine = err(L.IPX_EDEVICE, 0, 0, 0, 0, 0, "hipErrorLaunchFailure")
            if rank == 0:
                mine = err(L.IPX_EDATA, L.IPX_DATA_CIGAR, 3, 0, 0, ord("P"), "Unknown cigar operation 'P'")
        got = None
        try:
            if case == "none":
                agree_on_error(None)
            else:
                merge_shard_tables(empty_table("cpu"), empty_table("cpu"), rank << 32, local_error=mine or ShardError())
        except IpxError as e:
            got = (str(e), int(e.read_index), int(e.status), int(e.kind))
        everybody = [None] * world
        dist.all_gather_object(everybody, got)
        if rank == 0:
            q.put(everybody)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,case", [(2, "two"), (3, "two"), (2, "one"), (2, "hard"), (2, "none")])
def test_shard_errors_all_ranks_raise_the_earliest(world, case):
    """junctions_extractor.cc:73-74,312-313: the reference throws at the FIRST bad alignment of the input;
    sharded, every rank must leave with that one (a rank that raised alone left its peers in
    the next collective for good).  The bounded timeout of run_workers is the hang detector."""
    from tests.mp_util import run_workers
    from intronprospector_amd import _lib as L
    got = run_workers(_error_worker, world, (_free_port(), case), timeout=120)
    assert len(got) == world and all(g == got[0] for g in got), got
    if case == "two":
        assert got[0] == ("should never look at FQCAIL, FUNMAP, FDUP reads", 7, L.IPX_EDATA, L.IPX_DATA_CATEGORY)
    elif case == "one":
        assert got[0] == ("Unknown cigar operation 'B'", 500 + world - 1, L.IPX_EDATA, L.IPX_DATA_CIGAR)
    elif case == "hard":
        assert got[0][0] == "hipErrorLaunchFailure" and got[0][2] == L.IPX_EDEVICE
    else:
        assert got[0] is None


def test_shard_merge_world_8(oracle_lib):
    """the merge protocol with as many ranks as an MI355X node has GPUs"""
    from tests.mp_util import run_workers
    ok, msg, n = run_workers(_worker, 8, (_free_port(), False), timeout=600)
    assert ok, "merged table differs from the single run in: %s" % msg
    assert n > 100


def _config5_worker(rank, world, port, q):
    """BASELINE config 5's harness on CPU: the shares of the stream, and the cross-rank checks of the bench line over
    key-range halves of an oracle table (what the exchange path leaves on two ranks)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from intronprospector_amd import config5, synth
        chunk = 4000
        pieces = config5.rank_pieces(rank, chunk)
        batches = [config5.make_piece(c, lo, hi, chunk) for c, lo, hi in pieces]
        names = batches[0][1]
        n_local = sum(len(b["tid"]) for b, _ in batches)
        assert n_local == config5.share_alignments(chunk) == 50_000 and config5.rank_base(rank, chunk) == rank * 50_000
        # every rank can build the whole small stream (here: both shares) and its oracle table
        whole = []
        for r in range(world):
            whole += [config5.make_piece(c, lo, hi, chunk)[0] for c, lo, hi in config5.rank_pieces(r, chunk)]
        ref = _oracle_table(synth.concat_batches(whole), names, {}, None)
        oo = np.lexsort((ref["intron_end"], ref["intron_start"], ref["target"]))
        J = len(oo)
        cut = [0, J // 2, J][rank:rank + 2] if world == 2 else [J * rank // world, J * (rank + 1) // world]
        rows = oo[cut[0]:cut[1]]
        i64 = lambda a: torch.from_numpy(np.ascontiguousarray(a).astype(np.int64))
        starts = np.concatenate([[0], np.cumsum(ref["n_offsets"].astype(np.int64))])
        hv, hc = [], []
        for i in rows:
            v, c = np.unique(ref["offsets"][starts[i]:starts[i + 1]], return_counts=True)
            hv.append(v); hc.append(c)
        t = {k: i64(ref[k][rows]) for k in ("target", "intron_start", "intron_end", "anchor_start", "anchor_end", "strand")}
        t["counts"] = i64(ref["counts"][rows])
        t["hist_value"] = i64(np.concatenate(hv) if hv else np.zeros(0))
        t["hist_count"] = i64(np.concatenate(hc) if hc else np.zeros(0))
        ijunc = i64(ref["ijunc"][rows])
        n_raw = int(ref["counts"].sum()) if rank == 0 else 0         # (the scans' record count, wherever it was made)
        good = config5.cross_rank_checks(t, ijunc, n_raw)
        # ... and what they say when something is wrong: a junction numbered twice, a range out of place, a read lost
        bad_ij = ijunc.clone()
        if rank == 0 and len(bad_ij) > 1:
            bad_ij[0] = bad_ij[1]
        twice = config5.cross_rank_checks(t, bad_ij, n_raw)
        lost = config5.cross_rank_checks(t, ijunc, n_raw + (1 if rank == 0 else 0))
        t2 = dict(t)
        if rank == 1 and len(rows):
            t2["target"] = t["target"].clone()
            t2["target"][0] = 0
            t2["intron_start"] = t["intron_start"].clone()
            t2["intron_start"][0] = 0
        moved = config5.cross_rank_checks(t2, ijunc, n_raw)
        if rank == 0:
            q.put((good, twice["ijunc_is_a_permutation"], lost["conservation"], moved["key_ranges_in_order"], J,
                   [p for p in pieces], config5.rank_pieces(1, chunk)))
    finally:
        dist.destroy_process_group()


def test_config5_shares_and_cross_rank_checks(oracle_lib):
    from tests.mp_util import run_workers
    good, twice_ok, lost_ok, moved_ok, J, p0, p1 = run_workers(_config5_worker, 2, (_free_port(),), timeout=240)
    assert good["ijunc_is_a_permutation"] and good["conservation"] and good["key_ranges_in_order"]
    assert good["junctions"] == J and len(good["range_digests"]) == 2 and good["range_digests"][0] != good["range_digests"][1]
    assert not twice_ok and not lost_ok and not moved_ok
    # rank 0: chunks 0..11 whole and the first half of 12; rank 1: the second half of 12, then 13..24
    assert p0[0] == (0, 0, 4000) and p0[-1] == (12, 0, 2000) and len(p0) == 13
    assert p1[0] == (12, 2000, 4000) and p1[-1] == (24, 0, 4000) and len(p1) == 13


def test_tail_splits_of_an_overflowed_slotted_exchange():
    """RecordExchanger._exchange_tails derives the split sizes of its second all_to_all from the counts matrix that the
    slot headers carry: every rank must send exactly what its peer expects, and nothing where the slot had room."""
    import torch
    from intronprospector_amd.distributed import tail_splits
    counts = torch.tensor([[10, 0, 700], [512, 513, 5], [0, 2000, 511]], dtype=torch.int64)      # [sender][part]
    room = 512
    sends, recvs = zip(*(tail_splits(counts, room, r) for r in range(3)))
    assert sends == ([0, 0, 188], [0, 1, 0], [0, 1488, 0])
    for s in range(3):
        for p in range(3):
            assert sends[s][p] == recvs[p][s]                   # what s sends to p is what p expects from s
    assert sum(map(sum, sends)) == 188 + 1 + 1488
    assert tail_splits(torch.zeros((2, 2), dtype=torch.int64), 64, 1) == ([0, 0], [0, 0])

