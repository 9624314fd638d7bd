"""GPU: multi-GPU record exchange (key-range all_to_all) on two ranks sharing the one test
GPU, collectives over gloo; the concatenation of the ranks' tables must equal a single
oracle run.  The RCCL path differs only in the backend and in skipping the CPU staging."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, genome, reuse, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import tempfile
        import intronprospector_amd as ipa
        from intronprospector_amd import synth
        from intronprospector_amd.distributed import RecordExchanger, table_to_torch, _all_gather_var
        from tests.oracle_binding import Oracle
        params, fasta, gen2 = {}, None, None
        if genome:
            batch, names, gen = synth.short_reads(60_000, seed=41, n_targets=4, target_len=2_800_000, n_sites=700, genome=True)
            fasta = os.path.join(tempfile.mkdtemp(), "g.fa")
            gen.write_fasta(fasta)
            params = {"junction_filter": 2}
            gen2 = (gen.names, [(t, m, len(s)) for (t, m), s in zip(gen.twobit(), gen.seqs)])
        else:
            batch, names, _ = synth.short_reads(60_000, seed=42, n_targets=4, target_len=2_800_000, n_sites=700)
        ex = ipa.JunctionsExtractor(targets=names, genome=gen2, **params)
        xch = RecordExchanger(torch.device("cuda", 0), comm_device=torch.device("cpu"))

        def one_exchange(batch):
            n = len(batch["tid"])
            cuts = np.linspace(0, n, world + 1).astype(int)
            lo, hi = int(cuts[rank]), int(cuts[rank + 1])
            # ONE scan, one run: a batch that outgrows the slots sized from the one before is completed by a second
            # all_to_all of the tails (round 4), never by scanning again
            ex.clear()
            ex.push(synth.slice_batch(batch, lo, hi), base_read_index=lo)
            table, ijunc = xch.run(ex, hi - lo)
            t = table_to_torch(ex, table, "cpu")
            t["ijunc"] = ijunc.cpu()
            # gather every rank's rows (rank order = key order)
            gathered = {k: torch.cat(_all_gather_var(t[k])) for k in
                        ("target", "intron_start", "intron_end", "anchor_start", "anchor_end", "strand", "ijunc", "counts",
                         "hist_value", "hist_count")}
            nper = _all_gather_var(torch.tensor([t["target"].shape[0]]))
            if rank != 0:
                return None
            ora = Oracle(targets=names, genome_fasta=fasta, **params)
            ora.process(batch)
            ref = ora.table()
            ora.close()
            oo = np.lexsort((ref["intron_end"], ref["intron_start"], ref["target"]))
            bad = []
            for k in ("target", "intron_start", "intron_end", "anchor_start", "anchor_end", "strand", "ijunc"):
                a, b = gathered[k].numpy().astype(np.int64), ref[k][oo].astype(np.int64)
                if a.shape != b.shape or not np.array_equal(a, b):
                    bad.append(k)
            if not bad and not np.array_equal(gathered["counts"].numpy().astype(np.uint64), ref["counts"][oo]):
                bad.append("counts")
            if not bad:
                starts = np.concatenate([[0], np.cumsum(ref["n_offsets"].astype(np.int64))])
                flat = np.concatenate([ref["offsets"][starts[i]:starts[i + 1]] for i in oo])
                got = np.repeat(gathered["hist_value"].numpy(), gathered["hist_count"].numpy())
                if not np.array_equal(flat.astype(np.int64), got.astype(np.int64)):
                    bad.append("offsets")
            return bad, len(oo), [int(x.item()) for x in nper]

        out = one_exchange(batch)
        if reuse:
            # a second batch of the same kind keeps the splitters; one that sits on a single target upsets the
            # balance (still exact), so the run after it draws new ones
            drawn = [xch.resampled]
            b2, _, _ = synth.short_reads(50_000, seed=43, n_targets=4, target_len=2_800_000, n_sites=700)
            r2 = one_exchange(b2)
            drawn.append(xch.resampled)
            keep = np.nonzero(b2["tid"] == 0)[0]
            b3 = synth.concat_batches([synth.slice_batch(b2, int(i), int(i) + 1) for i in keep[:4000]])
            r3 = one_exchange(b3)
            drawn.append(xch.resampled)
            r4 = one_exchange(b3)
            drawn.append(xch.resampled)
            # ... and a batch far larger than the one before outgrows the slots sized from that: the tails follow in a second
            # all_to_all (counted in slot_overflows), the slots are larger afterwards
            r5 = one_exchange(b2)                 # (the skewed batch left no splitters: drawn afresh, exact exchange)
            r6 = one_exchange(b2)                 # in slots sized from b2
            over0 = xch.slot_overflows
            b7, _, _ = synth.short_reads(600_000, seed=44, n_targets=4, target_len=2_800_000, n_sites=700)
            r7 = one_exchange(b7)
            over1 = xch.slot_overflows
            r8 = one_exchange(b7)                 # (fits now)
            if rank == 0:
                out = (out[0] + r2[0] + r3[0] + r4[0] + r5[0] + r6[0] + r7[0] + r8[0], out[1], out[2], drawn, r4[2],
                       (over1 - over0, xch.slot_overflows - over1))
        ex.close()
        if rank == 0:
            q.put(out)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("genome", [False, True])
def test_record_exchange_two_ranks_equals_single_run(genome, oracle_lib):
    from tests.mp_util import run_workers
    bad, n, nper = run_workers(_worker, 2, (_free_port(), genome, False), timeout=300)
    assert not bad, "exchanged tables differ from the single run in: %s" % bad
    assert n > 100 and min(nper) > 0, nper       # both ranks own part of the key space


def test_record_exchange_keeps_its_splitters_while_balanced(oracle_lib):
    from tests.mp_util import run_workers
    bad, n, nper, drawn, nper_last, overflows = run_workers(_worker, 2, (_free_port(), False, True), timeout=300)
    assert not bad, "exchanged tables differ from the single run in: %s" % bad
    # drawn once for the first batch, kept for the second and (still) for the skewed third, drawn again for the fourth
    # (the second and third batch travel in slots: the imbalance is read off the slot headers, not off a counts collective)
    assert drawn == [1, 1, 1, 2], drawn
    assert min(nper_last) > 0, nper_last
    assert overflows == (1, 0), overflows


def _error_worker(rank, world, port, path, q):
    """A 'P' operation in rank 1's shard and a spliced FUNMAP read in rank 0's: both paths, both ranks,
    one error -- the one the reference's loop meets first."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import intronprospector_amd as ipa
        from intronprospector_amd import synth
        from intronprospector_amd.distributed import RecordExchanger, TableMerger
        from intronprospector_amd.extractor import IpxError
        from tests.oracle_binding import Oracle, OracleError
        batch, names, _ = synth.short_reads(40_000, seed=52, n_targets=4, target_len=2_800_000, n_sites=700)
        n = len(batch["tid"])
        cuts = np.linspace(0, n, world + 1).astype(int)
        plain = np.nonzero((batch["n_cigar"] == 3) & (batch["flag"] == 0) & (batch["nh"] == 1))[0]
        i0 = int(plain[(plain > 100) & (plain < cuts[1] - 100)][5])          # rank 0's shard
        i1 = int(plain[plain > cuts[1] + 100][5])                            # rank 1's shard
        batch["flag"][i0] |= 0x4                                             # FUNMAP with a spliced CIGAR: :73-74
        w = int(batch["cigar_off"][i1])
        batch["cigar"][w] = (batch["cigar"][w] & ~np.uint32(0xf)) | np.uint32(6)   # 'P': :312-313
        expect = None
        try:
            ora = Oracle(targets=names)
            ora.process(batch)
        except OracleError as e:
            expect = (e.msg, int(e.index))
        lo, hi = int(cuts[rank]), int(cuts[rank + 1])
        ex = ipa.JunctionsExtractor(targets=names)
        xch = RecordExchanger(torch.device("cuda", 0), comm_device=torch.device("cpu"))
        if path == "exchange-slotted":            # a clean batch first: the faulty one then travels in slots
            clean, _, _ = synth.short_reads(40_000, seed=53, n_targets=4, target_len=2_800_000, n_sites=700)
            ex.push(synth.slice_batch(clean, lo, hi), base_read_index=lo)
            xch.run(ex, hi - lo)
            assert xch._slot_records is not None
            ex.clear()
        ex.push(synth.slice_batch(batch, lo, hi), base_read_index=lo)
        got = None
        try:
            if path.startswith("exchange"):
                xch.run(ex, hi - lo)
            else:
                TableMerger(torch.device("cpu")).merge(ex, lo)
        except IpxError as e:
            got = (str(e), int(e.read_index))
        ex.close()
        everybody = [None] * world
        dist.all_gather_object(everybody, got)
        if rank == 0:
            q.put((everybody, expect, i0, i1))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("path", ["exchange", "exchange-slotted", "tables"])
def test_two_ranks_raise_the_earliest_error_of_the_run(path, oracle_lib):
    from tests.mp_util import run_workers
    everybody, expect, i0, i1 = run_workers(_error_worker, 2, (_free_port(), path), timeout=240)
    assert expect is not None and expect[1] == i0 < i1, (expect, i0, i1)
    assert everybody[0] == everybody[1] == expect, (everybody, expect)


def test_rccl_exchange_with_a_different_batch_every_step():
    """The slotted exchange on RCCL (one rank: the all_to_all is a device copy ordered by streams alone, no host in
    between) with batches that DIFFER from step to step, each table against the oracle.  Round 3's rehearsals resent one
    batch, so a step that adopted the previous step's slots, or sent slots still being written, produced the right table
    all the same; and the exchange ran beside the stream it was asked to run on (advisor, round 3)."""
    import intronprospector_amd as ipa
    from intronprospector_amd import synth
    from intronprospector_amd.distributed import RecordExchanger, SlotOverflow, table_to_torch
    from tests.oracle_binding import Oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        dev = torch.device("cuda", 0)
        names = None
        ex = None
        xch = RecordExchanger(dev, comm_device=dev)
        slotted = 0
        for step, (reads, seed) in enumerate([(300_000, 71), (280_000, 72), (310_000, 73), (150_000, 74), (305_000, 75)]):
            batch, nm, _ = synth.short_reads(reads, seed=seed, n_targets=4, target_len=9_000_000, n_sites=6000)
            if ex is None:
                names = nm
                ex = ipa.JunctionsExtractor(targets=names)
            ex.clear()
            ex.push(batch)
            before = xch.slot_overflows
            was_slotted = xch._splitters is not None and xch._slot_records is not None
            if step == 2:
                xch._slot_records = 40            # far too small: nearly every record travels in the second all_to_all
            table, ijunc = xch.run(ex, reads)
            slotted += int(was_slotted)
            if step == 2:
                assert xch.slot_overflows == before + 1 and xch._slot_records > 1000
            else:
                assert xch.slot_overflows == before         # (slots shrink by a tenth a step at most: the others fit)
            t = table_to_torch(ex, table, "cpu")
            ora = Oracle(targets=names)
            ora.process(batch)
            ref = ora.table()
            ora.close()
            oo = np.lexsort((ref["intron_end"], ref["intron_start"], ref["target"]))
            for k in ("target", "intron_start", "intron_end", "anchor_start", "anchor_end", "strand"):
                assert np.array_equal(t[k].numpy().astype(np.int64), ref[k][oo].astype(np.int64)), (step, k)
            assert np.array_equal(ijunc.cpu().numpy().astype(np.int64), ref["ijunc"][oo].astype(np.int64)), (step, "ijunc")
            assert np.array_equal(t["counts"].numpy().astype(np.uint64), ref["counts"][oo]), (step, "counts")
            starts = np.concatenate([[0], np.cumsum(ref["n_offsets"].astype(np.int64))])
            flat = np.concatenate([ref["offsets"][starts[i]:starts[i + 1]] for i in oo])
            got = np.repeat(t["hist_value"].numpy(), t["hist_count"].numpy())
            assert np.array_equal(flat.astype(np.int64), got.astype(np.int64)), (step, "offsets")
        assert slotted >= 3          # (the first step draws splitters: exact exchange; the others travel in slots)
        ex.close()
    finally:
        dist.destroy_process_group()


def test_bench_config5_line_with_two_ranks_on_one_gpu(tmp_path):
    """bench.py's N > 1 line (BASELINE config 5: every rank its 12.5 chunks in one table, record exchange, finalize, global
    sjN) on two ranks that share the GPU, collectives over gloo, a stream of small chunks: the line must carry the
    cross-rank checks and an oracle parity record, all good."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--backend", "gloo", "--chunk", "60000", "--no-configs"]
    r = subprocess.run(cmd, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    line = [l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["alignments_per_gpu"] == 750_000
    assert d["config"]["workload"].startswith("M: BASELINE config 5")
    ck = d["config"]["checks"]
    assert ck["conservation"] and ck["ijunc_is_a_permutation"] and ck["key_ranges_in_order"] and len(ck["range_digests"]) == 2
    par = d["config"]["parity"]
    assert par["equals_oracle"] and par["alignments"] == 2 * 25 * (par["chunk_alignments"] // 2) and par["junctions"] > 1000
    assert par["checks"]["conservation"] and par["checks"]["ijunc_is_a_permutation"]
    assert d["value"] > 0 and d["roofline"]["frac"] > 0
