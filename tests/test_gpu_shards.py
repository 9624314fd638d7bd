"""GPU: ipx_finalize_shard on two ranks sharing the one test GPU, tables merged over
gloo, against a single oracle run (the RCCL path differs only in the backend)."""
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


def _worker(rank, world, port, genome, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import ctypes as C
        import tempfile
        import intronprospector_amd as ipa
        from intronprospector_amd import synth, _lib as L
        from intronprospector_amd.distributed import merge_shard_tables, table_to_torch
        from tests.oracle_binding import Oracle
        params, fasta, gen2 = {}, None, None
        if genome:
            batch, names, gen = synth.short_reads(40_000, seed=31, n_targets=3, target_len=2_800_000, n_sites=500, genome=True)
            fasta = os.path.join(tempfile.mkdtemp(), "g.fa")
            gen.write_fasta(fasta)
            params = {"junction_filter": 2}
            gen2 = (gen.names, [(t, m, len(s)) for (t, m), s in zip(gen.twobit(), gen.seqs)])
        else:
            batch, names, _ = synth.short_reads(40_000, seed=32, n_targets=3, target_len=2_800_000, n_sites=500)
        n = len(batch["tid"])
        cuts = np.linspace(0, n, world + 1).astype(int)
        lo, hi = int(cuts[rank]), int(cuts[rank + 1])
        ex = ipa.JunctionsExtractor(targets=names, genome=gen2, **params)
        ex.push(synth.slice_batch(batch, lo, hi), base_read_index=lo)
        fc, al = L.Table(), L.Table()
        rc = ex.lib.ipx_finalize_shard(ex.h, L.IPX_MEM_DEVICE, C.byref(fc), C.byref(al))
        assert rc == 0
        t_fc = table_to_torch(ex, fc, "cpu")
        t_al = table_to_torch(ex, al, "cpu")
        merged = merge_shard_tables(t_fc, t_al, lo << 20)
        ex.close()
        if rank == 0:
            ora = Oracle(targets=names, genome_fasta=fasta, **params)
            ora.process(batch)
            ref = ora.table()
            ora.close()
            oo = np.lexsort((ref["intron_end"], ref["intron_start"], ref["target"]))
            bad = []
            for k in ("target", "intron_start", "intron_end", "anchor_start", "anchor_end", "strand", "ijunc"):
                a, b = merged[k].numpy().astype(np.int64), ref[k][oo].astype(np.int64)
                if a.shape != b.shape or not np.array_equal(a, b):
                    bad.append(k)
            if not bad and not np.array_equal(merged["counts"].numpy().astype(np.uint64), ref["counts"][oo]):
                bad.append("counts")
            if not bad:
                starts = np.concatenate([[0], np.cumsum(ref["n_offsets"].astype(np.int64))])
                flat = np.concatenate([ref["offsets"][starts[i]:starts[i + 1]] for i in oo])
                got = np.repeat(merged["hist_value"].numpy(), merged["hist_count"].numpy())
                if not np.array_equal(flat.astype(np.int64), got.astype(np.int64)):
                    bad.append("offsets")
            q.put((bad, len(oo)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("genome", [False, True])
def test_two_shards_on_gpu_merge_to_single_run(genome, oracle_lib):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, genome, q)) for r in range(2)]
    for p in procs:
        p.start()
    bad, n = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert not bad, "merged shard tables differ from the single run in: %s" % bad
    assert n > 100
