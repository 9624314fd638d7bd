"""GPU: BASELINE.json's configurations at full size.  The oracle is fast enough (a scalar C walk
with a hash table, tens of millions of alignments per second) to check the whole table bit for bit
at 10 M alignments; on top of that the size-independent properties of the domain: key order,
conservation of records, histogram sums, ijunc as the rank of the creation order, shard invariance
(one push == several pushes == two handles merged by hand) and idempotence of finalize."""
import hashlib

import numpy as np
import pytest

from tests import helpers
from tests.oracle_binding import Oracle

pytestmark = pytest.mark.gpu


def _digest(t):
    h = hashlib.sha256()
    for k in ("target", "intron_start", "intron_end", "anchor_start", "anchor_end", "strand", "motif", "counts",
              "creator_order", "ijunc", "hist_begin", "hist_value", "hist_count"):
        h.update(np.ascontiguousarray(t[k]).tobytes())
    return h.hexdigest()


def _check_properties(t, n_alignments):
    n = len(t["intron_start"])
    assert t["n_alignments"] == n_alignments and n > 0
    # std::map order: (target, start, end) strictly ascending
    key = (t["target"].astype(np.uint64) << np.uint64(40)) | t["intron_start"].astype(np.uint64)
    assert np.all((key[1:] > key[:-1]) | ((key[1:] == key[:-1]) & (t["intron_end"][1:] > t["intron_end"][:-1])))
    # without a filter every QC-passed record is counted exactly once
    assert int(t["counts"].sum()) == int(t["n_records"])
    # histogram: CSR, counts sum to the junction's reads, offsets ascending inside a junction
    hb = t["hist_begin"].astype(np.int64)
    assert hb[0] == 0 and hb[-1] == len(t["hist_value"]) == len(t["hist_count"])
    assert np.all(hb[1:] > hb[:-1])
    per_j = np.add.reduceat(t["hist_count"].astype(np.int64), hb[:-1])
    assert np.array_equal(per_j, t["counts"].sum(axis=1).astype(np.int64))
    inner = np.ones(len(t["hist_value"]), bool)
    inner[hb[:-1]] = False
    v = t["hist_value"].astype(np.int64)
    assert np.all(v[1:][inner[1:]] > v[:-1][inner[1:]])
    # ijunc: a permutation, and the rank of the creation order
    assert np.array_equal(np.sort(t["ijunc"]), np.arange(n, dtype=t["ijunc"].dtype))
    assert np.array_equal(np.argsort(t["creator_order"], kind="stable"), np.argsort(t["ijunc"], kind="stable"))
    assert np.all(t["anchor_start"] < t["intron_start"]) and np.all(t["anchor_end"] > t["intron_end"])


def _run(batch, names, n_push=1, **params):
    import intronprospector_amd as ipa
    from intronprospector_amd import synth
    ex = ipa.JunctionsExtractor(targets=names, **params)
    n = len(batch["tid"])
    cuts = np.linspace(0, n, n_push + 1).astype(int)
    for a, b in zip(cuts[:-1], cuts[1:]):
        ex.push(synth.slice_batch(batch, int(a), int(b)))
    t = ex.get_junctions()
    t2 = ex.get_junctions()                       # idempotence
    ex.close()
    assert _digest(t) == _digest(t2)
    return t


def test_config_S_10M_short_reads():
    from intronprospector_amd import synth
    batch, names, _ = synth.short_reads(10_000_000, seed=synth.SEED_S)
    t1 = _run(batch, names)
    _check_properties(t1, 10_000_000)
    assert _digest(_run(batch, names, n_push=3)) == _digest(t1)          # shard invariance
    ora = Oracle(targets=names)
    ora.process(batch)
    helpers.compare_tables(t1, ora.table())
    ora.close()


def test_config_S_10M_short_reads_in_random_order(monkeypatch, capfd):
    """The reference's --unsorted mode (junctions_extractor.cc:678-683) at BASELINE's size: the 10 M alignments of S in
    random order.  Nothing folds, 9.87 M records reach the sort -- the bucket sort's by default (8 M records on) --, and
    the 1 900 (junction, offset) keys with 500 to 16 000 copies each are fenced into leaves of their own; with fewer
    coarse ranges their fences outgrew a range's splitter table in one sort of eight and the handle fell back to the
    radix passes (profiles/experiments/bsort_splitter_model.py).  The whole table against the oracle's, five finalizes
    (every one draws another sample: the records' order in memory differs from push to push), none of them a fallback."""
    import intronprospector_amd as ipa
    from intronprospector_amd import synth
    monkeypatch.setenv("IPX_DEBUG_PLAN", "1")
    batch, names, _ = synth.short_reads(10_000_000, seed=synth.SEED_S)
    shuffled = synth.permute_batch(batch, np.random.default_rng(11).permutation(len(batch["tid"])))
    del batch
    ora = Oracle(targets=names)
    ora.process(shuffled)
    tab = ora.table()
    ora.close()
    ex = ipa.JunctionsExtractor(targets=names)
    digests = set()
    for _ in range(5):
        ex.clear()
        ex.push(shuffled, base_read_index=0)
        t = ex.get_junctions()
        digests.add(_digest(t))
    ex.close()
    _check_properties(t, 10_000_000)
    assert int(t["n_records"]) > 9_000_000
    helpers.compare_tables(t, tab)
    assert len(digests) == 1
    err = capfd.readouterr().err
    assert "bucket sort overflowed" not in err, [l for l in err.splitlines() if "overflowed" in l]


def test_config_L_long_reads():
    from intronprospector_amd import synth
    batch, names, _ = synth.long_reads(200_000, seed=synth.SEED_L)
    p = {"max_anchor_indel_size": 36}
    t1 = _run(batch, names, **p)
    _check_properties(t1, 200_000)
    assert _digest(_run(batch, names, n_push=4, **p)) == _digest(t1)
    ora = Oracle(targets=names, **p)
    ora.process(batch)
    helpers.compare_tables(t1, ora.table())
    ora.close()


def test_config_L_2M_long_reads():
    """BASELINE.json's long-read configuration at its full size: 2 M ONT-style alignments, 1.65 G CIGAR operations,
    40 M junction records, the whole table against the oracle (a minute of generating, seconds of oracle)"""
    from intronprospector_amd import synth
    batch, names, _ = synth.long_reads(2_000_000, seed=synth.SEED_L)
    p = {"max_anchor_indel_size": 36}
    t1 = _run(batch, names, n_push=2, **p)
    _check_properties(t1, 2_000_000)
    assert len(t1["intron_start"]) > 400_000 and int(t1["n_records"]) > 40_000_000
    ora = Oracle(targets=names, **p)
    ora.process(batch)
    helpers.compare_tables(t1, ora.table())
    ora.close()


def test_config_G_genome_strand_inference(tmp_path):
    """10 M short reads over a 160 Mb synthetic genome (2-bit on the device), --sj-filter=canon."""
    from intronprospector_amd import synth, _lib as L
    batch, names, gen = synth.short_reads(10_000_000, seed=synth.SEED_S + 1, target_len=20_000_000, n_sites=100_000,
                                          genome=True)
    fa = str(tmp_path / "g.fa")
    gen.write_fasta(fa)
    g2 = (gen.names, [(t, m, len(s)) for (t, m), s in zip(gen.twobit(), gen.seqs)])
    t1 = _run(batch, names, genome=g2, junction_filter=L.IPX_SJ_CANON)
    assert "error" not in t1
    n = len(t1["intron_start"])
    assert n > 50_000 and int(t1["counts"].sum()) <= int(t1["n_records"])      # the filter drops records
    ora = Oracle(targets=names, genome_fasta=fa, junction_filter=2)
    ora.process(batch)

    def fetch(tid, start, end):            # the four raw bases, as tests/test_gpu_parity.py::_fetcher
        s = gen.seqs[tid]
        return bytes([s[start], s[start + 1], s[end - 2], s[end - 1]])
    helpers.compare_tables(t1, ora.table(), fetch)
    ora.close()


def test_config_M_one_rank_share_125M_alignments():
    """BASELINE.json's config 5 (1 G short reads over 8 GPUs) as ONE rank sees it: 125 M alignments -- chunks 0..12 of
    the M workload (seeds SEED_M0 + i, 10 M alignments each, the thirteenth half) -- pushed chunk by chunk into one
    table (whole-run scope, thirteen pushes, 26 M combined records, the bucket sort's scale) and compared with the
    oracle's table of the same stream, junction by junction.  Chunks are generated, pushed and dropped one at a time."""
    import intronprospector_amd as ipa
    from intronprospector_amd import synth
    ex = ora = None
    total = 0
    for i in range(13):
        n = 10_000_000 if i < 12 else 5_000_000
        batch, names, _ = synth.short_reads(n, seed=synth.SEED_M0 + i)
        if ex is None:
            ex = ipa.JunctionsExtractor(targets=names)
            ora = Oracle(targets=names)
        ex.push(batch)
        ora.process(batch)
        total += n
        del batch
    dev = ex.get_junctions()
    ex.close()
    tab = ora.table()
    ora.close()
    assert dev["n_alignments"] == total == 125_000_000
    assert len(dev["intron_start"]) > 3_000_000 and dev["n_records"] > 100_000_000
    helpers.compare_tables(dev, tab)
    _check_properties(dev, total)


def test_same_input_gives_the_same_table_every_time():
    """Thirty million alignments in three pushes, forty times over: every table must be the same, bit for bit.  A race
    inside a kernel shows here long before a parity test that runs once meets it -- the divergent barrier in
    k_scan_short's early combine (fixed in round 3; profiles/consistency.py) garbled about one table in fifteen at this
    size and had passed every other test for a round and a half.  The first table is also the oracle's."""
    import torch
    import intronprospector_amd as ipa
    from intronprospector_amd import synth
    from bench import upload
    dev = torch.device("cuda", 0)
    parts = [synth.short_reads(10_000_000, seed=synth.SEED_M0 + 40 + i) for i in range(3)]
    names = parts[0][1]
    ora = Oracle(targets=names)
    for p in parts:
        ora.process(p[0])
    ref = ora.table()
    ora.close()
    ups = [upload(p[0], dev) for p in parts]
    sizes = [(len(p[0]["tid"]), len(p[0]["cigar"])) for p in parts]
    del parts
    ex = ipa.JunctionsExtractor(targets=names)
    seen = {}
    for r in range(40):
        ex.clear()
        base = 0
        for (na, nw), (_, ptrs) in zip(sizes, ups):
            ex.push_device(na, nw, ptrs, base_read_index=base)
            base += na
        t = ex.get_junctions()
        if r == 0:
            helpers.compare_tables(t, ref)
        d = _digest(t)
        seen[d] = seen.get(d, 0) + 1
    ex.close()
    assert len(seen) == 1, "tables of the same input differ from run to run: %s" % sorted(seen.values(), reverse=True)
