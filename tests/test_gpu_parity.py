"""GPU parity: the HIP path through the C ABI (libipx.so) against the CPU oracle,
bit for bit, on seeded synthetic batches and on the reference's own inputs."""
import os

import numpy as np
import pytest
import torch            # before libipx.so: the HIP runtime torch brings along must be the one the process keeps

from tests import helpers
from tests.oracle_binding import Oracle, OracleError

pytestmark = pytest.mark.gpu


def _extractor(**kw):
    import intronprospector_amd as ipa
    return ipa.JunctionsExtractor(**kw)


def _run_both(batch, names, params=None, genome=None, fasta=None, scope=0, n_push=1):
    """-> (device table, oracle table). params: dict with oracle-style keys."""
    from intronprospector_amd import synth
    params = dict(params or {})
    gen2 = None
    if genome is not None:
        gen2 = (genome.names, [(t, m, len(s)) for (t, m), s in zip(genome.twobit(), genome.seqs)])
    ex = _extractor(targets=names, scope=scope, genome=gen2, **params)
    n = len(batch["tid"])
    cuts = np.linspace(0, n, n_push + 1).astype(int)
    for a, b in zip(cuts[:-1], cuts[1:]):
        ex.push(synth.slice_batch(batch, int(a), int(b)))
    dev = ex.get_junctions()
    ex.close()
    ora = Oracle(targets=names, genome_fasta=fasta, **params)
    err = None
    try:
        ora.process(batch)
    except OracleError as e:
        err = e
    tab = ora.table()
    ora.close()
    return dev, tab, err


def _fetcher(genome):
    def fetch(tid, start, end):
        s = genome.seqs[tid]
        return bytes([s[start], s[start + 1], s[end - 2], s[end - 1]])
    return fetch


@pytest.mark.parametrize("params", [
    {},
    {"excludes": 1},
    {"strandness": 1},
    {"strandness": 2, "min_anchor_length": 1, "min_intron_length": 1, "max_intron_length": 4000000},
    {"min_anchor_length": 20, "min_intron_length": 200, "max_intron_length": 50000},
])
def test_short_reads_match_oracle(params):
    from intronprospector_amd import synth
    batch, names, _ = synth.short_reads(120_000, seed=11, n_targets=5, target_len=6_000_000, n_sites=4000)
    dev, ora, err = _run_both(batch, names, params)
    assert err is None and "error" not in dev
    assert len(ora["intron_start"]) > 1000
    helpers.compare_tables(dev, ora)


def test_short_reads_multiple_pushes_equal_single():
    from intronprospector_amd import synth
    batch, names, _ = synth.short_reads(60_000, seed=12, n_targets=3, target_len=4_000_000, n_sites=1500)
    dev, ora, err = _run_both(batch, names, {}, n_push=7)
    assert err is None
    helpers.compare_tables(dev, ora)


def test_short_reads_in_random_order_match_oracle():
    """identify_junctions_for_bam (junctions_extractor.cc:678-683) takes unsorted input: creation order, and with
    it sjN, follows the input order, nothing else changes."""
    from intronprospector_amd import synth
    batch, names, _ = synth.short_reads(60_000, seed=33, n_targets=3, target_len=4_000_000, n_sites=1500)
    shuffled = synth.permute_batch(batch, np.random.default_rng(34).permutation(len(batch["tid"])))
    dev, ora, err = _run_both(shuffled, names, {}, n_push=2)
    assert err is None
    helpers.compare_tables(dev, ora)
    sorted_dev, sorted_ora, _ = _run_both(batch, names, {})
    assert np.array_equal(sorted_ora["counts"], ora["counts"]) and not np.array_equal(sorted_ora["ijunc"], ora["ijunc"])


@pytest.mark.parametrize("n_reads", [127, 128, 129, 511, 512, 513, 1023, 1025, 128 * 9 + 1, 512 * 37 + 1, 512 * 40 - 1])
def test_ragged_last_tile_of_the_short_read_scan(n_reads):
    """The short-read scan works in tiles (a wave's 128 alignments, a block's 512; rounds 1-4: 512 per block) whose fields
    are fetched from a CLAMPED index: a batch that ends one alignment before, at, or after a tile boundary must count every read once and none twice (review of round 3, item 1)."""
    from intronprospector_amd import synth
    batch, names, _ = synth.short_reads(n_reads, seed=700 + n_reads % 97, n_targets=2, target_len=2_800_000, n_sites=60)
    dev, ora, err = _run_both(batch, names, {})
    assert err is None
    helpers.compare_tables(dev, ora)


@pytest.mark.parametrize("k_out_of_place,flt", [(1, 0), (17, 0), (400, 0), (17, 2)])
def test_sorted_input_with_a_few_reads_out_of_place(k_out_of_place, flt, tmp_path):
    """Coordinate-sorted input in which k reads sit far from their place (what a merge of sorted files with a few
    stragglers looks like): the tile combiner folds what is adjacent, the stragglers' records meet their junction's other
    records only in the sort -- creators, counts and sjN must be the oracle's, also under the canonical-motif filter, where
    a straggler may be the record that creates a junction (review of round 3, item 1)."""
    from intronprospector_amd import synth
    gen, fa = None, None
    if flt:
        batch, names, gen = synth.short_reads(90_000, seed=811, n_targets=3, target_len=2_800_000, n_sites=900, genome=True)
        fa = str(tmp_path / "g.fa")
        gen.write_fasta(fa)
    else:
        batch, names, _ = synth.short_reads(90_000, seed=810, n_targets=3, target_len=2_800_000, n_sites=900)
    n = len(batch["tid"])
    rng = np.random.default_rng(812 + k_out_of_place)
    perm = np.arange(n)
    src = rng.choice(n, size=k_out_of_place, replace=False)
    dst = rng.choice(n, size=k_out_of_place, replace=False)
    perm[src], perm[dst] = perm[dst].copy(), perm[src].copy()       # k pairs of reads trade places
    dev, ora, err = _run_both(synth.permute_batch(batch, perm), names, {"junction_filter": flt}, genome=gen, fasta=fa, n_push=2)
    assert err is None and "error" not in dev
    helpers.compare_tables(dev, ora, _fetcher(gen) if gen is not None else None)


def test_finalize_then_push_more_then_finalize_again():
    """The table may be asked for repeatedly while records accumulate (multi-input mode of the
    host): a second sort starts from an already permuted record array."""
    from intronprospector_amd import synth
    batch, names, _ = synth.short_reads(50_000, seed=17, n_targets=3, target_len=4_000_000, n_sites=1200)
    ex = _extractor(targets=names)
    half = len(batch["tid"]) // 2
    ex.push(synth.slice_batch(batch, 0, half))
    first = ex.get_junctions()
    again = ex.get_junctions()
    for k in ("intron_start", "intron_end", "anchor_start", "anchor_end", "ijunc", "counts", "hist_count"):
        assert np.array_equal(first[k], again[k])
    ex.push(synth.slice_batch(batch, half, len(batch["tid"])))
    dev = ex.get_junctions()
    ex.close()
    # (one oracle per comparison: like the reference, it caches a junction's confidence and
    # sorted offsets the first time they are asked for)
    ora = Oracle(targets=names)
    ora.process(synth.slice_batch(batch, 0, half))
    helpers.compare_tables(first, ora.table())
    ora.close()
    ora = Oracle(targets=names)
    ora.process(batch)
    helpers.compare_tables(dev, ora.table())
    ora.close()


@pytest.mark.parametrize("flt", [0, 1, 2])
def test_genome_filter_strand_motif(tmp_path, flt):
    """config G: device 2-bit genome, canonical filter, fail-then-pass ordering."""
    from intronprospector_amd import synth
    batch, names, gen = synth.short_reads(80_000, seed=13, n_targets=3, target_len=3_000_000, n_sites=1200, genome=True)
    fa = str(tmp_path / "g.fa")
    gen.write_fasta(fa)
    dev, ora, err = _run_both(batch, names, {"junction_filter": flt}, genome=gen, fasta=fa)
    assert err is None and "error" not in dev
    helpers.compare_tables(dev, ora, _fetcher(gen))
    if flt == 2:
        # the filter must actually have dropped something and flipped some strands
        assert (ora["strand"] == ord("-")).any() and (ora["strand"] == ord("+")).any()


def test_genome_global_scope_name_order(tmp_path):
    """--unsorted / multi-input semantics: one table keyed by chromosome NAME."""
    from intronprospector_amd import synth
    batch, names, gen = synth.short_reads(40_000, seed=14, n_targets=12, target_len=2_700_000, n_sites=900, genome=True)
    # chr1, chr10, chr11, chr12, chr2 ...: name order differs from tid order
    fa = str(tmp_path / "g.fa")
    gen.write_fasta(fa)
    rng = np.random.default_rng(5)
    perm = rng.permutation(len(batch["tid"]))          # unsorted input
    shuffled = synth.concat_batches([synth.slice_batch(batch, int(i), int(i) + 1) for i in perm[:6000]])
    dev, ora, err = _run_both(shuffled, names, {"junction_filter": 2}, genome=gen, fasta=fa, scope=1)
    assert err is None
    helpers.compare_tables(dev, ora, _fetcher(gen))
    # rows come back in std::map order: (name, start, end)
    nm = np.array(names)[dev["target"]]
    key = list(zip(nm.tolist(), dev["intron_start"].tolist(), dev["intron_end"].tolist()))
    assert key == sorted(key)


@pytest.mark.parametrize("max_indel", [0, 3, 36])
def test_long_reads_match_oracle(max_indel):
    from intronprospector_amd import synth
    batch, names, _ = synth.long_reads(1500, seed=15, n_targets=3, target_len=9_000_000, n_transcripts=60)
    dev, ora, err = _run_both(batch, names, {"max_anchor_indel_size": max_indel})
    assert err is None
    if max_indel:
        assert len(ora["intron_start"]) > 100
    helpers.compare_tables(dev, ora)


GOLDEN_INPUTS = ["test_basic.sam", "test_multi.sam", "ont.sam", "ont-skip-targets.sam", "fake10.star.sam",
                 "fake10.hisat2.sam", "sort-bug.sam"]


@pytest.mark.parametrize("sam", GOLDEN_INPUTS)
@pytest.mark.parametrize("indels", [0, 36])
def test_reference_inputs_match_oracle(golden, sam, indels):
    names, batch, _ = helpers.parse_sam(os.path.join(golden, "input", sam))
    dev, ora, err = _run_both(batch, names, {"max_anchor_indel_size": indels})
    assert err is None
    helpers.compare_tables(dev, ora)


def test_reference_fake10_with_genome(golden):
    from intronprospector_amd import synth
    fa = os.path.join(golden, "input", "fake10_scramble_genome.fa")
    seq = np.frombuffer(b"".join(l.strip() for l in open(fa, "rb").readlines()[1:]), np.uint8)
    gen = synth.Genome(["fake10"], [seq])
    names, batch, _ = helpers.parse_sam(os.path.join(golden, "input", "fake10.star.sam"))
    for flt in (1, 2):
        dev, ora, err = _run_both(batch, names, {"junction_filter": flt}, genome=gen, fasta=fa)
        assert err is None
        helpers.compare_tables(dev, ora, _fetcher(gen))


def test_known_answer_fail_then_pass(golden):
    """SURVEY.md 8a-7 recorded reference output, through the device path."""
    from intronprospector_amd import synth
    fa = os.path.join(golden, "input", "fake10_genome.fa")
    seq = np.frombuffer(b"".join(l.strip() for l in open(fa, "rb").readlines()[1:]), np.uint8)
    gen = synth.Genome(["fake10"], [seq])
    names, batch, _ = helpers.parse_sam(os.path.join(golden, "known", "fail_then_pass.sam"))
    dev, ora, _ = _run_both(batch, names, {"junction_filter": 2}, genome=gen, fasta=fa)
    assert dev["strand"].tolist() == [ord("+")] and dev["counts"].tolist() == [[2, 0, 0]]
    assert (dev["intron_start"][0] - dev["anchor_start"][0], dev["anchor_end"][0] - dev["intron_end"][0]) == (68, 43)
    dev, ora, _ = _run_both(batch, names, {"junction_filter": 1}, genome=gen, fasta=fa)
    assert dev["strand"].tolist() == [ord("-")] and dev["counts"].tolist() == [[3, 0, 0]]
    assert helpers.motif_string(int(dev["strand"][0]), int(dev["motif"][0])) == b"ct/ac"


def _one_read(cigar_words, flag=0, pos=1000, tid=0, nh=1):
    return {"tid": np.array([tid], np.int32), "pos": np.array([pos], np.int32),
            "flag": np.array([flag], np.uint16), "xs": np.zeros(1, np.uint8), "nh": np.array([nh], np.uint8),
            "cigar_off": np.zeros(1, np.uint32), "n_cigar": np.array([len(cigar_words)], np.uint32),
            "cigar": np.array(cigar_words, np.uint32)}


def _cig(s):
    out, num = [], ""
    for ch in s:
        if ch.isdigit():
            num += ch
        else:
            out.append((int(num) << 4) | "MIDNSHP=XB".index(ch))
            num = ""
    return out


def test_stale_indel_counter_quirk():
    """SURVEY.md 8a-1 recorded reference outputs."""
    from intronprospector_amd import synth
    p = {"max_anchor_indel_size": 3, "min_intron_length": 10}
    for cig, exp in (("20M5D9M100N20M", []), ("9M100N20M", [(1009, 1109, 9, 20)]),
                     ("10M50N5M80N10M1D2I30M200N10M", [(1186, 1386, 41, 10)])):
        dev, ora, _ = _run_both(_one_read(_cig(cig)), ["chr1"], p)
        got = [(int(s), int(e), int(s - a), int(b - e)) for s, e, a, b in
               zip(dev["intron_start"], dev["intron_end"], dev["anchor_start"], dev["anchor_end"])]
        assert got == exp, cig
        helpers.compare_tables(dev, ora)


def test_errors_earliest_in_input_order():
    from intronprospector_amd import synth
    import intronprospector_amd as ipa
    batch, names, _ = synth.short_reads(5000, seed=16, n_targets=2, target_len=3_000_000, n_sites=100)
    # plant: a padding op (P) in read 3000 and a B op in read 1200 -> the B must be reported
    bad = synth.concat_batches([
        synth.slice_batch(batch, 0, 1200), _one_read(_cig("10M100N10M5B")),
        synth.slice_batch(batch, 1200, 3000), _one_read(_cig("10M5P100N10M")),
        synth.slice_batch(batch, 3000, 5000)])
    dev, ora, err = _run_both(bad, names, {})
    assert err is not None and err.index == 1200 and err.msg == "Unknown cigar operation 'B'"
    e = dev["error"]
    assert e.kind == 1 and e.read_index == 1200 and e.detail == "Unknown cigar operation 'B'"
    # FUNMAP read with a spliced CIGAR reaches get_read_category -> logic_error
    bad = synth.concat_batches([synth.slice_batch(batch, 0, 10), _one_read(_cig("30M100N30M"), flag=4)])
    dev, ora, err = _run_both(bad, names, {})
    assert err is not None and err.index == 10
    assert dev["error"].kind == 2 and dev["error"].read_index == 10
    assert dev["error"].detail == err.msg


def test_genome_fetch_errors(tmp_path):
    from intronprospector_amd import synth
    gen = synth.Genome(["chr1", "chr2"], [np.frombuffer(b"ACGT" * 500, np.uint8).copy(), np.frombuffer(b"ACGT" * 500, np.uint8).copy()])
    fa = str(tmp_path / "g.fa")
    gen.write_fasta(fa)
    # intron running past the end of the 2000-base sequence
    rd = _one_read(_cig("20M1950N20M"), pos=100)
    dev, ora, err = _run_both(rd, ["chr1", "chr2"], {"junction_filter": 1}, genome=gen, fasta=fa)
    assert err is not None and dev["error"].kind == 4
    assert dev["error"].detail == err.msg
    # chromosome not in the genome
    rd = _one_read(_cig("20M100N20M"), pos=100, tid=2)
    dev, ora, err = _run_both(rd, ["chr1", "chr2", "chrX"], {"junction_filter": 1}, genome=gen, fasta=fa)
    assert err is not None and dev["error"].kind == 3
    assert dev["error"].detail == err.msg
    # ... tolerated with skip_missing_targets: junction kept with empty motif
    dev, ora, err = _run_both(rd, ["chr1", "chr2", "chrX"], {"junction_filter": 2, "skip_missing_targets": 1}, genome=gen, fasta=fa)
    assert err is None and len(dev["intron_start"]) == 1 and dev["motif"][0] == 0


def test_empty_and_degenerate_inputs():
    from intronprospector_amd import synth
    ex = _extractor(targets=["chr1"])
    t = ex.get_junctions()
    assert len(t["intron_start"]) == 0 and t["n_records"] == 0
    # reads without junctions, unmapped reads, single-op CIGARs
    b = synth.concat_batches([_one_read(_cig("100M")), _one_read(_cig("50M50S")), _one_read([], tid=-1),
                              _one_read(_cig("10M100N10M"), flag=1024), _one_read(_cig("10M100N10M"), flag=512),
                              _one_read(_cig("10M100N10M"), flag=256)])
    ex.push(b)
    t = ex.get_junctions()
    assert len(t["intron_start"]) == 0 and t["n_alignments"] == 6
    ex.clear()
    ex.push(_one_read(_cig("10M100N10M")))
    t = ex.get_junctions()
    assert t["intron_start"].tolist() == [1010] and t["intron_end"].tolist() == [1110]
    ex.close()


@pytest.mark.parametrize("ops", [(2, 9), (6, 16), (17, 40), (30, 400)])
@pytest.mark.parametrize("max_indel", [0, 4, 36])
def test_random_cigars_match_oracle(ops, max_indel, monkeypatch):
    """CIGARs without biological structure through the short-CIGAR kernel (2-9 ops; 6-16 ops: most
    alignments have more than 8 and go through its second stage, k_scan_rest) and the long-CIGAR kernel
    (more than 16 ops per alignment on average)."""
    n = 6000 if ops[1] < 100 else 1500
    if ops == (6, 16):
        monkeypatch.setenv("IPX_SHORT_MAX_AVG", "16")      # by default such a batch goes to the long-CIGAR kernel
    batch = helpers.fuzz_batch(n, seed=61 + ops[0], min_ops=ops[0], max_ops=ops[1])
    dev, ora, err = _run_both(batch, ["chrB", "chrA"], {"max_anchor_indel_size": max_indel, "min_anchor_length": 3,
                                                      "min_intron_length": 70})
    assert err is None
    assert len(ora["intron_start"]) > 20
    helpers.compare_tables(dev, ora)
    # the same through three pushes
    dev3, _, _ = _run_both(batch, ["chrB", "chrA"], {"max_anchor_indel_size": max_indel, "min_anchor_length": 3,
                                                    "min_intron_length": 70}, n_push=3)
    helpers.compare_tables(dev3, ora)


def test_long_cigar_kernel_errors_and_wide_offsets():
    """The long-CIGAR kernel walks 32-bit offsets from the alignment's position; alignments whose
    offsets leave 32 bits are finished by the 64-bit walk.  Both must agree with the oracle's
    int64 arithmetic, including the records that precede the wide part."""
    from intronprospector_amd import synth
    big = "268435455"                       # 2^28 - 1, the longest op BAM can hold
    filler = "5M1I" * 12                    # keeps the batch in the long-CIGAR kernel
    cases = [
        # fused intron of 2^32 + 1000 bases: rejected (a 32-bit difference would read 1000 and pass)
        filler + "20M100N20M" + (big + "N") * 16 + "1016N" + "20M",
        # left anchor of 2^32 + 5 bases: kept (a 32-bit difference would read 5 and fail min-anchor 8)
        filler + "20M100N20M" + (big + "M") * 16 + "1M100N30M",
        # wide offsets reached inside deletions; junctions before and after
        filler + "20M100N20M" + (big + "D") * 17 + "20M200N20M300N25M",
        # the carrying op itself ends a junction
        filler + "20M100N" + (big + "M") * 15 + "268435000M" + "5000N" + "20M",
    ]
    p = {"max_anchor_indel_size": 0xffffffff, "min_intron_length": 70, "max_intron_length": 0xffffffff}
    for cig in cases:
        for pos in (1000, 2_000_000_000):
            rd = _one_read(_cig(cig), pos=pos)
            assert len(rd["cigar"]) > 16
            dev, ora, err = _run_both(rd, ["chr1"], p)
            assert err is None
            helpers.compare_tables(dev, ora)
    # several such alignments among ordinary long ones
    batch, names, _ = synth.long_reads(300, seed=71, n_targets=2, target_len=9_000_000, n_transcripts=20)
    # (each at its own position: where two alignments that END beyond 2^32 share a junction, the
    # reference compares a 64-bit anchor with the stored 32-bit one, see DESIGN.md "Limits")
    mixed = synth.concat_batches([synth.slice_batch(batch, 0, 100), _one_read(_cig(cases[1]), pos=5000),
                                  synth.slice_batch(batch, 100, 250), _one_read(_cig(cases[2]), tid=1),
                                  _one_read(_cig(cases[3]), pos=9000), synth.slice_batch(batch, 250, 300)])
    dev, ora, err = _run_both(mixed, names, {"max_anchor_indel_size": 36})
    assert err is None
    helpers.compare_tables(dev, ora)
    # earliest error in input order, with the number of records before it
    bad = helpers.fuzz_batch(1200, seed=72, min_ops=20, max_ops=60, bad_ops=True)
    dev, ora, err = _run_both(bad, ["chr1", "chr2"], {"max_anchor_indel_size": 4})
    assert err is not None and dev["error"].kind == 1
    assert dev["error"].read_index == err.index and dev["error"].detail == err.msg


def test_striped_append_with_uneven_stripes_and_capacity_retry():
    """Scan launches of 64 or more blocks append through eight striped counters and close the gaps
    afterwards (ipx_scan.hip).  Uneven stripes: all junctions come from the first tiles of a large
    batch.  Capacity retry: a first push without junctions teaches the handle a low records-per-word
    ratio, the second push has one junction per alignment and must regrow the record arrays."""
    from intronprospector_amd import synth
    import intronprospector_amd as ipa
    spliced, names, _ = synth.short_reads(40_000, seed=91, n_targets=2, target_len=4_000_000, n_sites=700)
    n_plain = 90_000
    plain = {"tid": np.zeros(n_plain, np.int32), "pos": (np.arange(n_plain, dtype=np.int32) * 7 + 100),
             "flag": np.zeros(n_plain, np.uint16), "xs": np.zeros(n_plain, np.uint8), "nh": np.ones(n_plain, np.uint8),
             "cigar_off": (np.arange(n_plain, dtype=np.uint32) * 2), "n_cigar": np.full(n_plain, 2, np.uint32),
             "cigar": np.tile(np.array([(60 << 4) | 0, (40 << 4) | 4], np.uint32), n_plain)}      # 60M40S: no junction
    # uneven: junction reads first, then 90 k alignments that emit nothing (same launch, >= 64 blocks)
    uneven = synth.concat_batches([synth.slice_batch(spliced, 0, 6000), plain])
    dev, ora, err = _run_both(uneven, names, {})
    assert err is None and len(ora["intron_start"]) > 30 and int(ora["counts"].sum()) > 3000
    helpers.compare_tables(dev, ora)
    # retry: the ratio learnt from `plain` makes the next capacity guess far too small
    ex = ipa.JunctionsExtractor(targets=names)
    ex.push(plain)
    assert len(ex.get_junctions()["intron_start"]) == 0
    ex.push(spliced)
    ex.push(synth.slice_batch(spliced, 0, 20_000))
    dev = ex.get_junctions()
    ex.close()
    o = Oracle(targets=names)
    for b in (plain, spliced, synth.slice_batch(spliced, 0, 20_000)):
        o.process(b)
    helpers.compare_tables(dev, o.table())
    o.close()


def test_clipped_and_unspliced_short_cigars(monkeypatch):
    """The short-read scan's closed forms beyond "aM nN bM": one S or H clip in front and / or at the end keeps the closed forms;
    up to four ops without an N are dropped without a walk (unless one of them is an op the walk throws on); everything
    else -- two clips in a row, = and X, indels next to introns -- waits for a walk pass.  All of it against the oracle,
    in sorted order, several tiles' worth, with min-anchor / intron limits that reject some."""
    from intronprospector_amd import synth
    monkeypatch.setenv("IPX_SHORT_MAX_AVG", "16")
    templates = ["aSbMcNdM", "aMbNcMdS", "aHbMcNdMeS", "aSbMcNdMeH", "aMbNcMdNeM", "aSbMcNdMeNfM", "aMbNcMdNeMfS",
                 "aHbMcNdMeNfMgS", "aHbSbMcNdM", "aMbNcMdSeH", "aMbS", "aSbM", "aSbMcS", "aMbDcM", "aMbIcM", "aMbDcMdS",
                 "aMbNcM", "a=bNcM", "aMbNcX", "aSbMcNdMeIfM", "aMbNcN", "aNbMcNdM", "aSbNcMdNeM", "aMbNcMdNeMfNgMhS",
                 "aSbMcNdMeNfMgNhM", "aMbScNdM", "aMbHcM", "aM1DbMcNdM"]
    rng = np.random.default_rng(407)
    n = 60_000
    words, offs, ncs = [], [], []
    for _ in range(n):
        tp = templates[int(rng.integers(0, len(templates)))]
        cig = []
        num = None
        for ch in tp:
            if ch.isdigit():
                num = int(ch)
            elif ch.islower():
                num = None
            else:
                op = "MIDNSHP=XB".index(ch)
                ln = num if num is not None else int({3: rng.choice([40, 80, 150, 900, 600_000]), 1: rng.integers(1, 5),
                                                        2: rng.integers(1, 5)}.get(op, rng.integers(3, 45)))
                cig.append((ln << 4) | op)
                num = None
        offs.append(len(words)); ncs.append(len(cig)); words.extend(cig)
    batch = {"tid": np.sort(rng.integers(0, 3, n)).astype(np.int32), "pos": np.zeros(n, np.int32),
             "flag": rng.choice(np.array([0, 16, 99, 147, 256], np.uint16), n), "xs": rng.choice(np.array([0, 43, 45], np.uint8), n),
             "nh": rng.choice(np.array([0, 1, 1, 1, 3], np.uint8), n), "cigar_off": np.array(offs, np.uint32),
             "n_cigar": np.array(ncs, np.uint32), "cigar": np.array(words, np.uint32)}
    for t in range(3):
        m = batch["tid"] == t
        batch["pos"][m] = np.sort(rng.integers(1000, 200_000, int(m.sum()))).astype(np.int32)
    names = ["chr1", "chr2", "chr3"]
    for params in ({}, {"min_anchor_length": 12, "max_anchor_indel_size": 2, "min_intron_length": 70}):
        dev, ora, err = _run_both(batch, names, params)
        assert err is None and len(ora["intron_start"]) > 1000
        helpers.compare_tables(dev, ora)
    # an op the walk throws on, in an alignment that would otherwise be dropped unseen; the earliest one is reported
    bad = synth.concat_batches([synth.slice_batch(batch, 0, 700), _one_read(_cig("30M6P30M")), synth.slice_batch(batch, 700, 2000),
                                _one_read(_cig("5S30M4B"))])
    dev, ora, err = _run_both(bad, names, {})
    assert err is not None and err.index == 700 and err.msg == "Unknown cigar operation 'P'"
    assert dev["error"].kind == 1 and dev["error"].read_index == 700 and dev["error"].detail == err.msg


def test_short_cigar_tiles_that_overflow_the_record_pool(monkeypatch):
    """Three junctions per alignment: far more records per tile than the scan's LDS pool holds (rounds 1-4: 768 per
    block, and the tile went to k_scan_rest; round 5: 80 per wave, flushed as often as it fills), here together with
    the striped append (>= 64 blocks) and with a tail of ordinary alignments in the same launch."""
    from intronprospector_amd import synth
    monkeypatch.setenv("IPX_SHORT_MAX_AVG", "16")          # 5.6 words per alignment go there anyway; be explicit
    rng = np.random.default_rng(95)
    n = 40_000
    pos = np.sort(rng.integers(1000, 3_000_000, n)).astype(np.int32)
    a = rng.integers(9, 30, (n, 4)).astype(np.uint32)
    il = rng.choice(np.array([80, 150, 400, 2500], np.uint32), (n, 3))
    words = np.empty((n, 7), np.uint32)
    words[:, 0::2] = a << 4                                  # M
    words[:, 1::2] = (il << 4) | 3                           # N
    dense = {"tid": (rng.integers(0, 2, n)).astype(np.int32), "pos": pos, "flag": np.zeros(n, np.uint16),
             "xs": np.zeros(n, np.uint8), "nh": np.ones(n, np.uint8),
             "cigar_off": (np.arange(n, dtype=np.uint32) * 7), "n_cigar": np.full(n, 7, np.uint32),
             "cigar": words.reshape(-1)}
    usual, names, _ = synth.short_reads(30_000, seed=96, n_targets=2, target_len=4_000_000, n_sites=500)
    batch = synth.concat_batches([dense, usual])
    dev, ora, err = _run_both(batch, names, {})
    assert err is None and int(ora["counts"].sum()) > 3 * n
    helpers.compare_tables(dev, ora)
    dev2, _, _ = _run_both(batch, names, {}, n_push=3)
    helpers.compare_tables(dev2, ora)


def test_overflowing_tiles_that_also_hold_long_cigars_are_walked_once(monkeypatch):
    """A tile can be listed for the second stage twice over: its alignments of more than 8 ops one by one,
    and -- when its records overflow the pool -- all of it.  Every alignment must still be walked exactly
    once (counts would double otherwise).  Every fifth alignment here has 11 ops and four junctions, the
    others 7 ops and three; then one of the long ones gets a bad op (the error must be reported as the oracle's)."""
    monkeypatch.setenv("IPX_SHORT_MAX_AVG", "16")          # 7.8 words per alignment: the long-CIGAR kernel by default
    rng = np.random.default_rng(97)
    n = 30_000
    pos = np.sort(rng.integers(1000, 3_000_000, n)).astype(np.int32)
    words, offs, ncs = [], [], []
    total = 0
    lens = np.array([80, 150, 400, 2500], np.uint32)
    for i in range(n):
        k_intr = 4 if i % 5 == 0 else 3
        w = []
        for j in range(k_intr):
            w.append((int(rng.integers(9, 30)) << 4) | 0)
            w.append((int(lens[rng.integers(0, 4)]) << 4) | 3)
        w.append((int(rng.integers(9, 30)) << 4) | 0)
        if i % 5 == 0:
            w += [(2 << 4) | 2, (12 << 4) | 0]                     # ... D M: 11 ops
        offs.append(total); ncs.append(len(w)); total += len(w)
        words.extend(w)
    batch = {"tid": rng.integers(0, 2, n).astype(np.int32), "pos": pos, "flag": np.zeros(n, np.uint16),
             "xs": np.zeros(n, np.uint8), "nh": np.ones(n, np.uint8),
             "cigar_off": np.array(offs, np.uint32), "n_cigar": np.array(ncs, np.uint32),
             "cigar": np.array(words, np.uint32)}
    dev, ora, err = _run_both(batch, ["chr1", "chr2"], {})
    assert err is None and int(ora["counts"].sum()) > 3 * n
    helpers.compare_tables(dev, ora)
    dev3, _, _ = _run_both(batch, ["chr1", "chr2"], {}, n_push=3)
    helpers.compare_tables(dev3, ora)
    # room for 1000 records only: the first stage fits (it stores next to nothing here), the second does not ->
    # the record arrays grow and both stages run again
    ex = _extractor(targets=["chr1", "chr2"])
    ex.reserve(0, 0, 1000)
    ex.push(batch)
    small = ex.get_junctions()
    ex.close()
    helpers.compare_tables(small, ora)
    # a bad op in one of the listed alignments: same earliest error as the oracle
    bad = dict(batch)
    bad["cigar"] = batch["cigar"].copy()
    k = 5 * 1234
    bad["cigar"][offs[k] + 9] = (3 << 4) | 6                       # 'P'
    dev, _, err = _run_both(bad, ["chr1", "chr2"], {})
    assert err is not None and err.index == k and err.msg == "Unknown cigar operation 'P'"
    e = dev["error"]
    assert e.kind == 1 and e.read_index == k and e.detail == "Unknown cigar operation 'P'"


def test_ijunc_by_radix_sort_when_the_bitmap_does_not_apply(monkeypatch):
    """ijunc normally comes from a bitmap over the squeezed creation-order keys; IPX_RANK=sort forces
    the radix sort that serves keys with more than 28 varying bits or more than four runs of them."""
    from intronprospector_amd import synth
    monkeypatch.setenv("IPX_RANK", "sort")
    batch, names, _ = synth.short_reads(40_000, seed=81, n_targets=3, target_len=4_000_000, n_sites=900)
    dev, ora, err = _run_both(batch, names, {}, n_push=2)
    assert err is None
    helpers.compare_tables(dev, ora)
    batch, names, _ = synth.long_reads(600, seed=82, n_targets=2, target_len=9_000_000, n_transcripts=30)
    dev, ora, err = _run_both(batch, names, {"max_anchor_indel_size": 36})
    assert err is None
    helpers.compare_tables(dev, ora)


@pytest.mark.parametrize("tile", ["2048", "4096"])
def test_both_sort_tile_sizes_on_small_and_ragged_inputs(monkeypatch, tile):
    """The radix passes move 2048-item tiles (512 threads) on small inputs and 4096-item tiles (1024
    threads) from 3 M records on; IPX_SORT_TILE forces either one, here on record counts that are not a
    multiple of the tile and on one that is smaller than a tile."""
    from intronprospector_amd import synth
    monkeypatch.setenv("IPX_SORT_TILE", tile)
    for reads, seed in ((70_001, 61), (1_500, 62), (263_000, 63)):
        batch, names, _ = synth.short_reads(reads, seed=seed, n_targets=3, target_len=6_000_000, n_sites=2500)
        dev, ora, err = _run_both(batch, names, {}, n_push=2)
        assert err is None
        helpers.compare_tables(dev, ora)
    batch, names, _ = synth.long_reads(900, seed=64, n_targets=2, target_len=9_000_000, n_transcripts=40)
    dev, ora, err = _run_both(batch, names, {"max_anchor_indel_size": 36})
    assert err is None
    helpers.compare_tables(dev, ora)


def test_per_operation_scan_kernel_matches_oracle(monkeypatch):
    """ipx_scan_ops.hip (one lane per CIGAR operation, IPX_SCAN=ops) against the oracle on short,
    long and quirk inputs; it must also fall back cleanly on a CIGAR array that is not laid out
    back to back."""
    from intronprospector_amd import synth
    monkeypatch.setenv("IPX_SCAN", "ops")
    batch, names, _ = synth.short_reads(60_000, seed=51, n_targets=3, target_len=4_000_000, n_sites=1500)
    dev, ora, err = _run_both(batch, names, {"min_anchor_length": 3})
    assert err is None
    helpers.compare_tables(dev, ora)
    for max_indel in (0, 3, 36):
        batch, names, _ = synth.long_reads(800, seed=52, n_targets=2, target_len=9_000_000, n_transcripts=40)
        dev, ora, err = _run_both(batch, names, {"max_anchor_indel_size": max_indel})
        assert err is None
        helpers.compare_tables(dev, ora)
    p = {"max_anchor_indel_size": 3, "min_intron_length": 10}
    for cig in ("20M5D9M100N20M", "9M100N20M", "10M50N5M80N10M1D2I30M200N10M", "5S10M100N10M2I10M100N3D10M5H",
                "10M100N100N10M", "10M100N2D100N10M", "10M100N9D100N10M", "10M100N10M100N10M5P"):
        dev, ora, _ = _run_both(_one_read(_cig(cig)), ["chr1"], p)
        helpers.compare_tables(dev, ora)
    # scattered CIGAR layout: alignments in reverse word order -> lane-per-alignment fallback
    batch, names, _ = synth.short_reads(3000, seed=53, n_targets=2, target_len=3_000_000, n_sites=100)
    n = len(batch["tid"])
    rev_off = np.zeros(n, np.uint32)
    words = []
    pos = 0
    for i in range(n - 1, -1, -1):
        o, k = int(batch["cigar_off"][i]), int(batch["n_cigar"][i])
        rev_off[i] = pos
        words.append(batch["cigar"][o:o + k])
        pos += k
    scattered = dict(batch)
    scattered["cigar_off"] = rev_off
    scattered["cigar"] = np.concatenate(words)
    ex = _extractor(targets=names)
    ex.push(scattered)
    dev = ex.get_junctions()
    ora = Oracle(targets=names)
    ora.process(scattered)
    helpers.compare_tables(dev, ora.table())
    ora.close()
    # offsets that point outside the CIGAR array are reported, never dereferenced
    broken = dict(batch)
    broken["cigar_off"] = batch["cigar_off"] + np.uint32(len(batch["cigar"]))
    ex.clear()
    ex.push(broken)
    t = ex.get_junctions()
    assert t["error"].kind == 5 and "outside the batch" in t["error"].detail
    ex.close()


@pytest.mark.parametrize("which", ["1", "3"])
@pytest.mark.parametrize("max_indel", [0, 3, 36])
def test_both_long_scan_kernels_match_oracle(max_indel, which, monkeypatch):
    """k_scan_long (IPX_LONG=1): every op through the full state machine, lanes in lock step;
    k_scan_long3 (the default): plain ops one by one, the full state machine once per turn, CIGAR words through
    a per-lane ring in LDS"""
    from intronprospector_amd import synth
    monkeypatch.setenv("IPX_LONG", which)
    batch, names, _ = synth.long_reads(3000, seed=71, n_targets=2, target_len=9_000_000, n_transcripts=60)
    dev, ora, err = _run_both(batch, names, {"max_anchor_indel_size": max_indel}, n_push=2)
    assert err is None
    helpers.compare_tables(dev, ora)


def test_very_long_and_very_short_cigars_side_by_side():
    """a wave of the long-read kernel whose lanes hold CIGARs of 2 to 300 000 operations: lanes finish and take new
    alignments while others are hundreds of turns into theirs; ring refills at every alignment of the array"""
    rng = np.random.default_rng(97)
    M, I, D, N, S, EQ, X = 0, 1, 2, 3, 4, 7, 8
    reads = []
    for n_ops in [300_000, 2, 3, 70_001, 5, 17, 33, 64, 65, 4, 120_000] + [int(x) for x in rng.integers(2, 400, 120)]:
        ops = []
        for k in range(n_ops):
            r = rng.random()
            if k % 2 == 0:
                ops.append((int(rng.integers(1, 60)) << 4) | (M if r < 0.8 else (EQ if r < 0.9 else X)))
            elif r < 0.05:
                ops.append((int(rng.integers(20, 3000)) << 4) | N)
            elif r < 0.07:
                ops.append((int(rng.integers(1, 80)) << 4) | (I if r < 0.06 else D))      # some beyond the indel limit
            else:
                ops.append((int(rng.integers(1, 4)) << 4) | (I if r < 0.5 else D))
        if rng.random() < 0.5:
            ops[0] = (7 << 4) | S
        reads.append(ops)
    nc = np.array([len(r) for r in reads], np.uint32)
    off = np.zeros(len(reads), np.uint32)
    off[1:] = np.cumsum(nc[:-1])
    batch = {"tid": np.zeros(len(reads), np.int32), "pos": np.sort(rng.integers(0, 1000, len(reads))).astype(np.int32),
             "flag": rng.choice(np.array([0, 16, 256], np.uint16), len(reads)), "xs": np.zeros(len(reads), np.uint8),
             "nh": np.ones(len(reads), np.uint8), "cigar_off": off, "n_cigar": nc,
             "cigar": np.concatenate([np.array(r, np.uint32) for r in reads])}
    for max_indel in (3, 36):
        dev, ora, err = _run_both(batch, ["chr1"], {"max_anchor_indel_size": max_indel, "min_anchor_length": 4})
        assert err is None
        helpers.compare_tables(dev, ora)


def test_long_reads_in_device_memory_with_an_unaligned_cigar_array():
    """k_scan_long3 reads the CIGAR array in aligned 16-byte pieces; an array that starts elsewhere goes to
    k_scan_long (ipx_launch_scan's fallback) -- both through device-resident batches, against the oracle"""
    import torch
    from intronprospector_amd import synth
    batch, names, _ = synth.long_reads(600, seed=91, n_targets=2, target_len=9_000_000, n_transcripts=30)
    ora = Oracle(targets=names, max_anchor_indel_size=36)
    ora.process(batch)
    want = ora.table()
    ora.close()
    dev = torch.device("cuda:0")
    n, nw = len(batch["tid"]), len(batch["cigar"])
    for shift in (0, 1, 3):                                 # words in front of the array: 0 = aligned
        t = {k: torch.from_numpy(np.ascontiguousarray(batch[k], dt)).to(dev) for k, dt in (
            ("tid", np.int32), ("pos", np.int32), ("flag", np.uint16), ("xs", np.uint8), ("nh", np.uint8),
            ("cigar_off", np.uint32), ("n_cigar", np.uint32))}
        cig = torch.zeros(nw + 8, dtype=torch.int32, device=dev)
        cig[shift:shift + nw] = torch.from_numpy(batch["cigar"].astype(np.uint32).view(np.int32)).to(dev)
        assert cig.data_ptr() % 16 == 0
        ptrs = [t[k].data_ptr() for k in ("tid", "pos", "flag", "xs", "nh", "cigar_off", "n_cigar")] + [cig.data_ptr() + 4 * shift]
        ex = _extractor(targets=names, max_anchor_indel_size=36)
        ex.push_device(n, nw, ptrs)
        got = ex.get_junctions()
        ex.close()
        assert "error" not in got
        helpers.compare_tables(got, want)


def test_combiner_on_and_off_give_the_same_table(monkeypatch):
    """IPX_COMBINE=0: every junction record reaches the sort on its own; the table must not notice"""
    import hashlib
    from intronprospector_amd import synth
    batch, names, _ = synth.short_reads(150_000, seed=72, n_targets=3, target_len=3_000_000, n_sites=300)
    digests, n_rec = [], []
    for env in ("1", "0"):
        monkeypatch.setenv("IPX_COMBINE", env)
        ex = _extractor(targets=names)
        ex.push(batch)
        n_rec.append(ex.records_pending())
        t = ex.get_junctions()
        ex.close()
        h = hashlib.sha256()
        for k in ("target", "intron_start", "intron_end", "anchor_start", "anchor_end", "strand", "ijunc", "counts",
                  "hist_begin", "hist_value", "hist_count", "creator_order"):
            h.update(np.ascontiguousarray(t[k]).tobytes())
        digests.append((h.hexdigest(), t["n_records"]))
    assert digests[0] == digests[1]
    assert n_rec[0] * 3 < n_rec[1] == digests[1][1]        # hot junctions: far fewer records leave the scan


def _pileup(cigars, n, seed, n_pos=40, nh_mix=True):
    """n alignments drawn from a few CIGARs at a few positions of one locus, coordinate-sorted: the pile-up of a
    highly expressed gene, every tile full of copies of a handful of (junction, offset) pairs"""
    rng = np.random.default_rng(seed)
    words = [np.array(_cig(c), np.uint32) for c in cigars]
    which = rng.integers(0, len(words), n)
    pos = np.sort(rng.integers(50_000, 50_000 + n_pos, n)).astype(np.int32)
    nc = np.array([len(words[w]) for w in which], np.uint32)
    off = np.zeros(n, np.uint32)
    off[1:] = np.cumsum(nc[:-1])
    return {"tid": np.zeros(n, np.int32), "pos": pos,
            "flag": rng.choice(np.array([0, 16, 99, 147, 256], np.uint16), n),
            "xs": np.zeros(n, np.uint8), "nh": (rng.integers(0, 3, n) if nh_mix else np.ones(n)).astype(np.uint8),
            "cigar_off": off, "n_cigar": nc, "cigar": np.concatenate([words[w] for w in which])}


@pytest.mark.parametrize("cigars", [
    ["30M100N20M200N50M"],                                        # two records per read: 1024 per tile before folding
    ["30M100N20M200N50M", "30M100N20M200N25M300N25M", "10S20M100N20M200N15M300N20M150N25M"],   # walks with 3-4 records
    ["50M500N50M", "30M100N20M200N50M", "2S28M100N20M200N50M", "30M100N10M1D9M200N50M"],
])
def test_pile_up_tiles_fold_before_they_overflow_the_pool(cigars, monkeypatch):
    """pile-ups: thousands of copies of a handful of (junction, offset) pairs, the copies of different pairs
    alternating in the input -- the scan folds them (runs in registers, the rest in its pool) instead of handing
    every copy to the sort; against the oracle, and against the combiner switched off"""
    monkeypatch.setenv("IPX_SHORT_MAX_AVG", "16")        # the short-CIGAR kernel whatever the mean CIGAR length
    batch = _pileup(cigars, 40_000, seed=len(cigars))
    p = {"min_anchor_length": 8, "max_anchor_indel_size": 3}
    dev, ora, err = _run_both(batch, ["chr1"], p, n_push=1)
    assert err is None
    helpers.compare_tables(dev, ora)
    ex = _extractor(targets=["chr1"], **p)
    ex.push(batch)
    folded = ex.records_pending()
    ex.close()
    # a few hundred keys, tens of thousands of records; reads of more than 8 ops (a third of the second case)
    # go through the second stage, which does not fold.  (Round 5: a WAVE owns a tile of 128 alignments and a batch of
    # this size gives every wave one tile, so what is left is a handful of records per 128 alignments -- keys that
    # alternate in the input, the third case, are folded by the pool's second fold, k_scan_short3's dedupe)
    limit = dev["n_records"] * (0.6 if any(len(_cig(c)) > 8 for c in cigars) else 0.08)
    assert folded < limit, (folded, dev["n_records"])
    monkeypatch.setenv("IPX_COMBINE", "0")
    dev0, _, _ = _run_both(batch, ["chr1"], p)
    helpers.compare_tables(dev0, ora)


def test_reduce_on_small_and_ragged_tables(tmp_path, monkeypatch):
    """k_reduce at the edges of its chunks of 512 records: one-record and one-junction tables, tables that end inside a
    chunk or a record after it, combined and uncombined records, no filter and the filter with fail-then-pass records
    (modes 1 and 2), long reads."""
    from intronprospector_amd import synth
    for n, seed, sites in ((1, 5, 3), (70, 6, 2), (513, 7, 40), (40_000, 8, 150), (150_000, 9, 30_000)):
        batch, names, _ = synth.short_reads(n, seed=seed, n_targets=2, target_len=3_000_000, n_sites=sites)
        for comb in ("1", "0"):
            monkeypatch.setenv("IPX_COMBINE", comb)
            dev, ora, err = _run_both(batch, names, {})
            assert err is None
            helpers.compare_tables(dev, ora)
    monkeypatch.delenv("IPX_COMBINE")
    batch, names, genome = synth.short_reads(60_000, seed=31, n_targets=2, target_len=3_000_000, n_sites=900, genome=True)
    fa = str(tmp_path / "g.fa")
    genome.write_fasta(fa)
    for flt in (1, 2):
        dev, ora, err = _run_both(batch, names, {"junction_filter": flt}, genome=genome, fasta=fa)
        assert err is None
        helpers.compare_tables(dev, ora, _fetcher(genome))
    lr, names, _ = synth.long_reads(600, seed=77, n_targets=2, target_len=3_000_000, n_transcripts=30)
    dev, ora, err = _run_both(lr, names, {"max_anchor_indel_size": 36})
    assert err is None
    helpers.compare_tables(dev, ora)


def test_records_export_needs_the_combiner_off():
    """ipx_records_export hands out single junction records: refused once records were combined, fine with
    IPX_OPT_COMBINE 0 (what the command line sets for --debug-trace and the strand tags); the option cannot change
    while records are pending"""
    import ctypes as C
    from intronprospector_amd import synth, _lib as L
    batch = _pileup(["50M500N50M"], 5000, seed=9)

    class Rec(C.Structure):
        _fields_ = [("order", C.c_uint64), ("target", C.c_uint32), ("intron_start", C.c_uint32), ("intron_end", C.c_uint32),
                    ("anchor_start", C.c_uint32), ("anchor_end", C.c_uint32), ("read_offset", C.c_uint32)]
    for combine in (True, False):
        ex = _extractor(targets=["chr1"], combine=combine)
        ex.push(batch)
        n = ex.records_pending()
        recs = (Rec * max(n, 1))()
        got = C.c_uint64(0)
        rc = ex.lib.ipx_records_export(ex.h, recs, n, C.byref(got))
        if combine:
            assert rc == L.IPX_EINVAL and n < 500 and b"IPX_OPT_COMBINE" in ex.last_error().detail
        else:
            assert rc == 0 and got.value == n and n > 3000                  # secondary alignments are QC-rejected
            reads = sorted(int(r.order) >> 30 for r in recs[:n])
            assert len(set(reads)) == n                                       # one record per read here
        assert ex.lib.ipx_set_option(ex.h, L.IPX_OPT_COMBINE, 1) == L.IPX_EINVAL      # records pending
        ex.clear()
        assert ex.lib.ipx_set_option(ex.h, L.IPX_OPT_COMBINE, 1) == 0
        ex.close()
