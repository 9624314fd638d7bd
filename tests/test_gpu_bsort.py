"""GPU parity for the two sorts of ipx_finalize: the bucket sort (csrc/ipx_bsort.hip: sampled splitters, two radix
passes on a 16-bit leaf id, every leaf sorted in LDS) takes tables from 8 M records on, the LSD radix sort
(csrc/ipx_sort.hip) the rest and the bucket sort's overflow cases.  IPX_BSORT_MIN / IPX_SORT force either on any size.
Every case compares the whole device table with the oracle's (std::map order, junctions.hh:77-100)."""
import numpy as np
import pytest
import torch            # noqa: F401  (before libipx.so, as in test_gpu_parity.py)

from tests import helpers
from tests.test_gpu_parity import _fetcher, _run_both

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("reads,seed,n_push", [(1_500, 62, 1), (9_000, 65, 2), (70_001, 61, 2), (263_000, 63, 3)])
def test_bucket_sort_on_small_and_ragged_inputs(monkeypatch, reads, seed, n_push):
    """one leaf in one launch (fewer than 4096 records), a handful of leaves, a few hundred"""
    from intronprospector_amd import synth
    monkeypatch.setenv("IPX_BSORT_MIN", "1")
    batch, names, _ = synth.short_reads(reads, seed=seed, n_targets=3, target_len=6_000_000, n_sites=2500)
    dev, ora, err = _run_both(batch, names, {}, n_push=n_push)
    assert err is None
    helpers.compare_tables(dev, ora)


def test_bucket_sort_long_reads_and_unsorted_input(monkeypatch):
    from intronprospector_amd import synth
    monkeypatch.setenv("IPX_BSORT_MIN", "1")
    batch, names, _ = synth.long_reads(4_000, seed=64, n_targets=2, target_len=9_000_000, n_transcripts=60)
    dev, ora, err = _run_both(batch, names, {"max_anchor_indel_size": 36})
    assert err is None
    assert dev["n_records"] > 60_000
    helpers.compare_tables(dev, ora)
    batch, names, _ = synth.short_reads(150_000, seed=66, n_targets=4, target_len=8_000_000, n_sites=30_000)
    rng = np.random.default_rng(5)
    shuffled = synth.permute_batch(batch, rng.permutation(len(batch["tid"])))
    dev, ora, err = _run_both(shuffled, names, {}, n_push=2)
    assert err is None
    helpers.compare_tables(dev, ora)


def test_bucket_sort_with_genome_filter_and_global_scope(monkeypatch, tmp_path):
    from intronprospector_amd import synth
    monkeypatch.setenv("IPX_BSORT_MIN", "1")
    batch, names, genome = synth.short_reads(120_000, seed=67, n_targets=3, target_len=3_000_000, n_sites=5000, genome=True)
    fasta = str(tmp_path / "g.fa")
    genome.write_fasta(fasta)
    for scope in (0, 1):
        dev, ora, err = _run_both(batch, names, {"junction_filter": 1}, genome=genome, fasta=fasta, scope=scope, n_push=2)
        assert err is None
        helpers.compare_tables(dev, ora, _fetcher(genome))


def test_lsd_sort_still_serves_large_tables(monkeypatch):
    """IPX_SORT=lsd: the radix passes on a table the bucket sort would take by default"""
    from intronprospector_amd import synth
    monkeypatch.setenv("IPX_SORT", "lsd")
    batch, names, _ = synth.short_reads(400_000, seed=68, n_targets=4, target_len=20_000_000, n_sites=40_000)
    dev, ora, err = _run_both(batch, names, {}, n_push=2)
    assert err is None
    helpers.compare_tables(dev, ora)


def test_bucket_sort_on_a_million_records(monkeypatch):
    """a few hundred leaves per coarse range, both leaf size classes"""
    from intronprospector_amd import synth
    monkeypatch.setenv("IPX_BSORT_MIN", "1")
    monkeypatch.setenv("IPX_COMBINE", "0")
    batch, names, _ = synth.short_reads(1_200_000, seed=69, n_targets=4, target_len=30_000_000, n_sites=400_000)
    dev, ora, err = _run_both(batch, names, {}, n_push=2)
    assert err is None
    assert dev["n_records"] > 1_000_000
    helpers.compare_tables(dev, ora)


def test_leaf_overflow_falls_back_to_the_radix_passes(monkeypatch, capfd):
    """A leaf that does not fit LDS (here: IPX_BSORT_LEAF_AVG makes one leaf of the whole table) raises the overflow
    flag and the LSD sort redoes the table; the handle then skips the bucket sort for its next finalizes."""
    from intronprospector_amd import synth
    monkeypatch.setenv("IPX_BSORT_MIN", "1")
    monkeypatch.setenv("IPX_BSORT_LEAF_AVG", "1000000000")
    monkeypatch.setenv("IPX_DEBUG_PLAN", "1")
    batch, names, _ = synth.short_reads(200_000, seed=70, n_targets=2, target_len=4_000_000, n_sites=3000)
    dev, ora, err = _run_both(batch, names, {})
    assert err is None
    helpers.compare_tables(dev, ora)
    assert "bucket sort overflowed" in capfd.readouterr().err


def test_keys_with_thousands_of_copies_get_leaves_of_their_own(monkeypatch, capfd):
    """Unsorted copies of a few keys, the combiner off: a key drawn as two splitters in a row is fenced into a leaf
    of its own (k_bs_sort2), which may be of any size because its keys are all equal."""
    from intronprospector_amd import synth
    monkeypatch.setenv("IPX_BSORT_MIN", "1")
    monkeypatch.setenv("IPX_COMBINE", "0")
    monkeypatch.setenv("IPX_DEBUG_PLAN", "1")
    batch, names, _ = synth.short_reads(2_000_000, seed=71, n_targets=2, target_len=4_000_000, n_sites=40)
    rng = np.random.default_rng(6)
    shuffled = synth.permute_batch(batch, rng.permutation(len(batch["tid"])))
    dev, ora, err = _run_both(shuffled, names, {})
    assert err is None
    helpers.compare_tables(dev, ora)
    assert "bucket sort overflowed" not in capfd.readouterr().err


def test_sorted_and_unsorted_batches_in_one_table(monkeypatch, capfd):
    """sorted batch, shuffled batch, shuffled batch, sorted batch, sorted batch in ONE table: the oracle's table, and the
    scan folds where there is something to fold (rounds 3-4 switched the tile combiner to a probing mode after a batch
    that merged next to nothing; k_scan_short3's passes find out for themselves -- a pass without equal neighbours skips
    its scans, a pool that folds little flushes at once -- so there is no mode, and no IPX_COMBINE_ADAPT, any more)"""
    from intronprospector_amd import synth
    import intronprospector_amd as ipa
    from tests.oracle_binding import Oracle
    monkeypatch.setenv("IPX_DEBUG_PLAN", "1")
    parts = []
    for k, shuffle in enumerate((False, True, True, False, False)):
        b, names, _ = synth.short_reads(300_000, seed=90 + k, n_targets=3, target_len=5_000_000, n_sites=4000)
        if shuffle:
            b = synth.permute_batch(b, np.random.default_rng(k).permutation(len(b["tid"])))
        parts.append(b)
    whole = synth.concat_batches(parts)
    ora = Oracle(targets=names)
    ora.process(whole)
    tab = ora.table()
    ora.close()
    ex = ipa.JunctionsExtractor(targets=names)
    for b in parts:
        ex.push(b)
    dev = ex.get_junctions()
    ex.close()
    helpers.compare_tables(dev, tab)
    lines = [l for l in capfd.readouterr().err.splitlines() if "scan of" in l]
    assert len(lines) == 5
    merged = [int(l.split(" records, ")[1].split(" merged away")[0]) for l in lines]
    made = [int(l.split(" alignments: ")[1].split(" records")[0]) for l in lines]
    # sorted batches fold most of their records, shuffled ones next to nothing
    assert all(merged[k] > 2 * made[k] for k in (0, 3, 4)) and all(merged[k] < made[k] // 4 for k in (1, 2)), lines


def test_a_table_of_one_key_needs_no_sorting(monkeypatch, capfd):
    """20 000 copies of ONE (junction, offset): the leaf is larger than LDS but its keys are all equal -- accepted."""
    monkeypatch.setenv("IPX_BSORT_MIN", "1")
    monkeypatch.setenv("IPX_COMBINE", "0")
    monkeypatch.setenv("IPX_DEBUG_PLAN", "1")
    n = 20_000
    cigar = np.tile(np.array([(40 << 4) | 0, (500 << 4) | 3, (60 << 4) | 0], np.uint32), n)
    batch = {"tid": np.zeros(n, np.int32), "pos": np.full(n, 1000, np.int32), "flag": np.zeros(n, np.uint16),
             "xs": np.zeros(n, np.uint8), "nh": np.ones(n, np.uint8), "cigar_off": (np.arange(n) * 3).astype(np.uint32),
             "n_cigar": np.full(n, 3, np.uint32), "cigar": cigar}
    dev, ora, err = _run_both(batch, ["chr1"], {})
    assert err is None
    helpers.compare_tables(dev, ora)
    assert len(dev["intron_start"]) == 1
    assert "bucket sort overflowed" not in capfd.readouterr().err
