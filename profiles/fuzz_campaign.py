"""Differential fuzzing of the device path against the oracle: random batches (synthetic short / long reads,
structure-free CIGARs of several length ranges, mixtures, random order), random parameters, random number of
pushes.  Prints one line per failure and a summary.  Run on a GPU box: python profiles/fuzz_campaign.py [N] [seed0]
IPX_FUZZ_BAM=1: pushes of up to 12 000 alignments go in as BAM records (ipx_push_bam) instead of the SoA."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import helpers
import bam_encode
import intronprospector_amd as ipa
from intronprospector_amd import synth
from oracle_binding import Oracle, OracleError


def make_genome(rng):
    """short reads over a synthetic genome: -> batch, names, extractor genome argument, FASTA path, junction filter"""
    import tempfile
    b, names, gen = synth.short_reads(int(rng.integers(2000, 40000)), seed=int(rng.integers(1, 1 << 30)), n_targets=3,
                                      target_len=2_700_000, n_sites=int(rng.integers(20, 900)), genome=True)
    fasta = os.path.join(tempfile.mkdtemp(), "g.fa")
    gen.write_fasta(fasta)
    gen2 = (gen.names, [(t, m, len(q)) for (t, m), q in zip(gen.twobit(), gen.seqs)])
    if rng.integers(0, 3) == 0:
        b = synth.permute_batch(b, rng.permutation(len(b["tid"])))
    def fetch(tid, start, end, seqs=gen.seqs):
        q = seqs[tid]
        return bytes([q[start], q[start + 1], q[end - 2], q[end - 1]])
    return b, names, gen2, fasta, int(rng.integers(0, 3)), fetch


def make(rng):
    kind = rng.integers(0, 6)
    names = ["chrB", "chrA", "chrC"]
    if kind == 0:
        b, names, _ = synth.short_reads(int(rng.integers(2000, 120000)), seed=int(rng.integers(1, 1 << 30)), n_targets=3,
                                        target_len=5_000_000, n_sites=int(rng.integers(20, 3000)))
    elif kind == 1:
        b, names, _ = synth.long_reads(int(rng.integers(100, 1500)), seed=int(rng.integers(1, 1 << 30)), n_targets=2,
                                       target_len=9_000_000, n_transcripts=int(rng.integers(5, 60)))
    elif kind == 2:
        lo = int(rng.integers(1, 8)); hi = lo + int(rng.integers(0, 10))
        b = helpers.fuzz_batch(int(rng.integers(500, 9000)), seed=int(rng.integers(1, 1 << 30)), min_ops=lo, max_ops=hi,
                               n_targets=3, bad_ops=bool(rng.integers(0, 4) == 0))
    elif kind == 3:
        b = helpers.fuzz_batch(int(rng.integers(200, 2500)), seed=int(rng.integers(1, 1 << 30)), min_ops=10,
                               max_ops=int(rng.integers(20, 300)), n_targets=3)
    elif kind == 4:      # mixture: mostly short reads, some structure-free ones
        b1, names, _ = synth.short_reads(int(rng.integers(5000, 60000)), seed=int(rng.integers(1, 1 << 30)), n_targets=3,
                                         target_len=5_000_000, n_sites=int(rng.integers(50, 2000)))
        b2 = helpers.fuzz_batch(int(rng.integers(100, 3000)), seed=int(rng.integers(1, 1 << 30)), min_ops=2,
                                max_ops=int(rng.integers(6, 30)), n_targets=3)
        b = synth.concat_batches([b1, b2])
    else:                # dense junctions: overflowing tiles
        n = int(rng.integers(3000, 40000))
        k = int(rng.integers(2, 5))
        words = np.empty((n, 2 * k + 1), np.uint32)
        words[:, 0::2] = rng.integers(5, 40, (n, k + 1)).astype(np.uint32) << 4
        words[:, 1::2] = (rng.choice(np.array([60, 80, 150, 400, 2500], np.uint32), (n, k)) << 4) | 3
        b = {"tid": rng.integers(0, 3, n).astype(np.int32), "pos": np.sort(rng.integers(1000, 3_000_000, n)).astype(np.int32),
             "flag": rng.choice(np.array([0, 16, 256, 99], np.uint16), n), "xs": np.zeros(n, np.uint8),
             "nh": rng.integers(0, 3, n).astype(np.uint8), "cigar_off": (np.arange(n, dtype=np.uint32) * (2 * k + 1)),
             "n_cigar": np.full(n, 2 * k + 1, np.uint32), "cigar": words.reshape(-1)}
    if rng.integers(0, 3) == 0:
        b = synth.permute_batch(b, rng.permutation(len(b["tid"])))
    return b, names


def main():
    as_bam = os.environ.get("IPX_FUZZ_BAM") == "1"
    n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    bad = 0
    n_err = 0
    for it in range(n_iter):
        rng = np.random.default_rng(seed0 + it)
        gen2, fasta, fetch = None, None, None
        with_genome = rng.integers(0, 4) == 0
        if with_genome:
            batch, names, gen2, fasta, jf, fetch = make_genome(rng)
        else:
            batch, names = make(rng)
        params = {"min_anchor_length": int(rng.choice([1, 3, 8, 20])), "max_anchor_indel_size": int(rng.choice([0, 4, 36])),
                  "min_intron_length": int(rng.choice([1, 70, 200])), "max_intron_length": int(rng.choice([2000, 500000, 4000000])),
                  "strandness": int(rng.integers(0, 3)), "excludes": int(rng.integers(0, 2))}
        if with_genome:
            params["junction_filter"] = jf
        n_push = int(rng.integers(1, 4))
        ex = ipa.JunctionsExtractor(targets=names, genome=gen2, **params)
        n = len(batch["tid"])
        cuts = np.linspace(0, n, n_push + 1).astype(int)
        for k, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
            part = synth.slice_batch(batch, int(a), int(b))
            if as_bam and 0 < b - a <= 12000:
                # the same alignments as BAM records, decoded on the device (ipx_push_bam); tests/bam_encode.py
                data, sb, sf, nw = bam_encode.records(part, seed0 + it + 7919 * k)
                ex.push_bam(data, sb, sf, nw)
            else:
                ex.push(part)
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
        try:
            if err is not None:      # the reference throws: only WHICH alignment and WHY is defined, not a table
                e = dev["error"]
                assert e.read_index == err.index and e.detail == err.msg, (e.read_index, err.index, e.detail, err.msg)
                n_err += 1
            else:
                assert "error" not in dev
                helpers.compare_tables(dev, tab, fetch)
        except AssertionError as x:
            bad += 1
            print(json.dumps({"seed": seed0 + it, "alignments": n, "params": params, "n_push": n_push, "failure": str(x)[:300]}))
    print(json.dumps({"iterations": n_iter, "first_seed": seed0, "inputs_with_a_data_error": n_err, "failures": bad}))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
