"""Pin the CPU oracle against the reference's own golden outputs.

Every rule of the reference's tests/Makefile.in that exercises the junction
extraction path is replayed through oracle/ipxo_cli with the
same options and compared byte for byte with the reference's expected files
(copied as data into tests/golden/).  BAM-input rules (testBasic3/4/5) use the
SAM form of the same alignments: samtools is not available, and the
reference's expected outputs for testBasic3 are identical to testBasic1.
"""
import os
import subprocess

import pytest

CONF0 = "--min-confidence-score=0.0"

# name -> (options, inputs, extra outputs {option: suffix}, stderr golden or None)
RULES = {
    "testBasic1": ([CONF0], ["test_basic.sam"], {"--intron-bed6": "introns.bed6"}, None),
    "testBasic2": (["--strandness=RF", CONF0], ["test_basic.sam"], {}, None),
    "testBasic3": ([CONF0], ["test_basic.sam"], {}, None),
    "testBasic4": (["--min-confidence-score=1.0"], ["test_basic.sam"], {}, None),
    "testBasic5": (["--exclude=multi", CONF0], ["test_basic.sam"], {}, None),
    "testMulti1": ([CONF0], ["test_multi.sam"], {}, None),
    "testFake10a": ([CONF0, "--genome-fasta={G}/fake10_genome.fa"], ["fake10.star.sam"],
                    {"--debug-trace": "debug.tsv"}, None),
    "testFake10aConf": (["--genome-fasta={G}/fake10_genome.fa"], ["fake10.star.sam"],
                        {"--debug-trace": "debug.tsv"}, None),
    "testFake10b": ([CONF0, "--genome-fasta={G}/fake10_genome.fa"], ["fake10.hisat2.sam"], {}, None),
    "testFake10c": ([CONF0, "--genome-fasta={G}/fake10_genome.fa", "--skip-missing-targets"],
                    ["fake10.missing.sam"], {}, "testFake10c.err"),
    "testSjFilterDefault": ([CONF0, "--genome-fasta={G}/fake10_scramble_genome.fa"],
                            ["fake10.star.sam"], {}, None),
    "testSjFilterCanon": ([CONF0, "--sj-filter=canon", "--genome-fasta={G}/fake10_scramble_genome.fa"],
                          ["fake10.star.sam"], {}, None),
    "testSjFilterAll": ([CONF0, "--sj-filter=all", "--genome-fasta={G}/fake10_scramble_genome.fa"],
                        ["fake10.star.sam"], {}, None),
    "testOnt1": ([CONF0], ["ont.sam"], {}, None),
    "testOntIndel": ([CONF0, "--allow-anchor-indels"], ["ont.sam"], {}, None),
    "testOntSkipTargets": ([CONF0, "--allow-anchor-indels"], ["ont-skip-targets.sam"], {}, None),
    "testMultiSam": ([CONF0], ["test_multi.one.sam", "test_multi.two.sam"], {}, None),
    "testOntUnsorted": (["--unsorted", "--allow-anchor-indels"], ["ont-out-of-order-targets.sam"], {}, None),
}


def run_rule(cli, golden, name, outdir):
    opts, inputs, extra, err_golden = RULES[name]
    gin = os.path.join(golden, "input")
    cmd = [cli] + [o.format(G=gin) for o in opts]
    outs = {"--junction-bed": "juncs.bed", "--intron-bed": "introns.bed", "--intron-calls": "introns.tsv"}
    outs.update(extra)
    for opt, suf in outs.items():
        cmd.append("%s=%s" % (opt, os.path.join(outdir, "%s.%s" % (name, suf))))
    cmd += [os.path.join(gin, i) for i in inputs]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()
    for suf in outs.values():
        exp_name = name
        with open(os.path.join(golden, "expected", "%s.%s" % (exp_name, suf)), "rb") as f:
            exp = f.read()
        with open(os.path.join(outdir, "%s.%s" % (name, suf)), "rb") as f:
            got = f.read()
        assert got == exp, "%s.%s differs from the reference's golden file" % (name, suf)
    if err_golden:
        with open(os.path.join(golden, "expected", err_golden), "rb") as f:
            assert r.stdout == f.read()


@pytest.mark.parametrize("name", sorted(RULES))
def test_oracle_reproduces_reference_golden(oracle_cli, golden, tmp_path, name):
    run_rule(oracle_cli, golden, name, str(tmp_path))


def test_oracle_gzip_outputs(oracle_cli, golden, tmp_path):
    """testBasic1gz: '.gz' output names are gzip streams of the same bytes."""
    import gzip
    gin = os.path.join(golden, "input")
    out = {s: str(tmp_path / ("b1." + s + ".gz")) for s in ("juncs.bed", "introns.bed", "introns.tsv")}
    subprocess.run([oracle_cli, CONF0, "--junction-bed=" + out["juncs.bed"],
                    "--intron-bed=" + out["introns.bed"], "--intron-calls=" + out["introns.tsv"],
                    os.path.join(gin, "test_basic.sam")], check=True)
    for suf, path in out.items():
        with open(os.path.join(golden, "expected", "testBasic1." + suf), "rb") as f:
            assert gzip.open(path, "rb").read() == f.read()


def test_oracle_out_of_order_targets_error(oracle_cli, golden):
    """testOntOutOfOrderTargets: sortedness error text and exit status."""
    r = subprocess.run([oracle_cli, "--allow-anchor-indels",
                        os.path.join(golden, "input", "ont-out-of-order-targets.sam")],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 1
    with open(os.path.join(golden, "expected", "testOntOutOfOrderTargets.err"), "rb") as f:
        assert r.stdout == f.read()


def _calls(cli, sam_text, tmp_path, *opts):
    sam = tmp_path / "in.sam"
    sam.write_text(sam_text)
    r = subprocess.run([cli, "--min-confidence-score=0", "--intron-calls=/dev/stdout"] + list(opts) + [str(sam)],
                       stdout=subprocess.PIPE, check=True)
    return [l.split("\t") for l in r.stdout.decode().splitlines()[1:]]


def _sam(cigars, pos=1001):
    s = "@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:chr1\tLN:1000000\n"
    for i, c in enumerate(cigars):
        s += "r%d\t0\tchr1\t%d\t60\t%s\t*\t0\t0\t*\t*\n" % (i, pos, c)
    return s


def test_known_answer_stale_indel_counters(oracle_cli, tmp_path):
    """outputs of the unmodified reference recorded by the survey
    for the indel counters that leak across a limbo break."""
    opts = ["--allow-anchor-indels", "--max-anchor-indel-size=3", "--min-intron-length=10"]
    assert _calls(oracle_cli, _sam(["20M5D9M100N20M"]), tmp_path, *opts) == []
    rows = _calls(oracle_cli, _sam(["9M100N20M"]), tmp_path, *opts)
    assert [(r[1], r[2], r[7], r[8]) for r in rows] == [("1009", "1109", "9", "20")]
    rows = _calls(oracle_cli, _sam(["10M50N5M80N10M1D2I30M200N10M"]), tmp_path, *opts)
    assert [(r[1], r[2], r[7], r[8]) for r in rows] == [("1186", "1386", "41", "10")]


def test_known_answer_fail_then_pass(oracle_cli, golden, tmp_path):
    """reference outputs recorded by the survey for junction
    fake10:130577-131433 seen by reads with XS -,+,- and left anchors 72/68/58."""
    sam = os.path.join(golden, "known", "fail_then_pass.sam")
    fa = os.path.join(golden, "input", "fake10_genome.fa")

    def run(*opts):
        r = subprocess.run([oracle_cli, "--min-confidence-score=0", "--genome-fasta=" + fa,
                            "--intron-calls=/dev/stdout"] + list(opts) + [sam],
                           stdout=subprocess.PIPE, check=True)
        return r.stdout.decode().splitlines()[1:]

    assert run() == ["fake10\t130577\t131433\t+\t2\t0\t0\t68\t43\t1\tGT/AG\tsj0"]
    assert run("--sj-filter=all") == ["fake10\t130577\t131433\t-\t3\t0\t0\t72\t43\t1.58\tct/ac\tsj0"]


def test_canonical_predicate_matches_reference_build():
    """oracle/_ref: the reference's own splice_juncs.cc, compiled where it lies."""
    from tests import oracle_binding
    import itertools
    ref = oracle_binding.load_ref_splice()
    if ref is None:
        pytest.skip("oracle/_ref not built (reference tree absent)")
    lib = oracle_binding.load()
    alphabet = "ACGTNacgtn"
    n = 0
    for a, b, c, d in itertools.product(alphabet, repeat=4):
        s = ("%s%s/%s%s" % (a, b, c, d)).encode()
        assert bool(lib.ipxo_is_canonical(s)) == bool(ref.ref_is_canonical(s)), s
        n += 1
    for s in (b"", b"GT", b"GT/AGG", b"GTAAG", b"gt/ag", b"Gt/aG", b"GT-AG"):
        assert bool(lib.ipxo_is_canonical(s)) == bool(ref.ref_is_canonical(s)), s
    assert n == 10000
    # filter semantics, splice_juncs.cc:52-61
    for flt in (0, 1, 2):
        for s in (b"", b"GT/AG", b"gt/ag", b"ct/ac", b"NN/NN"):
            exp = ref.ref_junction_filter_check(flt, s)
            got = 1 if (s == b"" or flt != 2 or lib.ipxo_is_canonical(s)) else 0
            assert got == exp
