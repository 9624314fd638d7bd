"""intronProspectorMerge of the MI355X build (CPU tool) against every merge rule
of the reference's tests/Makefile.in:175-217 and its kept outputs without a rule."""
import gzip
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MERGE = os.path.join(ROOT, "intronprospector_amd", "bin", "intronProspectorMerge")


@pytest.fixture(scope="module")
def merge():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "intronprospector_amd", "host"),
                    "../bin/intronProspectorMerge"], check=True, stdout=subprocess.DEVNULL)
    return MERGE


E = "expected/"
RULES = {
    "testBasic134Merge": ([], [E + "testBasic1.introns.tsv", E + "testBasic3.introns.tsv", E + "testBasic4.introns.tsv"], True),
    "testBasic134LstMerge": (["--intron-calls-files=input/basic134.files.txt"], [], True),
    "testFake10abMerge": ([], [E + "testFake10a.introns.tsv", E + "testFake10b.introns.tsv"], False),
    "testStrandConflictMerge": ([], ["input/strand-confict/SRR62253%d.calls.tsv" % k for k in (60, 62, 61, 63)], False),
    "testSjFilterCanonMerge": (["--sj-filter=canon"], [E + "testSjFilterAll.introns.tsv"], True),
    "testSjFilterAllMerge": (["--sj-filter=all"], [E + "testSjFilterAll.introns.tsv"], True),
    "testSjFilterDefaultMerge": ([], [E + "testSjFilterAll.introns.tsv"], True),
    "testStrand12Merge": ([], ["input/merge1.calls.tsv", "input/merge2.calls.tsv"], False),
    "testStrand21Merge": ([], ["input/merge2.calls.tsv", "input/merge1.calls.tsv"], False),
}


@pytest.mark.parametrize("name", sorted(RULES))
def test_merge_reproduces_reference_golden(merge, golden, tmp_path, name):
    opts, inputs, beds = RULES[name]
    outs = {"--intron-calls": "introns.tsv"}
    if beds:
        outs.update({"--junction-bed": "juncs.bed", "--intron-bed": "introns.bed"})
    cmd = [merge] + opts + ["%s=%s" % (o, tmp_path / s) for o, s in outs.items()] + inputs
    r = subprocess.run(cmd, cwd=golden, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()
    for s in outs.values():
        with open(os.path.join(golden, "expected", "%s.%s" % (name, s)), "rb") as f:
            assert open(str(tmp_path / s), "rb").read() == f.read(), s
    err = os.path.join(golden, "expected", name + ".err")
    if os.path.exists(err):
        assert r.stderr == open(err, "rb").read()


def test_merge_gzip_in_and_out(merge, golden, tmp_path):
    ins = []
    for k in (1, 3, 4):
        dst = str(tmp_path / ("b%d.introns.tsv.gz" % k))
        with open(os.path.join(golden, "expected", "testBasic%d.introns.tsv" % k), "rb") as fi, gzip.open(dst, "wb") as fo:
            shutil.copyfileobj(fi, fo)
        ins.append(dst)
    outs = {"--junction-bed": "juncs.bed", "--intron-bed": "introns.bed", "--intron-calls": "introns.tsv",
            "--intron-bed6": "introns.bed6"}
    subprocess.run([merge] + ["%s=%s.gz" % (o, tmp_path / s) for o, s in outs.items()] + ins, check=True)
    for s in ("juncs.bed", "introns.bed", "introns.tsv"):
        with open(os.path.join(golden, "expected", "testBasic134Merge." + s), "rb") as f:
            assert gzip.open(str(tmp_path / s) + ".gz", "rb").read() == f.read()
    bed6 = gzip.open(str(tmp_path / "introns.bed6.gz"), "rt").read().splitlines()
    bed9 = open(os.path.join(golden, "expected", "testBasic134Merge.introns.bed")).read().splitlines()
    assert bed6 == ["\t".join(l.split("\t")[:6]) for l in bed9]


def test_merge_errors(merge, golden, tmp_path):
    r = subprocess.run([merge], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 1 and b"a least one input call TSV required" in r.stderr
    bad = tmp_path / "bad.tsv"
    bad.write_text("chrom\tintron_start\nchr1\t5\t7\n")
    r = subprocess.run([merge, "--intron-calls=/dev/null", str(bad)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 1 and r.stderr.decode().startswith("Error: expected 2 columns, got 2:")
    r = subprocess.run([merge, "--sj-filter=none", str(bad)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 1 and b"invalid splice junction filter 'none', expected 'all' or 'canon'" in r.stderr
    empty = tmp_path / "empty.tsv"
    empty.write_text("")
    r = subprocess.run([merge, "--intron-calls=/dev/null", str(empty)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 1 and b"empty TSV file" in r.stderr
