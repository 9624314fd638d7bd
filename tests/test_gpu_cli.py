"""GPU: the intronProspector-compatible command line of the product, byte for byte
against the reference's golden outputs (same rules as tests/test_oracle_golden.py)."""
import json
import os
import subprocess

import pytest

from tests import bamio
from tests.test_oracle_golden import RULES

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "intronprospector_amd", "bin", "intronProspector")


@pytest.fixture(scope="session")
def cli():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "intronprospector_amd", "host")], check=True,
                   stdout=subprocess.DEVNULL)
    return CLI


def _run(cli, golden, name, outdir, inputs=None, env=None):
    opts, rule_inputs, extra, err_golden = RULES[name]
    gin = os.path.join(golden, "input")
    cmd = [cli] + [o.format(G=gin) for o in opts]
    outs = {"--junction-bed": "juncs.bed", "--intron-bed": "introns.bed", "--intron-calls": "introns.tsv"}
    outs.update(extra)
    for opt, suf in outs.items():
        cmd.append("%s=%s" % (opt, os.path.join(outdir, "%s.%s" % (name, suf))))
    cmd += inputs if inputs else [os.path.join(gin, i) for i in rule_inputs]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env)
    assert r.returncode == 0, r.stdout.decode()
    for suf in outs.values():
        with open(os.path.join(golden, "expected", "%s.%s" % (name, suf)), "rb") as f:
            exp = f.read()
        with open(os.path.join(outdir, "%s.%s" % (name, suf)), "rb") as f:
            assert f.read() == exp, "%s.%s differs from the reference's golden file" % (name, suf)
    if err_golden:
        with open(os.path.join(golden, "expected", err_golden), "rb") as f:
            assert r.stdout == f.read()


@pytest.mark.parametrize("name", sorted(RULES))
def test_cli_reproduces_reference_golden(cli, golden, tmp_path, name):
    _run(cli, golden, name, str(tmp_path))


@pytest.mark.parametrize("name", ["testBasic3", "testBasic4", "testBasic5"])
def test_cli_bam_input(cli, golden, tmp_path, name):
    """the reference runs these rules on a BAM made from test_basic.sam"""
    bam = str(tmp_path / "test_basic.bam")
    bamio.sam_to_bam(os.path.join(golden, "input", "test_basic.sam"), bam)
    _run(cli, golden, name, str(tmp_path), inputs=[bam])


def test_cli_debug_trace_multiple_inputs_and_filter(cli, oracle_cli, golden, tmp_path):
    """--debug-trace beyond the reference's one golden trace: two inputs (one table, a header per
    open), and a filter that drops early records (fail-then-pass), against the oracle's trace."""
    gin = os.path.join(golden, "input")
    cases = [(["--min-confidence-score=0.0", "--unsorted"], ["test_multi.one.sam", "test_multi.two.sam"]),
             (["--min-confidence-score=0.0", "--genome-fasta=" + os.path.join(gin, "fake10_genome.fa"), "--sj-filter=canon"],
              [os.path.join("..", "known", "fail_then_pass.sam")]),
             (["--min-confidence-score=0.0", "--allow-anchor-indels"], ["ont.sam"])]
    for k, (opts, inputs) in enumerate(cases):
        paths = [os.path.join(gin, i) for i in inputs]
        got, exp = str(tmp_path / ("got%d.tsv" % k)), str(tmp_path / ("exp%d.tsv" % k))
        subprocess.run([cli] + opts + ["--debug-trace=" + got] + paths, check=True)
        subprocess.run([oracle_cli] + opts + ["--debug-trace=" + exp] + paths, check=True)
        with open(got, "rb") as f, open(exp, "rb") as g:
            a, b = f.read(), g.read()
        assert a == b and a.count(b"\n") > 1, "case %d" % k


def test_cli_pass_through_and_strand_tags(cli, golden, tmp_path):
    """tests/Makefile.in:120-123,133-136 diff the pass-through against the input; the reference also
    keeps two outputs of the strand-tag options (expected/testStrandTag{Add,Update}.sam, no rule)."""
    gin, gexp = os.path.join(golden, "input"), os.path.join(golden, "expected")
    fa = "--genome-fasta=" + os.path.join(gin, "fake10_genome.fa")
    star, hisat = os.path.join(gin, "fake10.star.sam"), os.path.join(gin, "fake10.hisat2.sam")
    out = str(tmp_path / "pt.sam")
    calls = "--intron-calls=" + str(tmp_path / "calls.tsv")
    subprocess.run([cli, "--min-confidence-score=0.0", fa, calls, "--pass-through=" + out, star], check=True)
    assert open(out, "rb").read() == open(star, "rb").read()
    with open(os.path.join(gexp, "testFake10a.introns.tsv"), "rb") as f:
        assert open(str(tmp_path / "calls.tsv"), "rb").read() == f.read()
    subprocess.run([cli, fa, "--set-XS-strand-tag", "--set-TS-strand-tag", "--pass-through=" + out, star], check=True)
    assert open(out, "rb").read() == open(os.path.join(gexp, "testStrandTagAdd.sam"), "rb").read()
    subprocess.run([cli, fa, "--set-XS-strand-tag", "--pass-through=" + out, hisat], check=True)
    assert open(out, "rb").read() == open(os.path.join(gexp, "testStrandTagUpdate.sam"), "rb").read()
    # BAM in, BAM out; same tags as the SAM run
    bam, bam_out = str(tmp_path / "star.bam"), str(tmp_path / "pt.bam")
    bamio.sam_to_bam(star, bam)
    subprocess.run([cli, fa, "--set-XS-strand-tag", "--pass-through=" + bam_out, bam], check=True)
    hc = os.path.join(os.path.dirname(cli), "ipx_host_check")
    got = subprocess.run([hc, "dump-aln", bam_out], stdout=subprocess.PIPE, check=True).stdout
    sam_out = str(tmp_path / "pt2.sam")
    subprocess.run([cli, fa, "--set-XS-strand-tag", "--pass-through=" + sam_out, star], check=True)
    exp = subprocess.run([hc, "dump-aln", sam_out], stdout=subprocess.PIPE, check=True).stdout
    assert got == exp and b"\t43\t" in got
    # the waiting alignments spilled to their temporary file (IPX_PASS_THROUGH_BUFFER_MB=0: every record goes to disk
    # and is read back in order): the same bytes
    env = dict(os.environ, IPX_PASS_THROUGH_BUFFER_MB="0")
    subprocess.run([cli, fa, "--set-XS-strand-tag", "--set-TS-strand-tag", "--pass-through=" + out, star], check=True, env=env)
    assert open(out, "rb").read() == open(os.path.join(gexp, "testStrandTagAdd.sam"), "rb").read()
    subprocess.run([cli, fa, "--set-XS-strand-tag", "--pass-through=" + bam_out, bam], check=True, env=env)
    assert subprocess.run([hc, "dump-aln", bam_out], stdout=subprocess.PIPE, check=True).stdout == exp
    # the reference's --unsorted loop never writes the alignments (identify_junctions_for_bam,
    # junctions_extractor.cc:678-683): header only
    subprocess.run([cli, "--unsorted", fa, "--pass-through=" + out, star], check=True)
    assert open(out).read() == "".join(l for l in open(star) if l.startswith("@"))
    # option checks, src/intronProspector.cc:269-271,294-298
    r = subprocess.run([cli, "--set-XS-strand-tag", star], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 1 and b"require --pass-through and --genome-fasta" in r.stderr
    r = subprocess.run([cli, "--pass-through=" + out, star, hisat], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 1 and b"can only have one input BAM" in r.stderr


def test_cli_reads_standard_input(cli, golden, tmp_path):
    """No input file = /dev/stdin (src/intronProspector.cc:266-268): SAM through a pipe, BAM through a
    pipe (zlib's gzip reader) and BAM as a redirected file (the parallel BGZF reader)."""
    sam = os.path.join(golden, "input", "test_basic.sam")
    bam = str(tmp_path / "b.bam")
    bamio.sam_to_bam(sam, bam)
    with open(os.path.join(golden, "expected", "testBasic1.introns.tsv"), "rb") as f:
        exp = f.read()
    for src, piped in ((sam, True), (bam, True), (bam, False)):
        out = str(tmp_path / "o.tsv")
        cmd = [cli, "--min-confidence-score=0.0", "--intron-calls=" + out]
        if piped:
            p1 = subprocess.Popen(["cat", src], stdout=subprocess.PIPE)
            subprocess.run(cmd, stdin=p1.stdout, check=True)
            p1.wait()
        else:
            with open(src, "rb") as fi:
                subprocess.run(cmd, stdin=fi, check=True)
        assert open(out, "rb").read() == exp, (src, piped)


def _fuzz_sam(path, seed, n=4000, sorted_by_target=True, mapped_unmapped=False):
    """SAM text with the odd cases of real files: unmapped reads, `*` CIGARs, `=`/`X`, clips, indels,
    NH of every integer type and value, XS of type A and of other types, duplicate tags, long names."""
    import random
    rng = random.Random(seed)
    names = ["chrB", "chrA", "chr10", "chr2"]
    recs = []
    for i in range(n):
        t = rng.randrange(len(names))
        pos = rng.randrange(1, 200000)
        ops = []
        for _ in range(rng.choice([1, 3, 3, 3, 5, 5, 7, 9])):
            ops.append(rng.choice(["M", "M", "M", "N", "N", "I", "D", "=", "X", "S"]))
        cig = "".join("%d%s" % (rng.choice([75, 80, 300, 4000]) if o == "N" else rng.randrange(1, 40), o) for o in ops)
        flag = rng.choice([0, 16, 99, 147, 83, 163, 256, 272, 1024, 512, 2048, 4])
        if rng.random() < 0.03:
            cig = "*"
        tags = []
        r = rng.random()
        if r < 0.5: tags.append("NH:i:%d" % rng.choice([0, 1, 1, 1, 2, 7, 300, 70000]))
        elif r < 0.55: tags.append("NH:Z:2")
        r = rng.random()
        if r < 0.3: tags.append("XS:A:%s" % rng.choice("+-.?"))
        elif r < 0.35: tags.append("XS:i:5")
        if rng.random() < 0.1: tags.append("XS:A:+")                     # a second XS: the first one counts
        if rng.random() < 0.2: tags.insert(0, "MD:Z:%d" % rng.randrange(100))
        # (an unmapped-flagged read that still names a target and a spliced CIGAR makes the reference throw)
        rname = names[t] if flag != 4 or (mapped_unmapped and i > n // 2 and rng.random() < 0.5) else "*"
        recs.append((t if rname != "*" else 99, pos, "read_%d_%s\t%d\t%s\t%d\t%d\t%s\t*\t0\t0\t*\t*%s" % (
            i, "x" * rng.choice([0, 0, 40]), flag, rname, pos if rname != "*" else 0, rng.randrange(61), cig,
            "".join("\t" + x for x in tags))))
    if sorted_by_target:
        recs.sort(key=lambda r: (r[0], r[1]))
    with open(path, "w") as f:
        f.write("@HD\tVN:1.6\tSO:coordinate\n")
        for nm in names:
            f.write("@SQ\tSN:%s\tLN:1000000\n" % nm)
        for r in recs:
            f.write(r[2] + "\n")


@pytest.mark.parametrize("opts", [
    ["--min-confidence-score=0.0"],
    ["--min-confidence-score=0.0", "--allow-anchor-indels", "--max-anchor-indel-size=5", "--min-anchor-length=3"],
    ["--strandness=RF", "--excludes=multi", "--min-confidence-score=0.5"],
    ["--unsorted", "--strandness=FR", "--min-intron-length=76", "--max-intron-length=3999"],
])
def test_cli_against_the_oracle_on_odd_sam_files(cli, oracle_cli, tmp_path, opts):
    """The whole command line (SAM parsing included) against the oracle's, outputs byte for byte."""
    sam = str(tmp_path / "odd.sam")
    _fuzz_sam(sam, seed=len(" ".join(opts)), sorted_by_target="--unsorted" not in opts)
    outs = {}
    for who, exe in (("got", cli), ("exp", oracle_cli)):
        o = {k: str(tmp_path / ("%s.%s" % (who, k))) for k in ("juncs.bed", "introns.bed", "introns.bed6", "introns.tsv", "trace.tsv")}
        r = subprocess.run([exe] + opts + ["--junction-bed=" + o["juncs.bed"], "--intron-bed=" + o["introns.bed"],
                                           "--intron-bed6=" + o["introns.bed6"], "--intron-calls=" + o["introns.tsv"],
                                           "--debug-trace=" + o["trace.tsv"], sam],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        outs[who] = (r.returncode, r.stderr, {k: open(v, "rb").read() for k, v in o.items()})
    assert outs["got"][0] == outs["exp"][0] == 0, (outs["got"][1], outs["exp"][1])
    assert outs["got"][1] == outs["exp"][1]
    for k in outs["exp"][2]:
        assert outs["got"][2][k] == outs["exp"][2][k], k
    assert outs["exp"][2]["trace.tsv"].count(b"\n") > 20
    # the same alignments as BAM (every integer width for NH, non-A XS, `*` CIGARs, unplaced reads)
    bam = str(tmp_path / "odd.bam")
    bamio.sam_to_bam(sam, bam)
    tsv = str(tmp_path / "bam.introns.tsv")
    trace = str(tmp_path / "bam.trace.tsv")
    subprocess.run([cli] + opts + ["--intron-calls=" + tsv, "--debug-trace=" + trace, bam], check=True)
    assert open(tsv, "rb").read() == outs["exp"][2]["introns.tsv"]
    assert open(trace, "rb").read() == outs["exp"][2]["trace.tsv"]
    # without the trace the BAM goes through the block-parallel feeder (host/bam_feeder.cc); with IPX_PARSE_THREADS=0
    # through the record-by-record loop: the same bytes either way
    # (and by default the device decodes the records itself, ipx_push_bam)
    for k, env in enumerate(({"IPX_PARSE_THREADS": "2"}, {"IPX_PARSE_THREADS": "0"}, {"IPX_PARSE_THREADS": "3", "IPX_DECODE_THREADS": "2"},
                             {"IPX_DEVICE_PARSE": "1"}, {"IPX_DEVICE_PARSE": "0"})):
        o = {name: str(tmp_path / ("feed%d.%s" % (k, name))) for name in ("juncs.bed", "introns.bed", "introns.bed6", "introns.tsv")}
        subprocess.run([cli] + opts + ["--junction-bed=" + o["juncs.bed"], "--intron-bed=" + o["introns.bed"],
                                       "--intron-bed6=" + o["introns.bed6"], "--intron-calls=" + o["introns.tsv"], bam],
                       check=True, env=dict(os.environ, **env))
        for name, path in o.items():
            assert open(path, "rb").read() == outs["exp"][2][name], (env, name)
    # get_read_category's logic_error (src/junctions_extractor.cc:73-74): same message, same exit status, and
    # what the earlier targets had already written stays written
    _fuzz_sam(sam, seed=7, sorted_by_target=True, mapped_unmapped=True)
    res = []
    for who, exe in (("got", cli), ("exp", oracle_cli)):
        tsv = str(tmp_path / (who + ".err.tsv"))
        r = subprocess.run([exe] + opts + ["--intron-calls=" + tsv, sam], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        res.append((r.returncode, r.stderr, open(tsv, "rb").read()))
    assert res[0] == res[1] and res[0][0] == 1 and b"should never look at" in res[0][1]
    bamio.sam_to_bam(sam, bam)                     # the same through the feeder
    tsv = str(tmp_path / "bam.err.tsv")
    r = subprocess.run([cli] + opts + ["--intron-calls=" + tsv, bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env=dict(os.environ, IPX_PARSE_THREADS="2"))
    assert (r.returncode, r.stderr, open(tsv, "rb").read()) == res[1]
    r = subprocess.run([cli] + opts + ["--intron-calls=" + tsv, bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env=dict(os.environ, IPX_DEVICE_PARSE="1"))
    assert (r.returncode, r.stderr, open(tsv, "rb").read()) == res[1]


def test_cli_block_parallel_bam_feeder(cli, tmp_path):
    """host/bam_feeder.cc on a BAM of a few hundred BGZF blocks (records straddle most block ends), sorted and
    --unsorted, several GPUs' worth of members included: byte-identical with the record-by-record reader"""
    import hashlib
    bam = str(tmp_path / "many.bam")
    subprocess.run([os.path.join(os.path.dirname(cli), "ipx_host_check"), "make-bam", "300000", bam], check=True)
    digests = set()
    for mode in ([], ["--unsorted"]):
        for env, extra in (({"IPX_PARSE_THREADS": "0"}, []), ({"IPX_PARSE_THREADS": "4"}, []), ({"IPX_PARSE_THREADS": "2"}, ["--devices=0,0"]),
                           ({"IPX_DEVICE_PARSE": "1"}, []), ({"IPX_DEVICE_PARSE": "1", "IPX_DECODE_THREADS": "3"}, [])):   # the last two: records decoded on the device
            tsv, bed = str(tmp_path / "o.tsv"), str(tmp_path / "o.bed")
            subprocess.run([cli, "--min-confidence-score=0.0", "-c", tsv, "-j", bed] + mode + extra + [bam], check=True,
                           env=dict(os.environ, **env))
            h = hashlib.sha256(open(tsv, "rb").read() + open(bed, "rb").read()).hexdigest()
            digests.add((bool(mode), h))
            assert os.path.getsize(tsv) > 10000
    assert len(digests) == 2, digests                   # one digest per mode, whatever the reader



def test_cli_bam_inflated_on_the_device(cli, oracle_cli, golden, tmp_path):
    """IPX_DEVICE_INFLATE=1: the BAM file goes to the device as it is (ipx_decode_bgzf) -- every BAM rule of the
    reference's suite, a fuzzed sorted file target by target and --unsorted, a few hundred blocks; a file whose
    records straddle block ends and damaged files are read the ordinary way and say what the ordinary reader says."""
    import hashlib
    env1 = dict(os.environ, IPX_DEVICE_INFLATE="1", IPX_TIMELINE="1")
    env0 = dict(os.environ, IPX_DEVICE_INFLATE="0")
    bam = str(tmp_path / "test_basic.bam")
    bamio.sam_to_bam(os.path.join(golden, "input", "test_basic.sam"), bam, aligned=True)
    for name in ("testBasic3", "testBasic4", "testBasic5"):
        _run(cli, golden, name, str(tmp_path), inputs=[bam], env=env1)
    # several hundred blocks, sorted and unsorted, against the host reader's output (aligned by the pass-through writer)
    hc = os.path.join(os.path.dirname(cli), "ipx_host_check")
    many, aligned = str(tmp_path / "many.bam"), str(tmp_path / "aligned.bam")
    subprocess.run([hc, "make-bam", "300000", many], check=True)
    subprocess.run([hc, "copy-aln", many, aligned], check=True)
    for mode in ([], ["--unsorted"]):
        digests = []
        for env in (env1, env0):
            tsv, bed = str(tmp_path / "o.tsv"), str(tmp_path / "o.bed")
            r = subprocess.run([cli, "--min-confidence-score=0.0", "-c", tsv, "-j", bed] + mode + [aligned], check=True, env=env,
                               stderr=subprocess.PIPE)
            digests.append(hashlib.sha256(open(tsv, "rb").read() + open(bed, "rb").read()).hexdigest())
            if env is env1:                      # it really took the device path: one push per target (or one in all), no reading time
                tl = json.loads([l for l in r.stderr.decode().splitlines() if l.startswith('{"ipx_timeline"')][-1])["ipx_timeline"]
                assert tl["alignments"] == 300000 and tl["pushes"] <= 9 and tl["files_decoded_on_device"] == 1, tl
        assert digests[0] == digests[1] and os.path.getsize(tsv) > 10000
    # a file above IPX_DEVICE_INFLATE_MAX (megabytes) is not read whole: it goes through the ordinary reader's batches
    # (round 4: the path had no bound, the advisor's finding) -- same bytes, and the timeline shows many pushes
    tsv, bed = str(tmp_path / "o.tsv"), str(tmp_path / "o.bed")
    assert os.path.getsize(aligned) > (2 << 20)
    r = subprocess.run([cli, "--min-confidence-score=0.0", "-c", tsv, "-j", bed, aligned], check=True, stderr=subprocess.PIPE,
                       env=dict(env1, IPX_DEVICE_INFLATE_MAX="1"))
    tl = json.loads([l for l in r.stderr.decode().splitlines() if l.startswith('{"ipx_timeline"')][-1])["ipx_timeline"]
    assert tl["alignments"] == 300000 and tl["files_decoded_on_device"] == 0, tl
    sorted_digest = hashlib.sha256(open(tsv, "rb").read() + open(bed, "rb").read()).hexdigest()
    r0 = subprocess.run([cli, "--min-confidence-score=0.0", "-c", tsv, "-j", bed, aligned], check=True, env=env0, stderr=subprocess.PIPE)
    assert sorted_digest == hashlib.sha256(open(tsv, "rb").read() + open(bed, "rb").read()).hexdigest()
    # fuzzed records (every aux type, odd CIGARs, unplaced reads), sorted: same bytes as the oracle's command line
    sam, fz = str(tmp_path / "fz.sam"), str(tmp_path / "fz.bam")
    _fuzz_sam(sam, seed=5, sorted_by_target=True)
    bamio.sam_to_bam(sam, fz, aligned=True)
    outs = []
    for exe, inp, env in ((cli, fz, env1), (oracle_cli, sam, env0)):
        tsv = str(tmp_path / "fz.tsv")
        subprocess.run([exe, "--min-confidence-score=0.0", "--intron-calls=" + tsv, inp], check=True, env=env)
        outs.append(open(tsv, "rb").read())
    assert outs[0] == outs[1] and len(outs[0]) > 1000
    # records that straddle block ends (make-bam cuts anywhere), a flipped byte, a cut file: the ordinary reader's results
    for path, damage in ((many, None), (aligned, "flip"), (aligned, "cut")):
        src = path
        if damage:
            raw = bytearray(open(path, "rb").read())
            if damage == "flip":
                raw[len(raw) // 2] ^= 0x55
            else:
                raw = raw[:len(raw) // 2]
            src = str(tmp_path / "damaged.bam")
            open(src, "wb").write(raw)
        res = []
        for env in (env1, env0):
            tsv = str(tmp_path / "d.tsv")
            r = subprocess.run([cli, "--min-confidence-score=0.0", "-c", tsv, src], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            err = b"\n".join(l for l in r.stderr.splitlines() if b"ipx_timeline" not in l)
            res.append((r.returncode, err, open(tsv, "rb").read() if os.path.exists(tsv) else b""))
        assert res[0] == res[1], (damage, res[0][:2], res[1][:2])
        assert (res[0][0] == 0) == (damage is None)


def test_cli_device_parse_then_sam_input(cli, golden, tmp_path):
    """two inputs into one table: a BAM whose records the device decodes, then SAM text read record by record"""
    sam = os.path.join(golden, "input", "test_basic.sam")
    bam = str(tmp_path / "first.bam")
    bamio.sam_to_bam(os.path.join(golden, "input", "test_multi.sam"), bam)
    outs = []
    for mode in ("1", "0"):
        tsv = str(tmp_path / ("two%s.tsv" % mode))
        subprocess.run([cli, "--min-confidence-score=0.0", "--intron-calls=" + tsv, bam, sam], check=True,
                       env=dict(os.environ, IPX_DEVICE_PARSE=mode))
        outs.append(open(tsv, "rb").read())
    assert outs[0] == outs[1] and len(outs[0]) > 500


def test_cli_device_parse_refuses_a_damaged_bam(cli, tmp_path):
    """a record whose tag section is cut short: "Error reader BAM record" from the device decoder as from the host's"""
    import gzip
    import struct
    import zlib
    sam = str(tmp_path / "d.sam")
    with open(sam, "w") as f:
        f.write("@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:chr1\tLN:100000\n")
        for i in range(50):
            f.write("r%d\t0\tchr1\t%d\t60\t20M100N30M\t*\t0\t0\t%s\t%s\tNH:i:1\n" % (i, 1000 + i, "A" * 50, "I" * 50))
    bam = str(tmp_path / "d.bam")
    bamio.sam_to_bam(sam, bam)
    # inflate all blocks, damage record 20's NH tag type, write the stream back as one BGZF block + EOF block
    raw = b""
    data = open(bam, "rb").read()
    p = 0
    while p < len(data):
        bsize = struct.unpack("<H", data[p + 16:p + 18])[0] + 1
        raw += zlib.decompress(data[p + 18:p + bsize - 8], -15)
        p += bsize
    l_text = struct.unpack("<i", raw[4:8])[0]
    q = 8 + l_text
    n_ref = struct.unpack("<i", raw[q:q + 4])[0]
    q += 4
    for _ in range(n_ref):
        l_name = struct.unpack("<i", raw[q:q + 4])[0]
        q += 8 + l_name
    for _ in range(20):
        q += 4 + struct.unpack("<i", raw[q:q + 4])[0]
    bs = struct.unpack("<i", raw[q:q + 4])[0]
    tag = raw.rfind(b"NHc", q, q + 4 + bs)               # (bamio writes the smallest integer type)
    assert tag > 0
    raw = raw[:tag + 2] + b"?" + raw[tag + 3:]

    def block(payload):
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        body = c.compress(payload) + c.flush()
        return (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(body) + 25) + body +
                struct.pack("<II", zlib.crc32(payload), len(payload)))
    bad = str(tmp_path / "bad.bam")
    with open(bad, "wb") as f:
        f.write(block(raw) + block(b""))
    outs = []
    for env in ({"IPX_DEVICE_PARSE": "1"}, {"IPX_DEVICE_PARSE": "0"}):
        r = subprocess.run([cli, "-c", str(tmp_path / "x.tsv"), bad], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           env=dict(os.environ, **env))
        outs.append((r.returncode, r.stderr))
    assert outs[0] == outs[1] and outs[0][0] == 1 and b"Error reader BAM record" in outs[0][1], outs


def test_cli_gzip_outputs(cli, golden, tmp_path):
    import gzip
    out = {s: str(tmp_path / ("b1." + s + ".gz")) for s in ("juncs.bed", "introns.bed", "introns.tsv")}
    subprocess.run([cli, "--min-confidence-score=0.0", "--junction-bed=" + out["juncs.bed"],
                    "--intron-bed=" + out["introns.bed"], "--intron-calls=" + out["introns.tsv"],
                    os.path.join(golden, "input", "test_basic.sam")], check=True)
    for suf, path in out.items():
        with open(os.path.join(golden, "expected", "testBasic1." + suf), "rb") as f:
            assert gzip.open(path, "rb").read() == f.read()


def test_cli_out_of_order_targets_error(cli, golden):
    r = subprocess.run([cli, "--allow-anchor-indels", os.path.join(golden, "input", "ont-out-of-order-targets.sam")],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 1
    with open(os.path.join(golden, "expected", "testOntOutOfOrderTargets.err"), "rb") as f:
        assert r.stdout == f.read()


def test_cli_unknown_cigar_op_error(cli, tmp_path):
    sam = tmp_path / "p.sam"
    sam.write_text("@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:chr1\tLN:1000000\n"
                   "r0\t0\tchr1\t100\t60\t20M100N20M\t*\t0\t0\t*\t*\n"
                   "r1\t0\tchr1\t1001\t60\t10M5P100N10M\t*\t0\t0\t*\t*\n")
    r = subprocess.run([cli, "--intron-calls=/dev/stdout", str(sam)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 1
    assert r.stderr.decode() == "Error: Unknown cigar operation 'P'\nUse --help to get usage\n"


def test_cli_confidence_at_scale_against_the_oracle(cli, oracle_cli, tmp_path):
    """`confidence` at scale.  1.2 M coordinate-sorted alignments over 30 000 junction sites
    (hundreds of reads and dozens of distinct read offsets a junction: entropies all over the range) as a BGZF BAM
    through the product's command line -- inflated, decoded, scanned, reduced on the device, confidence from the
    device's histogram on the host -- against the oracle's command line on the same alignments as SAM text: the calls
    file (junctions.cc:54-70 behind `%.3g`, the default 1.0 threshold AND threshold 0), byte for byte."""
    from intronprospector_amd import synth
    batch, names, _ = synth.short_reads(1_200_000, seed=4242, n_targets=4, target_len=30_000_000, n_sites=30_000)
    ops = "MIDNSHP=XB"
    sam = str(tmp_path / "big.sam")
    cig, off, nc = batch["cigar"], batch["cigar_off"], batch["n_cigar"]
    lens, codes = (cig >> 4).tolist(), (cig & 15).tolist()
    with open(sam, "w") as f:
        f.write("@HD\tVN:1.6\tSO:coordinate\n")
        for n in names:
            f.write("@SQ\tSN:%s\tLN:30000000\n" % n)
        tid, pos, flag, xs, nh = (batch[k].tolist() for k in ("tid", "pos", "flag", "xs", "nh"))
        o, c = off.tolist(), nc.tolist()
        for i in range(len(tid)):
            cs = "".join("%d%s" % (lens[k], ops[codes[k]]) for k in range(o[i], o[i] + c[i]))
            tags = []
            if nh[i] == 1:
                tags.append("NH:i:1")
            elif nh[i] == 2:
                tags.append("NH:i:3")
            if xs[i]:
                tags.append("XS:A:%s" % chr(xs[i]))
            f.write("\t".join(["r%d" % i, str(flag[i]), names[tid[i]], str(pos[i] + 1), "60", cs, "*", "0", "0", "*", "*"] + tags) + "\n")
    bam = str(tmp_path / "big.bam")
    bamio.sam_to_bam(sam, bam, aligned=True)
    for opts in (["--min-confidence-score=0.0"], []):
        outs = []
        for exe, inp, env in ((cli, bam, dict(os.environ, IPX_DEVICE_INFLATE="1")), (oracle_cli, sam, None)):
            tsv = str(tmp_path / "big.tsv")
            subprocess.run([exe] + opts + ["--intron-calls=" + tsv, inp], check=True, env=env)
            with open(tsv, "rb") as g:
                outs.append(g.read())
        assert outs[0] == outs[1], opts
        rows = outs[0].decode().splitlines()[1:]
        conf = [float(r.split("\t")[9]) for r in rows]
        assert len(rows) > 20_000 and max(conf) > 5.0 and len(set(conf)) > 200      # real entropies, not a table of zeros
