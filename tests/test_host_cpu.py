"""CPU-only tests of the host side of the product (no GPU, no compute through libipx):
the C ABI library loads and exports every declared symbol, alignment decoding (SAM, gzip'd
SAM, BAM), and ordering + confidence + writers against the oracle and the reference's
golden files."""
import ctypes
import gzip
import json
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

from tests import bamio, helpers

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST_CHECK = os.path.join(ROOT, "intronprospector_amd", "bin", "ipx_host_check")


@pytest.fixture(scope="session")
def host_check():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "intronprospector_amd", "host"), "../bin/ipx_host_check"],
                   check=True, stdout=subprocess.DEVNULL)
    return HOST_CHECK


def test_abi_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "ipx.h")).read()
    declared = set(re.findall(r"^(?:int|uint32_t)\s+(ipx_\w+)\s*\(", hdr, re.M))
    from intronprospector_amd import _lib
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), name
    assert lib.ipx_abi_version() == 5


def test_the_per_operation_scan_is_a_library_of_its_own():
    """round 4: k_scan_ops is test-only and lives in libipx_scanops.so, which libipx.so loads on IPX_SCAN=ops; the product
    library holds nothing of it"""
    from intronprospector_amd import _lib
    ops = os.path.join(os.path.dirname(_lib.LIB_PATH), "libipx_scanops.so")
    assert os.path.exists(ops), "make -C intronprospector_amd/csrc builds it"
    assert hasattr(ctypes.CDLL(ops), "ipx_scanops_launch")
    names = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], check=True, stdout=subprocess.PIPE).stdout.decode()
    assert "scan_ops" not in names and "scanops" not in names


def test_product_fails_loudly_without_a_device():
    """No CPU fallback: without a GPU the product path raises instead of computing."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import intronprospector_amd as ipa
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ipa.JunctionsExtractor()
    exe = os.path.join(ROOT, "intronprospector_amd", "bin", "intronProspector")
    if os.path.exists(exe):
        r = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "input", "ont.sam")],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 1 and b"no CPU path" in r.stderr


def _dump(host_check, path):
    r = subprocess.run([host_check, "dump-aln", path], stdout=subprocess.PIPE, check=True)
    names, rows = [], []
    for line in r.stdout.decode().splitlines():
        f = line.split("\t")
        if f[0] == "@SQ":
            names.append(f[1])
        else:
            rows.append((int(f[0]), int(f[1]), int(f[2]), int(f[3]), int(f[4]),
                         [int(x) for x in f[6].split(",")] if f[6] else []))
    return names, rows


@pytest.mark.parametrize("sam", ["test_basic.sam", "test_multi.sam", "ont.sam", "fake10.hisat2.sam",
                                 "fake10.missing.sam", "ont-out-of-order-targets.sam"])
def test_alignment_decoding_sam_gz_bam(host_check, golden, tmp_path, sam):
    path = os.path.join(golden, "input", sam)
    names, batch, _ = helpers.parse_sam(path)
    exp = []
    for i in range(len(batch["tid"])):
        o, n = int(batch["cigar_off"][i]), int(batch["n_cigar"][i])
        exp.append((int(batch["tid"][i]), int(batch["pos"][i]), int(batch["flag"][i]), int(batch["xs"][i]),
                    int(batch["nh"][i]), batch["cigar"][o:o + n].tolist()))
    got_names, got = _dump(host_check, path)
    assert got_names == names and got == exp
    gz = str(tmp_path / "x.sam.gz")
    with open(path, "rb") as fi, gzip.open(gz, "wb") as fo:
        shutil.copyfileobj(fi, fo)
    assert _dump(host_check, gz) == (names, exp)
    bam = str(tmp_path / "x.bam")
    bamio.sam_to_bam(path, bam)
    assert _dump(host_check, bam) == (names, exp)


def test_bam_long_cigar_in_cg_tag(host_check, tmp_path):
    sam = tmp_path / "long.sam"
    cig = "".join("%dM%dN" % (3 + (i % 5), 100 + i) for i in range(40)) + "7M"
    sam.write_text("@HD\tVN:1.6\n@SQ\tSN:chr1\tLN:100000000\nr1\t0\tchr1\t101\t60\t%s\t*\t0\t0\t%s\t*\tNH:i:1\n"
                   % (cig, "A" * (sum(3 + (i % 5) for i in range(40)) + 7)))
    bam = str(tmp_path / "long.bam")
    bamio.sam_to_bam(str(sam), bam, long_cigar_threshold=10)      # force the CG:B:I path
    assert _dump(host_check, bam) == _dump(host_check, str(sam))


def test_bgzf_blocks_inflate_in_parallel_in_file_order(host_check, tmp_path):
    """host/bgzf_reader.cc: many blocks, 1 / 3 / 8 inflate threads give the same
    records in the same order; damaged and truncated blocks are reported, not skipped."""
    bam = str(tmp_path / "many.bam")
    subprocess.run([host_check, "make-bam", "60000", bam], check=True)          # ~120 BGZF blocks
    outs = []
    for t in ("1", "3", "8", "12", None):             # None: the default thread count (what the command line uses)
        env = {k: v for k, v in os.environ.items() if k != "IPX_DECODE_THREADS"}
        if t:
            env["IPX_DECODE_THREADS"] = t
        r = subprocess.run([host_check, "dump-aln", bam], stdout=subprocess.PIPE, check=True, env=env)
        outs.append(r.stdout)
    assert all(o == outs[0] for o in outs)
    # zlib for every block instead of host/fast_inflate.cc: the same records
    r = subprocess.run([host_check, "dump-aln", bam], stdout=subprocess.PIPE, check=True, env=dict(os.environ, IPX_INFLATE="zlib"))
    assert r.stdout == outs[0]
    # the benchmark's checksum (what profiles/*_bam_decode_threads.jsonl records) is a function of the file alone
    sums = set()
    for t in ("1", "12"):
        r = subprocess.run([host_check, "bench-decode", bam, t], stdout=subprocess.PIPE, check=True)
        sums.add(json.loads(r.stdout)["checksum"])
    assert len(sums) == 1
    lines = outs[0].decode().splitlines()
    assert len(lines) == 60001 and lines[0].startswith("@SQ\tchr1")
    first = lines[1].split("\t")
    assert first[0] == "0" and first[5] == "3" and all(int(w) & 15 in (0, 3) for w in first[6].split(","))
    r = subprocess.run([host_check, "bench-decode", bam, "4"], stdout=subprocess.PIPE, check=True)
    rep = json.loads(r.stdout)
    assert rep["records"] == 60000 and rep["cigar_words"] == 180000 and rep["threads"] == 4
    # without a request: three quarters of the CPUs the cgroup allows, at most 12; the inflate-only
    # benchmark sees every byte the records came from
    env = {k: v for k, v in os.environ.items() if k != "IPX_DECODE_THREADS"}
    r = subprocess.run([host_check, "bench-decode", bam], stdout=subprocess.PIPE, check=True, env=env)
    assert 1 <= json.loads(r.stdout)["threads"] <= 12
    r = subprocess.run([host_check, "bench-inflate", bam, "2"], stdout=subprocess.PIPE, check=True)
    inf = json.loads(r.stdout)
    assert inf["threads"] == 2 and inf["inflated_MB"] * 1e6 > 60000 * 180
    raw = bytearray(open(bam, "rb").read())
    bad = str(tmp_path / "bad.bam")
    flipped = bytearray(raw)
    flipped[len(raw) // 2] ^= 0x55                                                  # inside some block's payload
    open(bad, "wb").write(flipped)
    r = subprocess.run([host_check, "dump-aln", bad], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 1 and b"Error reader BAM record" in r.stderr
    open(bad, "wb").write(raw[:len(raw) // 2])                                     # cut inside a block
    r = subprocess.run([host_check, "dump-aln", bad], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 1 and b"Error reader BAM record" in r.stderr


def test_block_parallel_bam_feeder_equals_the_record_reader(host_check, tmp_path):
    """host/bam_feeder.cc: blocks split on one thread, parsed on a pool -- the same records in the same order as
    AlnReader::read, whole-file and target by target, with batches so small that spans are cut everywhere, on odd
    records (every aux type, `*` CIGARs, unplaced reads), on a few hundred blocks whose ends cut through records,
    and on a long CIGAR kept in the CG tag"""
    from tests.test_gpu_cli import _fuzz_sam

    def dump(args):
        r = subprocess.run([host_check] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0, r.stderr
        return [l for l in r.stdout.splitlines() if not l.startswith(b"#")]
    sam, bam = str(tmp_path / "odd.sam"), str(tmp_path / "odd.bam")
    for seed in (3, 7):
        _fuzz_sam(sam, seed=seed, sorted_by_target=True, mapped_unmapped=seed == 7)
        bamio.sam_to_bam(sam, bam)
        ref = dump(["dump-aln", bam])
        assert dump(["dump-feed", bam, "all"]) == ref
        assert dump(["dump-feed", bam, "all", "37", "150"]) == ref
        by_target = dump(["dump-feed", bam, "by-target", "50", "300"])        # stops where the sorted-mode loop stops
        assert 3000 < len(by_target) <= len(ref) and by_target == ref[:len(by_target)]
    many = str(tmp_path / "many.bam")
    subprocess.run([host_check, "make-bam", "120000", many], check=True)
    ref = dump(["dump-aln", many])
    for args in (["all"], ["all", "777", "2500"], ["by-target", "100000", "800000"]):
        assert dump(["dump-feed", many] + args) == ref, args
    long_sam = tmp_path / "long.sam"
    cig = "".join("%dM%dN" % (3 + (i % 5), 100 + i) for i in range(40)) + "7M"
    seq = "A" * (sum(3 + (i % 5) for i in range(40)) + 7)
    long_sam.write_text("@HD\tVN:1.6\n@SQ\tSN:chr1\tLN:100000000\n" +
                        "".join("r%d\t0\tchr1\t%d\t60\t%s\t*\t0\t0\t%s\t*\tNH:i:1\n" % (k, 101 + k, cig if k % 2 else "20M100N20M", seq if k % 2 else "A" * 40)
                                for k in range(50)))
    long_bam = str(tmp_path / "long.bam")
    bamio.sam_to_bam(str(long_sam), long_bam, long_cigar_threshold=10)       # odd records: the CG:B:I path
    ref = dump(["dump-aln", long_bam])
    assert dump(["dump-feed", long_bam, "all", "7", "90"]) == ref and len(ref) == 51


def test_pass_through_writer_and_strand_tag_edits(host_check, golden, tmp_path):
    """host/pass_through.cc: SAM and BAM pass-through keep every record as it came;
    update_strand_tag semantics (src/junctions_extractor.cc:98-111): the first tag of that name goes,
    the new one (type A) is appended."""
    src = os.path.join(golden, "input", "fake10.hisat2.sam")          # carries XS:A tags in the middle
    out = str(tmp_path / "copy.sam")
    subprocess.run([host_check, "copy-aln", src, out], check=True)
    assert open(out, "rb").read() == open(src, "rb").read()
    subprocess.run([host_check, "copy-aln", src, out, "XS", "-", "TS", "-"], check=True)
    exp = []
    for line in open(src):
        line = line.rstrip("\n")
        if not line.startswith("@"):
            f = line.split("\t")
            tags = f[11:]
            for k, t in enumerate(tags):
                if t.startswith("XS:"):
                    del tags[k]
                    break
            line = "\t".join(f[:11] + tags + ["XS:A:-", "TS:A:-"])
        exp.append(line + "\n")
    assert open(out).read() == "".join(exp)
    # BAM: same records after the round trip; tags readable by the decoder; CG-tag CIGARs survive
    bam, bam2, bam3 = (str(tmp_path / n) for n in ("in.bam", "copy.bam", "tag.bam"))
    bamio.sam_to_bam(src, bam)
    subprocess.run([host_check, "copy-aln", bam, bam2], check=True)
    assert _dump(host_check, bam2) == _dump(host_check, bam) == _dump(host_check, src)
    subprocess.run([host_check, "copy-aln", bam, bam3, "XS", "+"], check=True)
    names, rows = _dump(host_check, bam3)
    ref_names, ref_rows = _dump(host_check, bam)
    assert names == ref_names and len(rows) == len(ref_rows)
    assert all(r[3] == ord("+") for r in rows)
    assert [r[:3] + r[4:] for r in rows] == [r[:3] + r[4:] for r in ref_rows]
    # the copied BAM is a well-formed BGZF file to an independent reader
    import gzip
    assert gzip.open(bam2, "rb").read() == gzip.open(bam, "rb").read()
    assert open(bam2, "rb").read()[-28:] == bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def _table_text(t, names):
    """oracle table (std::map order) -> the text form ipx_host_check print-table reads."""
    lines = []
    starts = np.concatenate([[0], np.cumsum(t["n_offsets"].astype(np.int64))])
    for i in range(len(t["intron_start"])):
        offs = t["offsets"][starts[i]:starts[i + 1]]
        _, cnt = np.unique(offs, return_counts=True)
        ss = bytes(t["splice_sites"][i]).rstrip(b"\0").decode() or "-"
        lines.append(" ".join(map(str, [names[t["target"][i]], t["intron_start"][i], t["intron_end"][i],
                                        t["anchor_start"][i], t["anchor_end"][i], chr(t["strand"][i]), ss,
                                        t["ijunc"][i], *t["counts"][i].tolist(), len(cnt), *cnt.tolist()])))
    return "\n".join(lines) + "\n"


@pytest.mark.parametrize("name,sam,opts,genome,minconf", [
    ("testBasic1", "test_basic.sam", {}, None, 0.0),
    ("testBasic4", "test_basic.sam", {}, None, 1.0),
    ("testBasic5", "test_basic.sam", {"excludes": 1}, None, 0.0),
    ("testFake10a", "fake10.star.sam", {"junction_filter": 2}, "fake10_genome.fa", 0.0),
    ("testSjFilterAll", "fake10.star.sam", {"junction_filter": 1}, "fake10_scramble_genome.fa", 0.0),
    ("testOntIndel", "ont.sam", {"max_anchor_indel_size": 36}, None, 0.0),
])
def test_writers_confidence_and_ordering_reproduce_reference_golden(host_check, oracle_lib, golden, tmp_path,
                                                                    name, sam, opts, genome, minconf):
    """Single-target inputs: the oracle's table, printed by the PRODUCT's host writers, must be
    byte-identical to the reference's golden files (ordering, BED/TSV formats, float confidence)."""
    from tests.oracle_binding import Oracle
    names, batch, _ = helpers.parse_sam(os.path.join(golden, "input", sam))
    ora = Oracle(targets=names, genome_fasta=os.path.join(golden, "input", genome) if genome else None, **opts)
    ora.process(batch)
    t = ora.table()
    ora.close()
    prefix = str(tmp_path / name)
    subprocess.run([host_check, "print-table", prefix, str(minconf)], input=_table_text(t, names).encode(), check=True)
    for suf in ("juncs.bed", "introns.bed", "introns.tsv") + (("introns.bed6",) if name == "testBasic1" else ()):
        with open(os.path.join(golden, "expected", "%s.%s" % (name, suf)), "rb") as f:
            exp = f.read()
        with open("%s.%s" % (prefix, suf), "rb") as f:
            assert f.read() == exp, suf


def test_confidence_matches_oracle_on_random_histograms(host_check, tmp_path):
    """float32 Shannon index: product host code vs the oracle's C restatement, bit for bit,
    through the TSV's 3-significant-digit print and on junctions with many offsets."""
    from tests.oracle_binding import Oracle
    rng = np.random.default_rng(3)
    n = 400
    # build reads directly: junction k gets reads at random offsets -> oracle computes confidence
    tid, pos, cig, off, nc = [], [], [], [], []
    for k in range(n):
        m = int(rng.integers(1, 300))
        offs = rng.integers(8, 8 + int(rng.integers(1, 90)), m)
        for o in offs:
            start = 1000 + k * 5000
            tid.append(0); pos.append(start - int(o)); off.append(len(cig)); nc.append(3)
            cig += [(int(o) << 4) | 0, (700 << 4) | 3, (50 << 4) | 0]
    batch = {"tid": np.array(tid, np.int32), "pos": np.array(pos, np.int32), "flag": np.zeros(len(tid), np.uint16),
             "xs": np.zeros(len(tid), np.uint8), "nh": np.ones(len(tid), np.uint8),
             "cigar_off": np.array(off, np.uint32), "n_cigar": np.array(nc, np.uint32), "cigar": np.array(cig, np.uint32)}
    ora = Oracle(targets=["chr1"])
    ora.process(batch)
    t = ora.table()
    ora.close()
    prefix = str(tmp_path / "conf")
    subprocess.run([host_check, "print-table", prefix, "0"], input=_table_text(t, ["chr1"]).encode(), check=True)
    rows = open(prefix + ".introns.tsv").read().splitlines()[1:]
    assert len(rows) == n
    order = np.lexsort((t["intron_end"], t["intron_start"]))
    for r, i in zip(rows, order):
        assert r.split("\t")[9] == "%.3g" % float(t["confidence"][i])


def test_fast_inflate_against_zlib(host_check, tmp_path):
    """host/fast_inflate.cc, the whole-block DEFLATE decoder behind the BGZF reader: byte-identical with zlib on
    streams of every block type, level and strategy (stored, fixed, dynamic, flush points, tiny windows, empty
    input), its CRC-32 equal to zlib's; and on damaged streams it stays inside its buffers and never accepts
    what zlib inflates differently (ipx_host_check check-inflate)."""
    import struct
    import zlib
    import numpy as np
    rng = np.random.default_rng(77)
    path = str(tmp_path / "streams.bin")
    n = 0
    with open(path, "wb") as out:
        def put(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, wbits=-15, memlevel=8, flushes=0):
            nonlocal n
            c = zlib.compressobj(level, zlib.DEFLATED, wbits, memlevel, strategy)
            comp = b""
            if flushes and len(data) > flushes:
                step = len(data) // flushes
                for k, i in enumerate(range(0, len(data), step)):
                    comp += c.compress(data[i:i + step]) + c.flush(zlib.Z_FULL_FLUSH if k % 2 else zlib.Z_SYNC_FLUSH)
            else:
                comp += c.compress(data)
            comp += c.flush()
            out.write(struct.pack("<II", len(comp), len(data)) + comp)
            n += 1

        def bamlike(k):
            return b"".join(struct.pack("<iiBBHHHIiii", int(rng.integers(0, 8)), 1000 + 3 * i, 10, 60, 4680, 3, 0, 100, -1, -1, 0) +
                            b"read%05d\0" % i + struct.pack("<III", 40 << 4, (int(rng.integers(70, 5000)) << 4) | 3, 60 << 4) +
                            bytes(rng.integers(0, 256, 50, dtype=np.uint8)) + bytes(rng.integers(30, 42, 100, dtype=np.uint8)) + b"NHc\x01"
                            for i in range(k))
        fixed = [b"", b"a", b"ab" * 3, b"\0" * 70000, bytes(rng.integers(0, 256, 65280, dtype=np.uint8)),
                 bytes(rng.integers(0, 4, 65280, dtype=np.uint8)), bamlike(280)[:65280], (b"ACGT" * 20000)[:65280],
                 b"".join(bytes([int(x)]) * int(k) for x, k in zip(rng.integers(0, 256, 2000), rng.integers(1, 300, 2000)))[:65280],
                 open(__file__, "rb").read()]
        for d in fixed:
            for lvl in (0, 1, 6, 9):
                put(d, lvl)
            for st in (zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED):
                put(d, 6, st)
            put(d, 6, flushes=7)
            put(d, 9, memlevel=1)
            put(d, 1, wbits=-9)
        for _ in range(300):
            k, mode = int(rng.integers(0, 70000)), int(rng.integers(0, 4))
            if mode == 0:
                d = bytes(rng.integers(0, 256, k, dtype=np.uint8))
            elif mode == 1:
                d = bytes(rng.integers(0, int(rng.integers(1, 20)), k, dtype=np.uint8))
            elif mode == 2:
                d = bamlike(max(1, k // 230))[:65280]
            else:
                base = bytes(rng.integers(0, 256, int(rng.integers(1, 400)), dtype=np.uint8))
                d = (base * (k // len(base) + 1))[:k]
            put(d, int(rng.integers(0, 10)), [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_RLE, zlib.Z_HUFFMAN_ONLY][int(rng.integers(0, 4))],
                memlevel=int(rng.integers(1, 10)))
    r = subprocess.run([host_check, "check-inflate", path, "8"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()
    rep = json.loads(r.stdout)
    assert rep["streams"] == n and rep["damaged"] >= 8 * (n - 20) and 0 < rep["damaged_still_valid"] < rep["damaged"]


def test_damaged_bam_under_address_sanitizer(tmp_path):
    """The reader's failure paths under AddressSanitizer (CPU build of the host sources): a BAM whose LATE block is
    damaged, read by the block-parallel feeder with two parse threads -- the exception unwinds while spans are still
    dispatched, and the workers must be stopped before the arrays they fill go away (the extractor's destructor and
    ipx_host_check's dump-feed keep that order); and a record whose last tag is cut short by its block_size -- the
    parser checks a value's size against the record before it reads the value."""
    import struct
    host = os.path.join(ROOT, "intronprospector_amd", "host")
    exe = str(tmp_path / "ipx_host_check_asan")
    srcs = [os.path.join(host, f) for f in ("host_check.cc", "junctions.cc", "aln_reader.cc", "bgzf_reader.cc", "fast_inflate.cc",
                                            "bam_feeder.cc", "pass_through.cc")]
    subprocess.run(["g++", "-O1", "-g", "-std=c++14", "-pthread", "-fsanitize=address", "-fno-omit-frame-pointer", "-o", exe] + srcs + ["-lz"],
                   check=True)
    env = dict(os.environ, IPX_PARSE_THREADS="2", ASAN_OPTIONS="detect_leaks=0")
    good = str(tmp_path / "many.bam")
    subprocess.run([exe, "make-bam", "120000", good], check=True, env=env)
    raw = bytearray(open(good, "rb").read())
    r = subprocess.run([exe, "dump-feed", good, "all", "5000", "20000"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    assert r.returncode == 0 and b"AddressSanitizer" not in r.stderr
    for where in (0.6, 0.8, 0.97):
        bad = str(tmp_path / "late.bam")
        flipped = bytearray(raw)
        k = int(len(raw) * where)
        for j in range(k, k + 40):
            flipped[j] ^= 0x5a
        open(bad, "wb").write(flipped)
        for args in (["all", "5000", "20000"], ["by-target", "100000", "800000"]):
            r = subprocess.run([exe, "dump-feed", bad] + args, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=env)
            assert r.returncode != 0 and b"AddressSanitizer" not in r.stderr, r.stderr[-2000:]
            assert b"Error reader BAM record" in r.stderr or b"BGZF" in r.stderr or b"inflate" in r.stderr, r.stderr[-500:]
    # one record, its NH:i tag's value cut off: block_size says the record ends two bytes into the value
    name = b"r1\0"
    cigar = struct.pack("<III", 40 << 4, (500 << 4) | 3, 60 << 4)
    core = struct.pack("<iiBBHHHIiii", 0, 1000, len(name), 60, 4680, 3, 0, 100, -1, -1, 0)
    body = core + name + cigar + bytes(50) + bytes([30]) * 100 + b"NHi\x01\x00"          # 4-byte value, two bytes present
    rec = struct.pack("<I", len(body)) + body
    hdr_text = b"@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:chr1\tLN:1000000\n"
    header = b"BAM\1" + struct.pack("<I", len(hdr_text)) + hdr_text + struct.pack("<I", 1) + struct.pack("<I", 5) + b"chr1\0" + struct.pack("<I", 1000000)
    cut = str(tmp_path / "cut.bam")
    with open(cut, "wb") as f:
        f.write(bamio._bgzf_block(header) + bamio._bgzf_block(rec) + bamio._bgzf_block(b""))
    for cmd in (["dump-aln", cut], ["dump-feed", cut, "all"]):
        r = subprocess.run([exe] + cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=env)
        assert r.returncode != 0 and b"Error reader BAM record" in r.stderr and b"AddressSanitizer" not in r.stderr, r.stderr[-2000:]
