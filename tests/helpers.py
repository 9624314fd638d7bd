"""Shared helpers for the parity tests: SAM text -> SoA batch, table comparison."""
import numpy as np

OPS = "MIDNSHP=XB"


def parse_sam(path):
    """Minimal SAM-text decoder for test fixtures -> (target names, SoA batch, qnames).
    Mirrors the encoding rules of include/ipx.h (xs / nh classes)."""
    names = []
    tid, pos, flag, xs, nh, off, nc, cigar, qn = [], [], [], [], [], [], [], [], []
    with open(path) as f:
        for line in f:
            line = line.rstrip("\n")
            if not line:
                continue
            if line[0] == "@":
                if line.startswith("@SQ\t"):
                    for fld in line.split("\t")[1:]:
                        if fld.startswith("SN:"):
                            names.append(fld[3:])
                            break
                continue
            f_ = line.split("\t")
            qn.append(f_[0])
            flag.append(int(f_[1], 0))
            tid.append(names.index(f_[2]) if f_[2] != "*" and f_[2] in names else -1)
            pos.append(int(f_[3]) - 1)
            off.append(len(cigar))
            k = 0
            if f_[5] != "*":
                num = ""
                for ch in f_[5]:
                    if ch.isdigit():
                        num += ch
                    else:
                        cigar.append((int(num) << 4) | OPS.index(ch))
                        num = ""
                        k += 1
            nc.append(k)
            x, h, have_x, have_h = 0, 0, False, False
            for t in f_[11:]:
                if t.startswith("XS:") and not have_x:
                    have_x = True
                    c = ord(t[5]) if t[3] == "A" and len(t) > 5 else 0
                    x = c if c else ord(".")
                elif t.startswith("NH:") and not have_h:
                    have_h = True
                    v = int(t[5:]) if t[3] == "i" else 0
                    h = 0 if v <= 0 else (1 if v == 1 else 2)
            xs.append(x)
            nh.append(h)
    batch = {
        "tid": np.array(tid, np.int32), "pos": np.array(pos, np.int32), "flag": np.array(flag, np.uint16),
        "xs": np.array(xs, np.uint8), "nh": np.array(nh, np.uint8), "cigar_off": np.array(off, np.uint32),
        "n_cigar": np.array(nc, np.uint32), "cigar": np.array(cigar, np.uint32),
    }
    return names, batch, qn


_COMP = {ord("A"): ord("T"), ord("T"): ord("A"), ord("G"): ord("C"), ord("C"): ord("G")}


def motif_string(strand, motif, fetch=None):
    """splice_sites string of a junction from the table's (final strand, motif word);
    fetch() -> the four raw bytes when the motif holds a non-ACGT base."""
    if not (motif & 0x8000):
        return b""
    if motif & 0x4000:
        raw = fetch()
        g = bytes([raw[0], raw[1], ord("/"), raw[2], raw[3]]).upper()
    else:
        b = [(motif >> 6) & 3, (motif >> 4) & 3, (motif >> 2) & 3, motif & 3]
        g = bytes(["ACGT"[b[0]].encode()[0], "ACGT"[b[1]].encode()[0], ord("/"),
                   "ACGT"[b[2]].encode()[0], "ACGT"[b[3]].encode()[0]])
    if strand == ord("-"):
        g = bytes(_COMP.get(c, c) for c in reversed(g))
        canon = bool(motif & 0x1000)
    else:
        canon = bool(motif & 0x2000)
    return g if canon else g.lower()


def sort_rows(keys, n):
    if n == 0:
        return np.zeros(0, np.int64)
    return np.lexsort(tuple(reversed(keys)))


def compare_tables(dev, ora, genome_fetch=None):
    """Assert that the device table equals the oracle table, field by field, bit for bit.
    dev: extractor.get_junctions(); ora: oracle_binding.Oracle.table()."""
    from intronprospector_amd.extractor import expand_histogram, offsets_per_junction
    nd, no = len(dev["intron_start"]), len(ora["intron_start"])
    assert nd == no, "junction count: device %d, oracle %d" % (nd, no)
    od = sort_rows([dev["target"], dev["intron_start"], dev["intron_end"]], nd)
    oo = sort_rows([ora["target"].astype(np.uint32), ora["intron_start"], ora["intron_end"]], no)
    for k_dev, k_ora in (("target", "target"), ("intron_start", "intron_start"), ("intron_end", "intron_end"),
                         ("anchor_start", "anchor_start"), ("anchor_end", "anchor_end"), ("strand", "strand"),
                         ("ijunc", "ijunc")):
        a = dev[k_dev][od].astype(np.int64)
        b = ora[k_ora][oo].astype(np.int64)
        bad = np.flatnonzero(a != b)
        assert bad.size == 0, "%s differs at %d rows, first row %d: device %d oracle %d (key %s:%d-%d)" % (
            k_dev, bad.size, bad[0], a[bad[0]], b[bad[0]], dev["target"][od][bad[0]],
            dev["intron_start"][od][bad[0]], dev["intron_end"][od][bad[0]])
    assert np.array_equal(dev["counts"][od], ora["counts"][oo]), "read counts differ"
    # offsets multiset per junction
    n_off_dev = offsets_per_junction(dev)
    assert np.array_equal(n_off_dev[od].astype(np.uint64), ora["n_offsets"][oo]), "offset counts differ"
    dev_off = expand_histogram(dev)
    # reorder the concatenated offset lists to the common row order
    def reorder(flat, counts, order):
        starts = np.concatenate([[0], np.cumsum(counts)])[:-1]
        c, st = counts[order], starts[order]
        tot = int(c.sum())
        base = np.repeat(st - (np.cumsum(c) - c), c)
        return flat[(base + np.arange(tot)).astype(np.int64)]
    a = reorder(dev_off, n_off_dev.astype(np.int64), od)
    b = reorder(ora["offsets"], ora["n_offsets"].astype(np.int64), oo)
    assert np.array_equal(a, b), "read offset multisets differ"
    # histogram values strictly ascending inside a junction
    hb = dev["hist_begin"].astype(np.int64)
    for j in range(min(nd, 2000)):
        v = dev["hist_value"][hb[j]:hb[j + 1]]
        assert np.all(v[1:] > v[:-1])
    # splice-site strings (no genome on either side: nothing to spell, and a table of millions of rows need not loop)
    if not np.any(dev["motif"]) and not np.any(ora["splice_sites"]):
        return
    for r in range(nd):
        i, k = od[r], oo[r]
        exp = bytes(ora["splice_sites"][k]).rstrip(b"\0")
        fetch = None
        if genome_fetch is not None:
            fetch = (lambda i=i: genome_fetch(int(dev["target"][i]), int(dev["intron_start"][i]), int(dev["intron_end"][i])))
        got = motif_string(int(dev["strand"][i]), int(dev["motif"][i]), fetch)
        assert got == exp, "splice sites differ for row %d: %r vs %r" % (r, got, exp)


def fuzz_batch(n, seed, min_ops, max_ops, n_targets=2, bad_ops=False):
    """Random CIGARs over every operation the reference knows (MIDNSH=X, optionally P/B): no
    biological structure, so runs like `N N`, `N D N`, `S` in the middle or an `I` at the start
    all occur.  Positions are drawn from a few hundred anchors so that junctions repeat."""
    rng = np.random.default_rng(seed)
    ops_all = np.array([0, 0, 0, 0, 1, 2, 3, 3, 3, 4, 5, 7, 8], np.uint32)
    words, offs, ncs = [], [], []
    total = 0
    anchors = rng.integers(1000, 2_000_000, 300)
    intron_lens = np.array([30, 70, 75, 100, 250, 1000, 40_000, 499_999, 500_001], np.uint32)
    for i in range(n):
        k = int(rng.integers(min_ops, max_ops + 1))
        op = ops_all[rng.integers(0, len(ops_all), k)]
        ln = rng.integers(1, 40, k).astype(np.uint32)
        is_n = op == 3
        ln[is_n] = intron_lens[rng.integers(0, len(intron_lens), int(is_n.sum()))]
        small = (op == 1) | (op == 2)
        ln[small] = rng.integers(1, 6, int(small.sum()))
        if bad_ops and i % 97 == 50:
            op[rng.integers(0, k)] = 6 if i % 2 else 9
        offs.append(total)
        ncs.append(k)
        words.append((ln << np.uint32(4)) | op)
        total += k
    flags = np.array([0, 16, 99, 147, 83, 163, 256, 2048, 1024, 512], np.uint16)
    return {"tid": rng.integers(-1, n_targets, n).astype(np.int32),
            "pos": anchors[rng.integers(0, len(anchors), n)].astype(np.int32),
            "flag": flags[np.minimum(rng.integers(0, 40, n), len(flags) - 1) * (rng.random(n) < 0.4)],
            "xs": np.array([0, ord("+"), ord("-")], np.uint8)[rng.integers(0, 3, n)],
            "nh": rng.integers(0, 3, n).astype(np.uint8),
            "cigar_off": np.array(offs, np.uint32), "n_cigar": np.array(ncs, np.uint32),
            "cigar": np.concatenate(words) if words else np.zeros(0, np.uint32)}
