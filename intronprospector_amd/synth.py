"""Deterministic synthetic alignment batches.

The generators produce the 20-byte-per-alignment SoA batch of include/ipx.h
directly (no SAM text), coordinate sorted, from a fixed seed:

  short_reads  config S / G: 100 bp reads, one or two N operations
  long_reads   config L: ONT-style reads, 20-30 N operations, ~1000 CIGAR ops

They are plain numpy; nothing here touches the GPU or the oracle.
"""
import numpy as np

# BAM CIGAR op codes, "MIDNSHP=XB"
OP_M, OP_I, OP_D, OP_N, OP_S, OP_H, OP_P, OP_EQ, OP_X, OP_B = range(10)

SEED_S = 20251024
SEED_L = 20251025
SEED_M0 = 20251100


def _pack(length, op):
    return (np.asarray(length, np.uint32) << np.uint32(4)) | np.uint32(op)


def _log_uniform(rng, lo, hi, n):
    return np.exp(rng.uniform(np.log(lo), np.log(hi), n)).astype(np.int64)


def _intron_lengths(rng, n):
    """log-uniform 70..1e5, 0.5 % below 70 and 0.5 % above 500000 (QC rejects)."""
    ln = _log_uniform(rng, 70, 100000, n)
    u = rng.random(n)
    short = u < 0.005
    long_ = u > 0.995
    ln[short] = rng.integers(20, 70, int(short.sum()))
    ln[long_] = rng.integers(500001, 600000, int(long_.sum()))
    return ln


def _zipf_choice(rng, n_items, n, s=1.1):
    w = 1.0 / np.arange(1, n_items + 1, dtype=np.float64) ** s
    cdf = np.cumsum(w)
    cdf /= cdf[-1]
    ranks = np.searchsorted(cdf, rng.random(n), side="right")
    perm = rng.permutation(n_items)          # rank -> site
    return perm[np.minimum(ranks, n_items - 1)]


def _flatten(ops, valid, tid, pos, flag, xs, nh):
    order = np.lexsort((pos, tid))
    ops, valid = ops[order], valid[order]
    n_cigar = valid.sum(1).astype(np.uint32)
    cigar = ops[valid]
    off = np.zeros(len(n_cigar), np.uint64)
    np.cumsum(n_cigar[:-1], out=off[1:])
    assert cigar.size < 2 ** 32
    return {
        "tid": tid[order].astype(np.int32), "pos": pos[order].astype(np.int32),
        "flag": flag[order].astype(np.uint16), "xs": xs[order].astype(np.uint8),
        "nh": nh[order].astype(np.uint8), "cigar_off": off.astype(np.uint32),
        "n_cigar": n_cigar, "cigar": np.ascontiguousarray(cigar, np.uint32),
    }


class Genome:
    """Synthetic genome: per target a uint8 array of ASCII bases."""

    def __init__(self, names, seqs):
        self.names, self.seqs = list(names), list(seqs)

    def write_fasta(self, path, width=60):
        with open(path, "wb") as fa, open(path + ".fai", "w") as fai:
            off = 0
            for name, seq in zip(self.names, self.seqs):
                hdr = (">%s\n" % name).encode()
                fa.write(hdr)
                off += len(hdr)
                n = len(seq)
                fai.write("%s\t%d\t%d\t%d\t%d\n" % (name, n, off, width, width + 1))
                full = n // width
                body = seq[:full * width].reshape(full, width)
                out = np.empty((full, width + 1), np.uint8)
                out[:, :width] = body
                out[:, width] = 10
                fa.write(out.tobytes())
                off += full * (width + 1)
                if n % width:
                    fa.write(seq[full * width:].tobytes() + b"\n")
                    off += n % width + 1

    def twobit(self):
        """(twobit bytes, mask bytes) per sequence, the layout ipx_set_genome_2bit takes."""
        return [pack_2bit(s) for s in self.seqs]


_CODE = np.full(256, 4, np.uint8)
for _c, _v in zip(b"ACGTacgt", (0, 1, 2, 3, 0, 1, 2, 3)):
    _CODE[_c] = _v


def pack_2bit(seq):
    """ASCII bases -> (2-bit codes, 4 per byte, first base in the low bits;
    non-ACGT mask, 8 per byte, first base in bit 0)."""
    seq = np.asarray(seq, np.uint8)
    n = len(seq)
    code = _CODE[seq]
    non = (code == 4)
    code = np.where(non, 0, code).astype(np.uint8)
    pad = (-n) % 4
    c = np.concatenate([code, np.zeros(pad, np.uint8)]).reshape(-1, 4)
    two = (c[:, 0] | (c[:, 1] << 2) | (c[:, 2] << 4) | (c[:, 3] << 6)).astype(np.uint8)
    mask = np.packbits(non, bitorder="little")
    return np.ascontiguousarray(two), np.ascontiguousarray(mask)


def short_reads(n, seed=SEED_S, n_targets=8, target_len=125_000_000, n_sites=200_000,
                genome=False, read_len=100):
    """Config S (genome=False) or G (genome=True).  Returns (batch, target names,
    Genome or None)."""
    rng = np.random.default_rng(seed)
    names = ["chr%d" % (i + 1) for i in range(n_targets)]
    # junction sites; 2-intron reads continue into a fixed second intron per site
    site_tid = rng.integers(0, n_targets, n_sites)
    margin = 1_300_000
    assert target_len > 2 * margin
    site_start = rng.integers(1000, target_len - margin, n_sites).astype(np.int64)
    site_len = _intron_lengths(rng, n_sites)
    site_exon = rng.integers(1, 60, n_sites).astype(np.int64)
    site_len2 = _intron_lengths(rng, n_sites)

    gen = None
    site_strand = np.zeros(n_sites, np.uint8)
    if genome:
        seqs = [rng.integers(0, 4, target_len, dtype=np.uint8) for _ in range(n_targets)]
        lut = np.frombuffer(b"ACGT", np.uint8)
        seqs = [lut[s] for s in seqs]
        u = rng.random(n_sites)
        kind = np.select([u < 0.45, u < 0.90, u < 0.95, u < 0.97, u < 0.99], [0, 1, 2, 3, 4], 5)
        motifs = {0: (b"GT", b"AG", ord("+")), 1: (b"CT", b"AC", ord("-")), 2: (b"GC", b"AG", ord("+")),
                  3: (b"AT", b"AC", ord("+"))}
        e1 = site_start + site_len
        s2 = e1 + site_exon
        e2 = s2 + site_len2
        for k in range(n_sites):
            t = site_tid[k]
            kd = int(kind[k])
            if kd in motifs:
                d, a, st = motifs[kd]
                site_strand[k] = st
                for (s, e) in ((site_start[k], e1[k]), (s2[k], e2[k])):
                    seqs[t][s:s + 2] = np.frombuffer(d, np.uint8)
                    seqs[t][e - 2:e] = np.frombuffer(a, np.uint8)
            elif kd == 5:     # N or lower case at the donor
                site_strand[k] = ord("+")
                seqs[t][site_start[k]:site_start[k] + 2] = np.frombuffer(b"Nt" if k & 1 else b"gt", np.uint8)
                seqs[t][e1[k] - 2:e1[k]] = np.frombuffer(b"ag" if k & 2 else b"AG", np.uint8)
            else:
                site_strand[k] = ord("+") if k & 1 else ord("-")
        gen = Genome(names, seqs)

    site = _zipf_choice(rng, n_sites, n)
    two = rng.random(n) < 0.3
    b_mid = site_exon[site]
    a = np.where(two, rng.integers(1, 10 ** 9, n) % np.maximum(read_len - 1 - b_mid, 1) + 1,
                 rng.integers(1, read_len, n)).astype(np.int64)
    a = np.minimum(a, read_len - 1 - np.where(two, b_mid, 0))
    b = np.where(two, b_mid, read_len - a)
    c = np.where(two, read_len - a - b_mid, 0)
    two &= c >= 1
    b = np.where(two, b, read_len - a)

    # optional edits
    u = rng.random(n)
    clip_lead = u < 0.025
    clip_trail = (u >= 0.025) & (u < 0.05)
    clip_len = rng.integers(1, 11, n)
    u = rng.random(n)
    indel = (u < 0.02) & (a >= 4)
    indel_is_del = rng.random(n) < 0.5
    indel_len = rng.integers(1, 4, n)
    a1 = np.where(indel, rng.integers(1, 10 ** 9, n) % np.maximum(a - 1, 1) + 1, a)
    a1 = np.where(indel, np.minimum(a1, np.maximum(a - 1, 1)), a)      # without an indel the anchor is one M op
    a2 = a - a1
    u = rng.random(n)
    eqx = u < 0.01
    first_op = np.where(eqx, np.where(rng.random(n) < 0.5, OP_EQ, OP_X), OP_M).astype(np.uint32)

    tid = site_tid[site]
    start = site_start[site]
    pos = start - a - np.where(indel & indel_is_del, indel_len, 0)

    ops = np.zeros((n, 9), np.uint32)
    valid = np.zeros((n, 9), bool)
    ops[:, 0] = _pack(clip_len, OP_S); valid[:, 0] = clip_lead
    ops[:, 1] = (a1.astype(np.uint32) << 4) | first_op; valid[:, 1] = True
    ops[:, 2] = _pack(indel_len, np.where(indel_is_del, OP_D, OP_I)); valid[:, 2] = indel
    ops[:, 3] = _pack(a2, OP_M); valid[:, 3] = indel
    ops[:, 4] = _pack(site_len[site], OP_N); valid[:, 4] = True
    ops[:, 5] = _pack(b, OP_M); valid[:, 5] = True
    ops[:, 6] = _pack(site_len2[site], OP_N); valid[:, 6] = two
    ops[:, 7] = _pack(np.maximum(c, 1), OP_M); valid[:, 7] = two
    ops[:, 8] = _pack(clip_len, OP_S); valid[:, 8] = clip_trail

    flag = rng.choice(np.array([0, 16, 99, 147, 83, 163], np.uint16), n)
    u = rng.random(n)
    flag = np.where(u < 0.03, flag | 256, flag)
    flag = np.where((u >= 0.03) & (u < 0.0325), flag | 1024, flag)
    flag = np.where((u >= 0.0325) & (u < 0.035), flag | 512, flag)
    u = rng.random(n)
    nh = np.select([u < 0.90, u < 0.98], [1, 2], 0).astype(np.uint8)
    xs = np.zeros(n, np.uint8)
    if genome:
        u = rng.random(n)
        agree = site_strand[site]
        flip = np.where(agree == ord("+"), ord("-"), ord("+")).astype(np.uint8)
        xs = np.where(u < 0.5, agree, np.where(u < 0.51, flip, 0)).astype(np.uint8)
    return _flatten(ops, valid, tid, pos, flag, xs, nh), names, gen


def long_reads(n, seed=SEED_L, n_targets=8, target_len=125_000_000, n_transcripts=20_000,
               mean_len=10_000.0, sigma=0.4):
    """Config L: ONT-style reads over a fixed set of transcripts (so that reads
    share junctions): 20-30 introns, exons broken into M(5-40) pieces joined by
    1-3 bp I/D (2 %: 37-80 bp), leading/trailing S or H, flags
    {0,16,256,2048,2064}, no NH, no XS."""
    rng = np.random.default_rng(seed)
    names = ["chr%d" % (i + 1) for i in range(n_targets)]
    T = n_transcripts
    t_tid = rng.integers(0, n_targets, T)
    t_nint = rng.integers(20, 31, T)
    max_int = 30
    t_ilen = _log_uniform(rng, 70, 20000, T * max_int).reshape(T, max_int)
    t_start = rng.integers(1000, target_len - 2_000_000, T).astype(np.int64)

    tr = rng.integers(0, T, n)
    nint = t_nint[tr]
    nex = nint + 1
    rlen = np.exp(rng.normal(np.log(mean_len), sigma, n))
    exon_len = np.maximum((rlen / nex).astype(np.int64), 30)     # same for all exons of a read... but
    # junction coordinates must be shared: exon lengths come from the transcript, the read length
    # only trims the first and last exon
    t_elen = np.maximum((mean_len / (t_nint + 1)).astype(np.int64), 30)
    E = t_elen[tr]
    first_trim = rng.integers(0, np.maximum(E // 2, 1))
    last_trim = rng.integers(0, np.maximum(E // 2, 1))
    del exon_len

    # pieces per exon: about one M piece per 23 reference bases
    ex_read = np.repeat(np.arange(n), nex)                 # exon -> read
    ex_idx = np.arange(len(ex_read)) - np.repeat(np.cumsum(nex) - nex, nex)
    ex_ref = E[ex_read].copy()
    ex_ref[ex_idx == 0] -= first_trim[ex_read[ex_idx == 0]]
    is_last = ex_idx == (nex[ex_read] - 1)
    ex_ref[is_last] -= last_trim[ex_read[is_last]]
    k = np.maximum(ex_ref // 23, 1)                       # M pieces in the exon
    n_ex = len(k)
    # gaps between pieces: (k-1) per exon
    g_ex = np.repeat(np.arange(n_ex), k - 1)
    n_gap = len(g_ex)
    g_is_del = rng.random(n_gap) < 0.5
    g_len = rng.integers(1, 4, n_gap)
    big = rng.random(n_gap) < 0.02
    g_len[big] = rng.integers(37, 81, int(big.sum()))
    del_ref = np.bincount(g_ex, weights=np.where(g_is_del, g_len, 0), minlength=n_ex).astype(np.int64)
    # if deletions eat too much of the exon, turn that exon's deletions into insertions
    too_much = del_ref > ex_ref - k
    fix = too_much[g_ex]
    g_is_del &= ~fix
    del_ref[too_much] = 0
    m_total = ex_ref - del_ref                            # reference bases in M pieces, >= k
    # partition m_total into k positive parts: k-1 distinct cut points
    p_ex = np.repeat(np.arange(n_ex), k)                  # piece -> exon
    p_idx = np.arange(len(p_ex)) - np.repeat(np.cumsum(k) - k, k)
    base = m_total // k
    extra = m_total - base * k
    m_len = base[p_ex] + (p_idx < extra[p_ex])
    # jitter: move up to (len-5) bases between neighbours inside an exon, keeping parts >= 1
    jit = (rng.integers(0, 8, len(m_len)) * (m_len > 12)).astype(np.int64)
    same_next = np.zeros(len(m_len), bool)
    same_next[:-1] = p_ex[1:] == p_ex[:-1]
    give = np.where(same_next, jit, 0)
    m_len = m_len - give
    m_len[1:] += give[:-1]

    # op stream per read: [clip] { M (gap M)* } (N { M (gap M)* })* [clip]
    # build by assigning every op a sort key (read, exon, piece, kind)
    lead = rng.random(n) < 0.7
    trail = rng.random(n) < 0.7
    clip_op = np.where(rng.random(n) < 0.8, OP_S, OP_H)
    clip_len = rng.integers(5, 300, n)

    piece_read = ex_read[p_ex]
    gap_read = ex_read[g_ex]
    g_idx = np.arange(n_gap) - np.repeat(np.cumsum(k - 1) - (k - 1), k - 1)
    intr_read = np.repeat(np.arange(n), nint)
    intr_idx = np.arange(len(intr_read)) - np.repeat(np.cumsum(nint) - nint, nint)
    intr_len = t_ilen[tr[intr_read], intr_idx]

    def key(read, exon, piece, kind):
        return (read.astype(np.int64) << 34) | ((exon.astype(np.int64) + 1) << 20) | (piece.astype(np.int64) << 2) | kind

    keys = np.concatenate([
        key(np.flatnonzero(lead), np.full(int(lead.sum()), -1), np.zeros(int(lead.sum()), np.int64), 0),
        key(piece_read, ex_idx[p_ex], p_idx, 0),
        key(gap_read, ex_idx[g_ex], g_idx, 1),
        key(intr_read, intr_idx, np.full(len(intr_read), (1 << 18) - 1), 2),
        key(np.flatnonzero(trail), np.full(int(trail.sum()), 40), np.zeros(int(trail.sum()), np.int64), 0),
    ])
    vals = np.concatenate([
        _pack(clip_len[lead], clip_op[lead]),
        _pack(m_len, OP_M),
        _pack(g_len, np.where(g_is_del, OP_D, OP_I)),
        _pack(intr_len, OP_N),
        _pack(clip_len[trail], clip_op[trail]),
    ])
    order = np.argsort(keys, kind="stable")
    cigar_by_read = vals[order]
    n_cigar = (lead.astype(np.int64) + trail + np.bincount(ex_read, weights=2 * k - 1, minlength=n).astype(np.int64)
               + nint)

    pos = t_start[tr] + first_trim
    tid = t_tid[tr]
    flag = rng.choice(np.array([0, 16, 256, 2048, 2064], np.uint16), n, p=[0.42, 0.42, 0.06, 0.05, 0.05])
    # coordinate sort the reads, keeping each read's op run intact
    rorder = np.lexsort((pos, tid))
    off_old = np.zeros(n + 1, np.int64)
    np.cumsum(n_cigar, out=off_old[1:])
    nc_new = n_cigar[rorder]
    off_new = np.zeros(n + 1, np.int64)
    np.cumsum(nc_new, out=off_new[1:])
    src = np.repeat(off_old[rorder] - off_new[:-1], nc_new) + np.arange(off_new[-1])
    cigar = cigar_by_read[src]
    assert cigar.size < 2 ** 32
    batch = {
        "tid": tid[rorder].astype(np.int32), "pos": pos[rorder].astype(np.int32),
        "flag": flag[rorder].astype(np.uint16), "xs": np.zeros(n, np.uint8), "nh": np.zeros(n, np.uint8),
        "cigar_off": off_new[:-1].astype(np.uint32), "n_cigar": nc_new.astype(np.uint32),
        "cigar": np.ascontiguousarray(cigar, np.uint32),
    }
    return batch, names, None


def concat_batches(batches):
    """Concatenate SoA batches (offsets rebased)."""
    out = {k: np.concatenate([b[k] for b in batches]) for k in ("tid", "pos", "flag", "xs", "nh", "n_cigar", "cigar")}
    offs, base = [], 0
    for b in batches:
        offs.append(b["cigar_off"].astype(np.uint64) + base)
        base += len(b["cigar"])
    out["cigar_off"] = np.concatenate(offs).astype(np.uint32)
    return out


def permute_batch(batch, perm):
    """The alignments of a batch in the order perm (CIGAR words moved along, offsets rebuilt)."""
    perm = np.asarray(perm, np.int64)
    nc = batch["n_cigar"][perm].astype(np.int64)
    off_old = batch["cigar_off"][perm].astype(np.int64)
    off_new = np.zeros(len(perm), np.int64)
    np.cumsum(nc[:-1], out=off_new[1:])
    src = np.repeat(off_old - off_new, nc) + np.arange(int(nc.sum()), dtype=np.int64)
    out = {k: batch[k][perm].copy() for k in ("tid", "pos", "flag", "xs", "nh", "n_cigar")}
    out["cigar_off"] = off_new.astype(np.uint32)
    out["cigar"] = batch["cigar"][src].copy()
    return out


def slice_batch(batch, lo, hi):
    """Alignments [lo, hi) of a batch as a self-contained batch."""
    if hi <= lo:
        return {k: batch[k][:0].copy() for k in batch}
    c0 = int(batch["cigar_off"][lo])
    c1 = int(batch["cigar_off"][hi - 1]) + int(batch["n_cigar"][hi - 1])
    out = {k: batch[k][lo:hi].copy() for k in ("tid", "pos", "flag", "xs", "nh", "n_cigar")}
    out["cigar_off"] = (batch["cigar_off"][lo:hi].astype(np.int64) - c0).astype(np.uint32)
    out["cigar"] = batch["cigar"][c0:c1].copy()
    return out


def cigar_string(batch, i):
    ops = "MIDNSHP=XB"
    o, n = int(batch["cigar_off"][i]), int(batch["n_cigar"][i])
    return "".join("%d%s" % (w >> 4, ops[w & 15]) for w in batch["cigar"][o:o + n])
