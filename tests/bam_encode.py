"""Test helper: the alignments of a synthetic batch (the SoA of include/ipx.h) written as BAM alignment records
(SAM spec 4.2) with everything the device decoder has to step over -- read names, sequences and qualities of
random length, tags of every type in random order, duplicate XS / NH tags, XS / NH of the wrong type, long
CIGARs moved to a CG tag -- and cut into spans of whole records."""
import struct

import numpy as np


def _tag(name, typ, payload):
    return name.encode() + typ.encode() + payload


def _filler(rng):
    """a tag the hot path does not read"""
    k = int(rng.integers(0, 10))
    if k == 0:
        return _tag("MD", "Z", bytes(rng.integers(48, 90, int(rng.integers(0, 40)), dtype=np.uint8)) + b"\0")
    if k == 1:
        return _tag("AS", "C", struct.pack("<B", int(rng.integers(0, 256))))
    if k == 2:
        return _tag("NM", "s", struct.pack("<h", int(rng.integers(-300, 300))))
    if k == 3:
        return _tag("XX", "H", b"0A1B2C"[: 2 * int(rng.integers(0, 4))] + b"\0")
    if k == 4:
        sub = "cCsSiIf"[int(rng.integers(0, 7))]
        cnt = int(rng.integers(0, 9))
        es = {"c": 1, "C": 1, "s": 2, "S": 2, "i": 4, "I": 4, "f": 4}[sub]
        return _tag("ZB", "B", sub.encode() + struct.pack("<I", cnt) + bytes(rng.integers(0, 256, cnt * es, dtype=np.uint8)))
    if k == 5:
        return _tag("ZF", "f", struct.pack("<f", 1.5))
    if k == 6:
        return _tag("ZD", "d", struct.pack("<d", 2.5))
    if k == 7:
        return _tag("RG", "Z", b"grp1\0")
    if k == 8:
        return _tag("XA", "A", b"q")
    return _tag("ZI", "I", struct.pack("<I", int(rng.integers(0, 2**32))))


def _nh_tag(rng, cls):
    """an NH tag whose bam_aux2i value falls in class cls (0: <= 0 or not an integer, 1: == 1, 2: > 1)"""
    if cls == 0:
        k = int(rng.integers(0, 3))
        if k == 0:
            return _tag("NH", "c", struct.pack("<b", int(rng.integers(-100, 1))))
        if k == 1:
            return _tag("NH", "f", struct.pack("<f", 3.0))            # bam_aux2i of a float tag: 0
        return _tag("NH", "i", struct.pack("<i", -5))
    v = 1 if cls == 1 else int(rng.integers(2, 200))
    typ = "cCsSiI"[int(rng.integers(0, 6))]
    fmt = {"c": "<b", "C": "<B", "s": "<h", "S": "<H", "i": "<i", "I": "<I"}[typ]
    if typ == "c" and v > 127:
        v = 127
    return _tag("NH", typ, struct.pack(fmt, v))


def records(batch, seed, cg_fraction=0.02, max_span_records=300):
    """-> (uint8 array of the records, span_begin, span_first, n_cigar_words)"""
    rng = np.random.default_rng(seed)
    n = len(batch["tid"])
    out = bytearray()
    rec_begin = np.zeros(n + 1, np.uint64)
    for i in range(n):
        nc = int(batch["n_cigar"][i])
        off = int(batch["cigar_off"][i])
        words = np.asarray(batch["cigar"][off:off + nc], dtype="<u4")
        name = bytes(rng.integers(33, 127, int(rng.integers(1, 40)), dtype=np.uint8)) + b"\0"
        l_seq = int(rng.integers(0, 160))
        tags = [_filler(rng) for _ in range(int(rng.integers(0, 4)))]
        xs = int(batch["xs"][i])
        if xs:
            if xs == ord(".") and rng.random() < 0.5:
                tags.append(_tag("XS", "Z", b"+\0"))                   # bam_aux2A of a non-A tag: 0 -> '.'
            else:
                tags.append(_tag("XS", "A", bytes([xs])))
            if rng.random() < 0.2:
                tags.append(_tag("XS", "A", b"-" if xs != ord("-") else b"+"))     # only the first counts
        nh = int(batch["nh"][i])
        if nh or rng.random() < 0.5:
            tags.append(_nh_tag(rng, nh))
            if rng.random() < 0.2:
                tags.append(_nh_tag(rng, (nh + 1) % 3))
        # keep the first XS / NH first among their kind, shuffle the rest around them
        order = rng.permutation(len(tags))
        firsts = {}
        shuffled = [tags[j] for j in order]
        for want in (b"XS", b"NH"):
            idx = [k for k, t in enumerate(shuffled) if t[:2] == want]
            orig = [t for t in tags if t[:2] == want]
            for k, t in zip(idx, orig):
                shuffled[k] = t
        cigar_field = words.tobytes()
        n_cigar_field = nc
        if nc >= 1 and (nc > 65535 or rng.random() < cg_fraction):
            ref_len = int(rng.integers(1, 1 << 20))
            cigar_field = struct.pack("<II", (l_seq << 4) | 4, (ref_len << 4) | 3)
            n_cigar_field = 2
            shuffled.insert(int(rng.integers(0, len(shuffled) + 1)), _tag("CG", "B", b"I" + struct.pack("<I", nc) + words.tobytes()))
        body = struct.pack("<iiBBHHHIiii", int(batch["tid"][i]), int(batch["pos"][i]), len(name), int(rng.integers(0, 61)),
                           4680, n_cigar_field, int(batch["flag"][i]), l_seq, -1, -1, 0)
        body += name + cigar_field + bytes(rng.integers(0, 256, (l_seq + 1) // 2, dtype=np.uint8)) + \
            bytes(rng.integers(0, 42, l_seq, dtype=np.uint8)) + b"".join(shuffled)
        rec_begin[i] = len(out)
        out += struct.pack("<I", len(body)) + body
    rec_begin[n] = len(out)
    firsts = [0]
    while firsts[-1] < n:
        firsts.append(min(n, firsts[-1] + int(rng.integers(1, max_span_records + 1))))
    span_first = np.array(firsts, np.uint64)
    span_begin = rec_begin[span_first.astype(np.int64)]
    return np.frombuffer(bytes(out), np.uint8).copy(), span_begin, span_first, len(batch["cigar"])
