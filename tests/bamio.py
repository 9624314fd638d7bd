"""Test helper: SAM text -> BAM (BGZF) written from the SAM/BAM specification, so that
the BAM-input rules of the reference's test-suite (testBasic3/4/5) can run without
samtools."""
import struct
import zlib

OPS = "MIDNSHP=XB"
SEQ = "=ACMGRSVTWYHKDBN"


def _bgzf_block(data):
    comp = zlib.compressobj(6, zlib.DEFLATED, -15)
    cdata = comp.compress(data) + comp.flush()
    bsize = len(cdata) + 25
    hdr = struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, bsize)
    return hdr + cdata + struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data))


def _reg2bin(beg, end):
    end -= 1
    for shift, off in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> shift == end >> shift:
            return off + (beg >> shift)
    return 0


def _aux(tag):
    name, typ, val = tag.split(":", 2)
    t = name.encode()
    if typ == "A":
        return t + b"A" + val.encode()[:1]
    if typ == "i":
        v = int(val)
        for code, fmt, lo, hi in (("c", "<b", -128, 127), ("C", "<B", 0, 255), ("s", "<h", -32768, 32767),
                                  ("S", "<H", 0, 65535), ("i", "<i", -2 ** 31, 2 ** 31 - 1), ("I", "<I", 0, 2 ** 32 - 1)):
            if lo <= v <= hi:
                return t + code.encode() + struct.pack(fmt, v)
        raise ValueError(tag)
    if typ == "f":
        return t + b"f" + struct.pack("<f", float(val))
    if typ in "ZH":
        return t + typ.encode() + val.encode() + b"\0"
    if typ == "B":
        sub, *items = val.split(",")
        fmt = {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I", "f": "f"}[sub]
        conv = float if sub == "f" else int
        return t + b"B" + sub.encode() + struct.pack("<I", len(items)) + struct.pack("<%d%s" % (len(items), fmt), *map(conv, items))
    raise ValueError(tag)


def sam_to_bam(sam_path, bam_path, long_cigar_threshold=65535, aligned=False):
    """aligned: blocks end at record boundaries (what htslib writes: bgzf_flush_try before a record that does not fit);
    else the stream is cut every 60000 bytes wherever that falls"""
    header_lines, names, lens, records = [], [], [], []
    with open(sam_path) as f:
        for line in f:
            line = line.rstrip("\n")
            if line.startswith("@"):
                header_lines.append(line)
                if line.startswith("@SQ\t"):
                    d = dict(x.split(":", 1) for x in line.split("\t")[1:] if ":" in x)
                    names.append(d["SN"])
                    lens.append(int(d.get("LN", 0)))
            elif line:
                records.append(line.split("\t"))
    text = ("\n".join(header_lines) + "\n").encode() if header_lines else b""
    out = bytearray(b"BAM\1" + struct.pack("<I", len(text)) + text + struct.pack("<I", len(names)))
    for n, l in zip(names, lens):
        out += struct.pack("<I", len(n) + 1) + n.encode() + b"\0" + struct.pack("<I", l)
    ends = [len(out)]                  # where the header and every record end
    for f in records:
        qname, flag, rname, pos, mapq, cigar, rnext, pnext, tlen, seq, qual = f[:11]
        tid = names.index(rname) if rname != "*" else -1
        ntid = tid if rnext == "=" else (names.index(rnext) if rnext != "*" else -1)
        cig, num = [], ""
        if cigar != "*":
            for ch in cigar:
                if ch.isdigit():
                    num += ch
                else:
                    cig.append((int(num) << 4) | OPS.index(ch))
                    num = ""
        ref_len = sum(w >> 4 for w in cig if OPS[w & 15] in "MDN=X")
        p0 = int(pos) - 1
        l_seq = 0 if seq == "*" else len(seq)
        aux = b"".join(_aux(t) for t in f[11:])
        cig_words = cig
        if len(cig) > long_cigar_threshold:           # SAM spec 4.2.2: real CIGAR in CG:B:I
            aux += b"CGBI" + struct.pack("<I", len(cig)) + struct.pack("<%dI" % len(cig), *cig)
            cig_words = [(l_seq << 4) | 4, (ref_len << 4) | 3]
        sq = bytearray((l_seq + 1) // 2)
        for i, ch in enumerate(seq if seq != "*" else ""):
            v = SEQ.index(ch.upper()) if ch.upper() in SEQ else 15
            sq[i // 2] |= v << (4 if i % 2 == 0 else 0)
        ql = bytes([0xff] * l_seq) if qual == "*" else bytes(ord(c) - 33 for c in qual)
        body = struct.pack("<iiBBHHHIiii", tid, p0, len(qname) + 1, int(mapq), _reg2bin(p0, p0 + max(ref_len, 1)),
                           len(cig_words), int(flag), l_seq, ntid, int(pnext) - 1, int(tlen))
        body += qname.encode() + b"\0" + struct.pack("<%dI" % len(cig_words), *cig_words) + bytes(sq) + ql + aux
        out += struct.pack("<I", len(body)) + body
        ends.append(len(out))
    with open(bam_path, "wb") as fo:
        if aligned:
            start = 0
            for a, b in zip(ends[:-1], ends[1:]):
                if b - start > 0xff00 and a > start:
                    fo.write(_bgzf_block(bytes(out[start:a])))
                    start = a
                while b - start > 0xff00:             # a record larger than a block
                    fo.write(_bgzf_block(bytes(out[start:start + 0xff00])))
                    start += 0xff00
            if len(out) > start:
                fo.write(_bgzf_block(bytes(out[start:])))
        else:
            for i in range(0, len(out), 60000):
                fo.write(_bgzf_block(bytes(out[i:i + 60000])))
        fo.write(_bgzf_block(b""))
