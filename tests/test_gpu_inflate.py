"""GPU: BGZF on the device (csrc/ipx_inflate.hip): ipx_bgzf_inflate against zlib, and ipx_decode_bgzf -- a whole BAM file
inflated, CRC-checked, cut into records and decoded into the SoA batch -- against the batch the file was written from."""
import ctypes as C
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _inflate_on_device(comp, in_off, isize):
    from intronprospector_amd import _lib as L
    lib = L.load()
    comp = np.frombuffer(comp, np.uint8)
    in_off = np.asarray(in_off, np.uint64)
    isize = np.asarray(isize, np.uint32)
    out_off = np.zeros(len(isize), np.uint64)
    out_off[1:] = np.cumsum(isize[:-1].astype(np.uint64))
    total = int(isize.astype(np.uint64).sum())
    out = np.zeros(max(total, 1), np.uint8)
    status = np.full(len(isize), 99, np.uint32)
    ms = C.c_float(0)
    rc = lib.ipx_bgzf_inflate(0, comp.ctypes.data, len(comp), in_off.ctypes.data, isize.ctypes.data, out_off.ctypes.data,
                              len(isize), out.ctypes.data, len(out), status.ctypes.data, C.byref(ms))
    assert rc == 0
    return out[:total], out_off, status, ms.value


def test_bgzf_blocks_of_a_bam_file(tmp_path):
    bam = str(tmp_path / "b.bam")
    subprocess.run([os.path.join(ROOT, "intronprospector_amd", "bin", "ipx_host_check"), "make-bam", "200000", bam], check=True)
    data = open(bam, "rb").read()
    in_off, isize, want = [], [], []
    p = 0
    while p < len(data):
        xlen = struct.unpack("<H", data[p + 10:p + 12])[0]
        bsize = struct.unpack("<H", data[p + 16:p + 18])[0] + 1
        in_off.append(p + 12 + xlen)
        isize.append(struct.unpack("<I", data[p + bsize - 4:p + bsize])[0])
        want.append(zlib.decompress(data[p + 12 + xlen:p + bsize - 8], -15))
        p += bsize
    # the stream of block b ends where its trailer begins, not where block b + 1 starts
    ends = []
    p = 0
    while p < len(data):
        bsize = struct.unpack("<H", data[p + 16:p + 18])[0] + 1
        ends.append(p + bsize - 8)
        p += bsize
    # ipx_bgzf_inflate takes one offsets array (stream b = [in_off[b], in_off[b + 1])): hand it the streams repacked
    packed = b"".join(data[a:e] for a, e in zip(in_off, ends))
    off = np.zeros(len(in_off) + 1, np.uint64)
    off[1:] = np.cumsum([e - a for a, e in zip(in_off, ends)])
    out, out_off, status, ms = _inflate_on_device(packed, off, isize)
    assert (status == 0).all(), status[status != 0][:10]
    assert out.tobytes() == b"".join(want)


def test_every_block_type_and_damage():
    rng = np.random.default_rng(5)
    datas = [b"", b"a", b"ab" * 3, b"\0" * 65536, bytes(rng.integers(0, 256, 65280, dtype=np.uint8)),
             bytes(rng.integers(0, 4, 65280, dtype=np.uint8)), (b"ACGT" * 20000)[:65280],
             b"".join(bytes([int(x)]) * int(k) for x, k in zip(rng.integers(0, 256, 2000), rng.integers(1, 300, 2000)))[:65280],
             open(__file__, "rb").read()]
    streams, sizes = [], []
    for d in datas:
        for lvl, st in ((0, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY),
                        (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE)):
            c = zlib.compressobj(lvl, zlib.DEFLATED, -15, 8, st)
            streams.append(c.compress(d) + c.flush())
            sizes.append(len(d))
    for _ in range(200):
        k = int(rng.integers(0, 65536))
        base = bytes(rng.integers(0, 256, int(rng.integers(1, 400)), dtype=np.uint8))
        d = (base * (k // len(base) + 1))[:k] if rng.random() < 0.5 else bytes(rng.integers(0, int(rng.integers(2, 256)), k, dtype=np.uint8))
        c = zlib.compressobj(int(rng.integers(0, 10)), zlib.DEFLATED, -15, int(rng.integers(1, 10)),
                             [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_RLE, zlib.Z_HUFFMAN_ONLY][int(rng.integers(0, 4))])
        streams.append(c.compress(d) + c.flush())
        sizes.append(len(d))
    off = np.zeros(len(streams) + 1, np.uint64)
    off[1:] = np.cumsum([len(s) for s in streams])
    packed = b"".join(streams)
    out, out_off, status, _ = _inflate_on_device(packed, off, sizes)
    assert (status == 0).all(), np.nonzero(status)[0][:10]
    want = b"".join(zlib.decompress(s, -15) for s in streams)
    assert out.tobytes() == want
    # damage: every stream with a few bits flipped; whatever comes out, the call returns and good neighbours stay good
    bad = bytearray(packed)
    hit = set()
    for b in range(0, len(streams), 2):
        lo, hi = int(off[b]), int(off[b + 1])
        if hi > lo:
            for _ in range(3):
                bad[int(rng.integers(lo, hi))] ^= 1 << int(rng.integers(0, 8))
            hit.add(b)
    out2, _, status2, _ = _inflate_on_device(bytes(bad), off, sizes)
    for b in range(len(streams)):
        if b not in hit:
            assert status2[b] == 0
            lo = int(out_off[b])
            assert out2[lo:lo + sizes[b]].tobytes() == want[lo:lo + sizes[b]]
        elif status2[b] == 0:
            # accepted although damaged: then zlib accepts the same bytes to the same output
            lo, hi = int(off[b]), int(off[b + 1])
            try:
                z = zlib.decompressobj(-15).decompress(bytes(bad[lo:hi]))
            except zlib.error:
                z = None
            o = int(out_off[b])
            assert z is not None and z[:sizes[b]] == out2[o:o + sizes[b]].tobytes()


def _bgzf_file(header, record_bytes, rec_begin, aligned=True, level=6):
    """BAM header + records as BGZF blocks; aligned: no record straddles a block end unless it is larger than a block
    (what htslib writes); else blocks are cut every 0xff00 bytes wherever that falls"""
    from tests import bamio
    blocks = [header]
    if aligned:
        cur = bytearray()
        for i in range(len(rec_begin) - 1):
            r = record_bytes[int(rec_begin[i]):int(rec_begin[i + 1])]
            if len(cur) + len(r) > 0xff00 and len(cur):
                blocks.append(bytes(cur))
                cur = bytearray()
            cur += r
            while len(cur) > 0xff00:             # a record larger than a block: it has to straddle
                blocks.append(bytes(cur[:0xff00]))
                cur = cur[0xff00:]
        if cur:
            blocks.append(bytes(cur))
    else:
        body = bytes(record_bytes)
        blocks += [body[i:i + 0xff00] for i in range(0, len(body), 0xff00)]
    return b"".join(bamio._bgzf_block(b) for b in blocks) + bamio._bgzf_block(b"")


def _block_table(data):
    pb, pe, isz, crc = [], [], [], []
    p = 0
    while p < len(data):
        xlen = struct.unpack("<H", data[p + 10:p + 12])[0]
        bsize = struct.unpack("<H", data[p + 16:p + 18])[0] + 1
        pb.append(p + 12 + xlen)
        pe.append(p + bsize - 8)
        c, s = struct.unpack("<II", data[p + bsize - 8:p + bsize])
        crc.append(c)
        isz.append(s)
        p += bsize
    return np.array(pb, np.uint64), np.array(pe, np.uint64), np.array(isz, np.uint32), np.array(crc, np.uint32)


def _decode_on_device(ex, data, records_begin, tid_map=None):
    import torch
    from intronprospector_amd import _lib as L
    pb, pe, isz, crc = _block_table(data)
    raw = np.frombuffer(data, np.uint8)
    tm = np.asarray(tid_map, np.int32) if tid_map is not None else None
    f = L.BgzfFile(raw.ctypes.data, len(raw), len(pb), pb.ctypes.data, pe.ctypes.data, isz.ctypes.data, crc.ctypes.data,
                   records_begin, tm.ctypes.data if tm is not None else None, len(tm) if tm is not None else 0)
    b = L.Batch()
    rc = ex.lib.ipx_decode_bgzf(ex.h, C.byref(f), C.byref(b))
    if rc:
        return rc, ex.last_error().detail.decode(), None
    n, nw = int(b.n_alignments), int(b.n_cigar_words)
    out = {}
    for k, dt in (("tid", np.int32), ("pos", np.int32), ("flag", np.uint16), ("xs", np.uint8), ("nh", np.uint8),
                  ("cigar_off", np.uint32), ("n_cigar", np.uint32)):
        a = np.zeros(n, dt)
        if n:
            ex._check(ex.lib.ipx_copy(ex.h, a.ctypes.data, getattr(b, k), a.nbytes))
        out[k] = a
    cg = np.zeros(nw, np.uint32)
    if nw:
        ex._check(ex.lib.ipx_copy(ex.h, cg.ctypes.data, b.cigar, cg.nbytes))
    out["cigar"] = cg
    return 0, "", (out, b)


def test_whole_bam_file_decoded_on_the_device():
    """ipx_decode_bgzf: inflate, CRC, record chain per block, fields, tags, CIGAR -- every alignment of the file equals
    the alignment it was written from (tags of every type, CG-tag CIGARs, a record larger than a block), the decoded
    batch pushed slice by slice gives the oracle's table, and damaged or unaligned files are refused."""
    import torch    # noqa: F401
    import intronprospector_amd as ipa
    from intronprospector_amd import synth
    from tests import bam_encode, helpers
    from tests.oracle_binding import Oracle
    short, names, _ = synth.short_reads(60_000, seed=21, n_targets=3, target_len=4_000_000, n_sites=1500)
    long_, _, _ = synth.long_reads(40, seed=22, n_targets=3, target_len=4_000_000, n_transcripts=5)     # ~1000-op CIGARs
    batch = synth.concat_batches([short, long_])
    rec, sb, sf, nw = bam_encode.records(batch, seed=23)
    n = len(batch["tid"])
    # record starts from the block_size chain
    rb = np.zeros(n + 1, np.uint64)
    off = 0
    for i in range(n):
        rb[i] = off
        off += 4 + struct.unpack("<I", rec[off:off + 4].tobytes())[0]
    rb[n] = off
    text = b"@HD\tVN:1.6\n" + b"".join(b"@SQ\tSN:%s\tLN:4000000\n" % nm.encode() for nm in names)
    header = b"BAM\1" + struct.pack("<I", len(text)) + text + struct.pack("<I", len(names)) + \
        b"".join(struct.pack("<I", len(nm) + 1) + nm.encode() + b"\0" + struct.pack("<I", 4000000) for nm in names)
    data = _bgzf_file(header, rec.tobytes(), rb)
    ex = ipa.JunctionsExtractor(targets=names)
    rc, msg, got = _decode_on_device(ex, data, len(header))
    assert rc == 0, msg
    dev, b = got
    for k in ("tid", "pos", "flag", "xs", "nh", "n_cigar"):
        assert np.array_equal(dev[k].astype(np.int64), batch[k].astype(np.int64)), k
    for i in (0, 1, n // 2, n - 41, n - 1):
        o, c = int(dev["cigar_off"][i]), int(dev["n_cigar"][i])
        o2 = int(batch["cigar_off"][i])
        assert np.array_equal(dev["cigar"][o:o + c], batch["cigar"][o2:o2 + c])
    assert len(dev["cigar"]) == len(batch["cigar"])
    # slices of the device batch through the ordinary push (cigar_off stays absolute)
    from intronprospector_amd import _lib as L
    cut = n // 3
    for lo, hi in ((0, cut), (cut, n)):
        s = L.Batch(hi - lo, b.n_cigar_words, lo, b.tid + 4 * lo, b.pos + 4 * lo, b.flag + 2 * lo, b.xs + lo, b.nh + lo,
                    b.cigar_off + 4 * lo, b.n_cigar + 4 * lo, b.cigar, L.IPX_MEM_DEVICE)
        ex._check(ex.lib.ipx_push_batch(ex.h, C.byref(s)))
    tab = ex.get_junctions()
    ora = Oracle(targets=names)
    ora.process(batch)
    helpers.compare_tables(tab, ora.table())
    ora.close()
    # a tid map is applied (refID 0 <-> 2)
    rc, msg, got = _decode_on_device(ex, data, len(header), tid_map=[2, 1, 0])
    assert rc == 0 and np.array_equal(got[0]["tid"], np.where(batch["tid"] >= 0, 2 - batch["tid"], batch["tid"]))
    # damage: a flipped payload byte (inflate error or CRC mismatch), a wrong CRC, records that straddle block ends
    bad = bytearray(data)
    bad[len(data) // 2] ^= 0x40
    rc, msg, _ = _decode_on_device(ex, bytes(bad), len(header))
    assert rc != 0 and ("ISIZE" in msg or "CRC" in msg), msg
    pb, pe, isz, crc = _block_table(data)
    bad = bytearray(data)
    bad[int(pe[3])] ^= 1                                           # the CRC field of block 3
    rc, msg, _ = _decode_on_device(ex, bytes(bad), len(header))
    assert rc != 0 and "CRC-32 mismatch in block 3" in msg, msg
    rc, msg, _ = _decode_on_device(ex, _bgzf_file(header, rec.tobytes(), rb, aligned=False), len(header))
    assert rc != 0 and "straddle" in msg, msg
    ex.close()
