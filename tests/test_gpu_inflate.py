"""GPU: ipx_bgzf_inflate (csrc/ipx_inflate.hip, an experiment: BGZF blocks inflated one lane per block) against zlib."""
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
