"""GPU: ipx_push_bam -- batches handed over as BAM alignment records and decoded on the device (ipx_bam.hip) --
against the same batches handed over as the SoA, and against the oracle."""
import numpy as np
import pytest

from tests import bam_encode
from tests import helpers
from tests.oracle_binding import Oracle

pytestmark = pytest.mark.gpu


def _tables(batch, names, params, seed, pieces=1, tid_map=None):
    import intronprospector_amd as ipa
    from intronprospector_amd import synth
    ex = ipa.JunctionsExtractor(targets=names, **params)
    n = len(batch["tid"])
    cuts = [n * k // pieces for k in range(pieces + 1)]
    for k in range(pieces):
        part = synth.slice_batch(batch, cuts[k], cuts[k + 1])
        if tid_map is not None:                       # the records carry refIDs the map turns back into the tids
            part = dict(part)
            inv = {int(t): r for r, t in enumerate(tid_map)}
            part["tid"] = np.array([inv.get(int(t), int(t)) if t >= 0 else int(t) for t in part["tid"]], np.int32)
        data, sb, sf, nw = bam_encode.records(part, seed + k)
        ex.push_bam(data, sb, sf, nw, tid_map=tid_map)
    dev = ex.get_junctions()
    ex.close()
    ora = Oracle(targets=names, **params)
    ora.process(batch)
    want = ora.table()
    ora.close()
    return dev, want


def test_short_reads_as_bam_records_match_oracle(oracle_lib):
    from intronprospector_amd import synth
    batch, names, _ = synth.short_reads(30_000, seed=301, n_targets=3, target_len=3_000_000, n_sites=800)
    dev, want = _tables(batch, names, {"min_anchor_length": 3}, seed=302, pieces=3)
    assert "error" not in dev
    helpers.compare_tables(dev, want)


def test_long_reads_as_bam_records_match_oracle(oracle_lib):
    from intronprospector_amd import synth
    batch, names, _ = synth.long_reads(700, seed=303, n_targets=2, target_len=9_000_000, n_transcripts=40)
    dev, want = _tables(batch, names, {"max_anchor_indel_size": 36}, seed=304, pieces=2)
    assert "error" not in dev
    helpers.compare_tables(dev, want)


def test_tid_map_is_applied(oracle_lib):
    from intronprospector_amd import synth
    batch, names, _ = synth.short_reads(5_000, seed=305, n_targets=3, target_len=3_000_000, n_sites=200)
    # refIDs 0..5 in the file stand for tids 0,1,2,0,1,2 (the same chromosome named twice in the header)
    tid_map = np.array([0, 1, 2, 0, 1, 2], np.int32)
    batch2 = dict(batch)
    dev, want = _tables(batch2, names, {}, seed=306, tid_map=tid_map)
    helpers.compare_tables(dev, want)


def _one(words, **kw):
    b = {"tid": np.array([0], np.int32), "pos": np.array([1000], np.int32), "flag": np.array([0], np.uint16),
         "xs": np.array([0], np.uint8), "nh": np.array([1], np.uint8), "cigar_off": np.array([0], np.uint32),
         "n_cigar": np.array([len(words)], np.uint32), "cigar": np.array(words, np.uint32)}
    b.update(kw)
    return b


@pytest.mark.parametrize("damage", ["block_size_small", "block_size_past_span", "bad_tag_type", "tag_past_record",
                                    "name_past_record", "word_count"])
def test_malformed_records_are_refused(damage):
    """what host/aln_reader.cc: parse_bam_record throws "Error reader BAM record" for"""
    import struct
    import intronprospector_amd as ipa
    from intronprospector_amd import _lib as L
    M, N = 0, 3
    words = [(20 << 4) | M, (100 << 4) | N, (30 << 4) | M]
    good = {"tid": np.zeros(3, np.int32), "pos": np.array([100, 200, 300], np.int32), "flag": np.zeros(3, np.uint16),
            "xs": np.zeros(3, np.uint8), "nh": np.ones(3, np.uint8), "cigar_off": np.array([0, 3, 6], np.uint32),
            "n_cigar": np.full(3, 3, np.uint32), "cigar": np.array(words * 3, np.uint32)}
    data, sb, sf, nw = bam_encode.records(good, 310, cg_fraction=0.0)
    data = data.copy()
    second = int(4 + struct.unpack("<I", data[:4].tobytes())[0])          # offset of record 1
    bs1 = struct.unpack("<I", data[second:second + 4].tobytes())[0]
    sb = np.array([0, len(data)], np.uint64)
    sf = np.array([0, 3], np.uint64)
    if damage == "block_size_small":
        data[second:second + 4] = np.frombuffer(struct.pack("<I", 20), np.uint8)
    elif damage == "block_size_past_span":
        data[second:second + 4] = np.frombuffer(struct.pack("<I", bs1 + 100_000), np.uint8)
    elif damage == "bad_tag_type":
        data = np.concatenate([data[:second + 4 + bs1], np.frombuffer(b"QQ?x", np.uint8), data[second + 4 + bs1:]])
        data[second:second + 4] = np.frombuffer(struct.pack("<I", bs1 + 4), np.uint8)
        sb = np.array([0, len(data)], np.uint64)
    elif damage == "tag_past_record":
        data = np.concatenate([data[:second + 4 + bs1], np.frombuffer(b"QQi\1", np.uint8), data[second + 4 + bs1:]])
        data[second:second + 4] = np.frombuffer(struct.pack("<I", bs1 + 4), np.uint8)
        sb = np.array([0, len(data)], np.uint64)
    elif damage == "name_past_record":
        data[second + 4 + 8] = 255
        data[second + 4 + 12:second + 4 + 14] = np.frombuffer(struct.pack("<H", 60000), np.uint8)
    elif damage == "word_count":
        nw += 1
    ex = ipa.JunctionsExtractor(targets=["chr1"])
    with pytest.raises(ipa.IpxError) as e:
        ex.push_bam(data, sb, sf, nw)
    assert e.value.status == L.IPX_EINVAL
    # the handle is as it was: a good batch still goes through, and only its alignments count
    data, sb, sf, nw = bam_encode.records(good, 311)
    ex.push_bam(data, sb, sf, nw)
    t = ex.get_junctions()
    assert "error" not in t and len(t["intron_start"]) == 3 and int(t["counts"].sum()) == 3
    ex.close()


def test_span_tables_are_checked():
    import intronprospector_amd as ipa
    good = _one([(20 << 4), (100 << 4) | 3, (30 << 4)])
    data, sb, sf, nw = bam_encode.records(good, 312)
    ex = ipa.JunctionsExtractor(targets=["chr1"])
    for bad_sb, bad_sf in ((np.array([4, len(data)], np.uint64), sf), (np.array([0, len(data) - 1], np.uint64), sf),
                           (sb, np.array([0, 2], np.uint64)), (np.array([0, len(data) + 4, len(data)], np.uint64), np.array([0, 1, 1], np.uint64))):
        with pytest.raises(ipa.IpxError):
            ex.push_bam(data, bad_sb, bad_sf, nw)
    ex.close()
