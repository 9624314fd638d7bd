"""Python host-side mirror of the reference's JunctionsExtractor for SoA batches.

This is harness plumbing over the C ABI (include/ipx.h) used by the tests and by
bench.py; the product host is the C++ code in intronprospector_amd/host.  The
method names follow src/junctions_extractor.hh:176-255 of the reference where
one exists (clear, get_junctions); errors raised by the input are re-raised with
the reference's message text.
"""
import ctypes as C

import numpy as np

from . import _lib as L


class IpxError(RuntimeError):
    def __init__(self, status, err):
        self.status = status
        self.kind = err.kind
        self.read_index = err.read_index
        self.target = err.target
        self.detail = err.detail.decode(errors="replace")
        self.order = int(getattr(err, "order", (1 << 64) - 1))
        self.err = err               # the ipx_error itself (a multi-GPU caller hands it to its peers)
        super().__init__(self.detail or ("ipx status %d" % status))


_NP = {
    "target": np.uint32, "intron_start": np.uint32, "intron_end": np.uint32, "anchor_start": np.uint32,
    "anchor_end": np.uint32, "strand": np.uint8, "motif": np.uint16, "creator_order": np.uint64,
    "ijunc": np.uint32,
}


def _view(ptr, n, dtype):
    if not n:
        return np.zeros(0, dtype)
    dt = np.dtype(dtype)
    buf = (C.c_char * (n * dt.itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dt, count=n).copy()


class JunctionsExtractor:
    def __init__(self, min_anchor_length=8, max_anchor_indel_size=0, min_intron_length=70,
                 max_intron_length=500000, junction_filter=L.IPX_SJ_NULL, strandness=L.IPX_UNSTRANDED,
                 excludes=0, skip_missing_targets=False, scope=L.IPX_SCOPE_TARGET, device=0,
                 targets=(), target_lengths=None, genome=None, combine=True):
        self.lib = L.load()
        p = L.Params(min_anchor_length, max_anchor_indel_size, min_intron_length, max_intron_length,
                     junction_filter, strandness, excludes, scope, int(bool(skip_missing_targets)), device)
        h = C.c_void_p()
        rc = self.lib.ipx_create(C.byref(p), C.byref(h))
        if rc == L.IPX_ENODEVICE:
            raise RuntimeError("intronprospector_amd: no gfx950 device available (ipx_create -> IPX_ENODEVICE); "
                               "there is no CPU fallback")
        if rc:
            raise RuntimeError("ipx_create failed with status %d" % rc)
        self.h = h
        if not combine:      # every junction record stays a record of its own (records_export)
            self._check(self.lib.ipx_set_option(self.h, L.IPX_OPT_COMBINE, 0))
        self.next_read_index = 0
        self._keep = []
        if len(targets):
            self.set_targets(targets, target_lengths)
        if genome is not None:
            self.set_genome(*genome)

    # -- configuration -----------------------------------------------------
    def last_error(self):
        e = L.Error()
        self.lib.ipx_last_error(self.h, C.byref(e))
        return e

    def _check(self, rc):
        if rc:
            raise IpxError(rc, self.last_error())

    def set_targets(self, names, lengths=None):
        arr = (C.c_char_p * len(names))(*[n.encode() if isinstance(n, str) else n for n in names])
        lens = None
        if lengths is not None:
            lens = (C.c_int64 * len(names))(*[int(x) for x in lengths])
        self._check(self.lib.ipx_set_targets(self.h, len(names), arr, lens))

    def set_genome(self, names, seqs_2bit):
        """names, [(twobit bytes, mask bytes, length), ...] as produced by synth.pack_2bit."""
        n = len(names)
        arr = (C.c_char_p * n)(*[s.encode() for s in names])
        lens = (C.c_int64 * n)(*[int(t[2]) for t in seqs_2bit])
        two = (C.c_void_p * n)(*[t[0].ctypes.data for t in seqs_2bit])
        msk = (C.c_void_p * n)(*[t[1].ctypes.data for t in seqs_2bit])
        self._check(self.lib.ipx_set_genome_2bit(self.h, n, arr, lens, two, msk))

    def reserve(self, n_alignments, n_cigar_words, n_records):
        self._check(self.lib.ipx_reserve(self.h, n_alignments, n_cigar_words, n_records))

    # -- data --------------------------------------------------------------
    def push(self, batch, base_read_index=None):
        """batch: dict of numpy arrays (host) with the keys of synth._flatten."""
        n = len(batch["tid"])
        if base_read_index is None:
            base_read_index = self.next_read_index
        arrs = {k: np.ascontiguousarray(batch[k], dt) for k, dt in (
            ("tid", np.int32), ("pos", np.int32), ("flag", np.uint16), ("xs", np.uint8), ("nh", np.uint8),
            ("cigar_off", np.uint32), ("n_cigar", np.uint32), ("cigar", np.uint32))}
        b = L.Batch(n, len(arrs["cigar"]), base_read_index,
                    *[arrs[k].ctypes.data for k in ("tid", "pos", "flag", "xs", "nh", "cigar_off", "n_cigar", "cigar")],
                    L.IPX_MEM_HOST)
        self._check(self.lib.ipx_push_batch(self.h, C.byref(b)))
        self.next_read_index = base_read_index + n

    def push_bam(self, data, span_begin, span_first, n_cigar_words, tid_map=None, base_read_index=None):
        """data: uint8 array of BAM alignment records (each with its block_size word) back to back; span k is
        data[span_begin[k]:span_begin[k+1]] and holds records span_first[k] .. span_first[k+1]-1 (ipx_push_bam)."""
        data = np.ascontiguousarray(data, np.uint8)
        sb = np.ascontiguousarray(span_begin, np.uint64)
        sf = np.ascontiguousarray(span_first, np.uint64)
        n = int(sf[-1])
        if base_read_index is None:
            base_read_index = self.next_read_index
        tm = None if tid_map is None else np.ascontiguousarray(tid_map, np.int32)
        b = L.BamBatch(n, int(n_cigar_words), base_read_index, data.ctypes.data, len(data), sb.ctypes.data, sf.ctypes.data,
                       len(sb) - 1, None if tm is None else tm.ctypes.data, 0 if tm is None else len(tm), L.IPX_MEM_HOST)
        self._check(self.lib.ipx_push_bam(self.h, C.byref(b)))
        self.next_read_index = base_read_index + n

    def push_device(self, n, n_words, ptrs, base_read_index=None):
        """ptrs: device addresses in the order tid,pos,flag,xs,nh,cigar_off,n_cigar,cigar."""
        if base_read_index is None:
            base_read_index = self.next_read_index
        b = L.Batch(n, n_words, base_read_index, *ptrs, L.IPX_MEM_DEVICE)
        self._check(self.lib.ipx_push_batch(self.h, C.byref(b)))
        self.next_read_index = base_read_index + n

    # -- multi-GPU record exchange ------------------------------------------
    def records_pending(self):
        n = C.c_uint64(0)
        self._check(self.lib.ipx_records_pending(self.h, C.byref(n)))
        return int(n.value)

    def records_sample(self, n_samples):
        out = np.zeros(n_samples, np.uint64)
        self._check(self.lib.ipx_records_sample(self.h, n_samples, out.ctypes.data_as(C.POINTER(C.c_uint64))))
        return out

    def records_partition(self, splitters, dst_ptr):
        """splitters: ascending uint64 keys (n_parts - 1 of them); dst_ptr: device buffer with room
        for 32 bytes per pending record.  -> counts per part."""
        spl = np.ascontiguousarray(splitters, np.uint64)
        n_parts = len(spl) + 1
        counts = np.zeros(n_parts, np.uint64)
        self._check(self.lib.ipx_records_partition(self.h, n_parts, spl.ctypes.data_as(C.POINTER(C.c_uint64)),
                                                   counts.ctypes.data_as(C.POINTER(C.c_uint64)), dst_ptr))
        return counts

    def records_adopt(self, n, src_ptr, n_alignments=0):
        self._check(self.lib.ipx_records_adopt(self.h, n, src_ptr, n_alignments))

    def records_partition_slots(self, splitters, slot_records, dst_ptr, error_order=(1 << 64) - 1, error_status=0,
                                spill_ptr=None, spill_records=0):
        """The pending records into len(splitters) + 1 slots of slot_records 32-byte records each (the first is the
        slot's header) at dst_ptr (device); queued on the handle's stream, nothing read back.  spill_ptr: device memory
        for spill_records records -- what the slots cannot hold goes there, part after part (ipx.h)."""
        spl = np.ascontiguousarray(splitters, np.uint64)
        self._check(self.lib.ipx_records_partition_slots_spill(self.h, len(spl) + 1, spl.ctypes.data_as(C.POINTER(C.c_uint64)),
                                                               int(slot_records), int(error_order), int(error_status), dst_ptr,
                                                               spill_ptr, int(spill_records) if spill_ptr else 0))

    def records_adopt_more(self, n, src_ptr):
        """n more received records behind the pending ones (the tails of a slotted exchange that overflowed)"""
        self._check(self.lib.ipx_records_adopt_more(self.h, int(n), src_ptr))

    def records_adopt_slots(self, n_slots, slot_records, src_ptr, n_alignments=0):
        """-> SlotsInfo: records adopted, the largest part any sender had, overflow, the earliest sender-side error"""
        info = L.SlotsInfo()
        self._check(self.lib.ipx_records_adopt_slots(self.h, int(n_slots), int(slot_records), src_ptr, int(n_alignments), C.byref(info)))
        return info

    def use_stream(self, stream_ptr):
        """queue everything on the caller's HIP stream.  None: the handle's own again.  0 -- what torch calls its default
        stream -- is the legacy null stream and is passed on as hipStreamLegacy (include/ipx.h: a null pointer means "the
        handle's own", so the null stream has to be NAMED; round 3 read 0 as None and ran beside the stream it was asked
        to run on)."""
        if stream_ptr is None:
            self._check(self.lib.ipx_use_stream(self.h, None))
        else:
            self._check(self.lib.ipx_use_stream(self.h, int(stream_ptr) or 1))

    def copy_async(self, dst_ptr, src_ptr, n_bytes):
        self._check(self.lib.ipx_copy_async(self.h, dst_ptr, src_ptr, int(n_bytes)))

    def finalize_raw(self, memory=L.IPX_MEM_HOST):
        t = L.Table()
        rc = self.lib.ipx_finalize(self.h, memory, C.byref(t))
        return rc, t

    def get_junctions(self):
        """The table as numpy arrays, rows in std::map order."""
        rc, t = self.finalize_raw(L.IPX_MEM_HOST)
        if rc and rc != L.IPX_EDATA:
            self._check(rc)
        n = t.n_junctions
        out = {k: _view(getattr(t, k), n, dt) for k, dt in _NP.items()}
        out["counts"] = _view(t.read_counts, 3 * n, np.uint64).reshape(n, 3)
        out["hist_begin"] = _view(t.hist_begin, n + 1, np.uint64) if n else np.zeros(1, np.uint64)
        out["hist_value"] = _view(t.hist_value, t.n_hist, np.uint16)
        out["hist_count"] = _view(t.hist_count, t.n_hist, np.uint32)
        out["n_records"] = t.n_records
        out["n_alignments"] = t.n_alignments
        if rc == L.IPX_EDATA:
            e = L.Error()
            self.lib.ipx_last_error(self.h, C.byref(e))
            out["error"] = IpxError(rc, e)
        return out

    def timing(self):
        a, b = C.c_float(0), C.c_float(0)
        self.lib.ipx_last_timing(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    def profile(self, enable=True):
        """Switch per-kernel HIP-event timing on/off (resets the sums)."""
        self._check(self.lib.ipx_profile(self.h, int(enable)))

    def profile_only(self, kernel):
        """Time the launches of one kernel class ("k_scan", "k_scatter", "k_reduce") and nothing else."""
        self._check(self.lib.ipx_profile_only(self.h, kernel.encode()))

    def profile_read(self):
        """-> {kernel name: (launches, total_ms)} since profile(True)."""
        arr = (L.KernelStat * 32)()
        n = C.c_uint32(0)
        self._check(self.lib.ipx_profile_read(self.h, arr, 32, C.byref(n)))
        return {arr[i].name.decode(): (int(arr[i].launches), float(arr[i].total_ms)) for i in range(min(n.value, 32))}

    def clear(self):
        self._check(self.lib.ipx_clear(self.h))

    def close(self):
        if getattr(self, "h", None):
            self.lib.ipx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _table_to_numpy(t):
    n = t.n_junctions
    out = {k: _view(getattr(t, k), n, dt) for k, dt in _NP.items()}
    out["counts"] = _view(t.read_counts, 3 * n, np.uint64).reshape(n, 3)
    out["hist_begin"] = _view(t.hist_begin, n + 1, np.uint64) if n else np.zeros(1, np.uint64)
    out["hist_value"] = _view(t.hist_value, t.n_hist, np.uint16)
    out["hist_count"] = _view(t.hist_count, t.n_hist, np.uint32)
    out["n_records"] = t.n_records
    out["n_alignments"] = t.n_alignments
    return out


class JunctionsExtractorGroup:
    """Several GPUs behind one extractor (ipx_group of include/ipx.h): every pushed batch is spread over the
    devices by read index, get_junctions() returns ONE table.  A device listed twice shares that GPU between
    two members (the rehearsal a one-GPU box allows)."""

    def __init__(self, devices, min_anchor_length=8, max_anchor_indel_size=0, min_intron_length=70,
                 max_intron_length=500000, junction_filter=L.IPX_SJ_NULL, strandness=L.IPX_UNSTRANDED,
                 excludes=0, skip_missing_targets=False, scope=L.IPX_SCOPE_TARGET, targets=(), genome=None):
        self.lib = L.load()
        p = L.Params(min_anchor_length, max_anchor_indel_size, min_intron_length, max_intron_length,
                     junction_filter, strandness, excludes, scope, int(bool(skip_missing_targets)), 0)
        devs = (C.c_int32 * len(devices))(*devices)
        g = C.c_void_p()
        rc = self.lib.ipx_group_create(C.byref(p), len(devices), devs, C.byref(g))
        if rc:
            raise RuntimeError("ipx_group_create failed with status %d" % rc)
        self.g = g
        self.next_read_index = 0
        if len(targets):
            arr = (C.c_char_p * len(targets))(*[n.encode() if isinstance(n, str) else n for n in targets])
            self._check(self.lib.ipx_group_set_targets(self.g, len(targets), arr, None))
        if genome is not None:
            names, seqs = genome
            n = len(names)
            arr = (C.c_char_p * n)(*[s.encode() for s in names])
            lens = (C.c_int64 * n)(*[int(t[2]) for t in seqs])
            two = (C.c_void_p * n)(*[t[0].ctypes.data for t in seqs])
            msk = (C.c_void_p * n)(*[t[1].ctypes.data for t in seqs])
            self._check(self.lib.ipx_group_set_genome_2bit(self.g, n, arr, lens, two, msk))

    def uses_rccl(self):
        return bool(self.lib.ipx_group_uses_rccl(self.g))

    def _check(self, rc):
        if rc:
            e = L.Error()
            self.lib.ipx_group_last_error(self.g, C.byref(e))
            raise IpxError(rc, e)

    def push(self, batch, base_read_index=None):
        n = len(batch["tid"])
        if base_read_index is None:
            base_read_index = self.next_read_index
        arrs = {k: np.ascontiguousarray(batch[k], dt) for k, dt in (
            ("tid", np.int32), ("pos", np.int32), ("flag", np.uint16), ("xs", np.uint8), ("nh", np.uint8),
            ("cigar_off", np.uint32), ("n_cigar", np.uint32), ("cigar", np.uint32))}
        b = L.Batch(n, len(arrs["cigar"]), base_read_index,
                    *[arrs[k].ctypes.data for k in ("tid", "pos", "flag", "xs", "nh", "cigar_off", "n_cigar", "cigar")],
                    L.IPX_MEM_HOST)
        self._check(self.lib.ipx_group_push_batch(self.g, C.byref(b)))
        self.next_read_index = base_read_index + n

    def push_member_device(self, member, n, n_words, ptrs, base_read_index):
        """one member's share from that member's device memory (ptrs: tid,pos,flag,xs,nh,cigar_off,n_cigar,cigar)"""
        b = L.Batch(n, n_words, base_read_index, *ptrs, L.IPX_MEM_DEVICE)
        self._check(self.lib.ipx_group_push_member(self.g, member, C.byref(b)))

    def get_junctions(self):
        t = L.Table()
        rc = self.lib.ipx_group_finalize(self.g, C.byref(t))
        if rc and rc != L.IPX_EDATA:
            self._check(rc)
        out = _table_to_numpy(t)
        if rc == L.IPX_EDATA:
            e = L.Error()
            self.lib.ipx_group_last_error(self.g, C.byref(e))
            out["error"] = IpxError(rc, e)
        return out

    def clear(self):
        self._check(self.lib.ipx_group_clear(self.g))

    def close(self):
        if getattr(self, "g", None):
            self.lib.ipx_group_destroy(self.g)
            self.g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def expand_histogram(table):
    """CSR histogram -> the sorted uint16 offset multiset, concatenated per junction
    (what Junction::read_offsets holds after calculate_confidence sorted it)."""
    return np.repeat(table["hist_value"], table["hist_count"].astype(np.int64))


def offsets_per_junction(table):
    hb = table["hist_begin"].astype(np.int64)
    csum = np.concatenate([[0], np.cumsum(table["hist_count"].astype(np.int64))])
    return csum[hb[1:]] - csum[hb[:-1]]
