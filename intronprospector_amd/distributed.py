"""Combining the partial junction tables of contiguous shards.

Alignments shard naturally: rank g holds a contiguous range of the run's
alignments and reduces it on its own GPU (ipx_finalize_shard).  The only
exchange is on the small junction tables, over torch.distributed ("nccl" is RCCL
over xGMI on MI355X; "gloo" in the CPU tests):

  1. all_gather of the shards' key lists -> sorted union -> dense index
  2. all_reduce(MIN) of the packed creation order: the global creator of a key
     is the earliest creating record over all shards
  3. every shard picks what it contributes to each key -- nothing (shard lies
     before the creator), its from-creator tallies (it holds the creator) or its
     all-records tallies (it lies after) -- then all_reduce SUM (counts), MIN
     (anchor start), MAX (anchor end, motif word)
  4. all_gather of the contributing histogram bins, summed per (key, offset)

This is NOT Junction::merge (junctions.cc:76-105) of the reference's merge tool,
whose max-of-confidences rule differs from a single run over the same input.
Everything here is plumbing on tables of at most a few million rows; torch ops
are used so the same code runs on CPU tensors under gloo.
"""
import ctypes as C
import os

import torch
import torch.distributed as dist

BIG = (1 << 62)
ERR_NONE = (1 << 64) - 1        # ipx_error.order of "no error"


class ShardError:
    """What one rank found wrong with its share of the input: `order` places it in the run's input order
    (ipx_error.order; 0 for a failure that is not about the data, so that it wins), `blob` is the rank's
    ipx_error as bytes, `status` its ipx status."""

    def __init__(self, order=ERR_NONE, blob=b"", status=0):
        self.order, self.blob, self.status = int(order), bytes(blob), int(status)

    @classmethod
    def from_ipx(cls, exc):
        import ctypes as C
        blob = bytes((C.c_char * C.sizeof(exc.err)).from_buffer_copy(exc.err)) if getattr(exc, "err", None) is not None else b""
        from . import _lib as L
        return cls(exc.order if exc.status == L.IPX_EDATA else 0, blob, exc.status)

    def raise_(self):
        from . import _lib as L
        from .extractor import IpxError
        e = L.Error.from_buffer_copy(self.blob.ljust(__import__("ctypes").sizeof(L.Error), b"\0"))
        raise IpxError(self.status or e.status or L.IPX_EDEVICE, e)


def _all_gather_var(t, group=None, local_error=None, pad_value=0, stacked=False):
    """all_gather of 1-D/2-D tensors whose first dimension differs per rank.  stacked: return the ranks'
    arrays as ONE tensor [world, longest, ...] whose rows are filled up with pad_value, and the lengths.

    local_error (a ShardError or None on EVERY rank that passes the argument): the ranks' sizes travel
    with the ranks' error orders; if some rank has an error, every rank raises the one that comes first
    in input order -- the reference throws at the first bad alignment its loop meets
    (junctions_extractor.cc:73-74,312-313) -- before anybody enters the data collective, so nobody is
    left waiting in it."""
    world = dist.get_world_size(group)
    head = [t.shape[0]]
    if local_error is not None:
        head += [local_error.order >> 32, local_error.order & 0xffffffff, local_error.status]
    n = torch.tensor(head, dtype=torch.int64, device=t.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    rows = torch.stack(sizes).cpu().tolist()         # one read-back
    if local_error is not None:
        _raise_earliest([(r[1] << 32) | r[2] for r in rows], local_error, group, t.device)
    sizes = [r[0] for r in rows]
    m = max(max(sizes), 1)
    pad = torch.full((m,) + tuple(t.shape[1:]), pad_value, dtype=t.dtype, device=t.device)
    pad[:t.shape[0]] = t
    if stacked:
        out = torch.empty((world,) + tuple(pad.shape), dtype=t.dtype, device=t.device)
        dist.all_gather(list(out.unbind(0)), pad, group=group)
        return out, sizes
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return [b[:s] for b, s in zip(bufs, sizes)]


def _raise_earliest(orders, local_error, group, device):
    """orders: every rank's error order (ERR_NONE = fine).  Returns if all are fine; otherwise the rank
    with the smallest order (ties: the lowest rank) broadcasts its ipx_error and EVERY rank raises it."""
    import ctypes as C
    from . import _lib as L
    best = min(range(len(orders)), key=lambda r: (orders[r], r))
    if orders[best] == ERR_NONE:
        return
    size = C.sizeof(L.Error) + 8
    rank = dist.get_rank(group)
    buf = torch.zeros(size, dtype=torch.uint8, device=device)
    if rank == best:
        raw = local_error.status.to_bytes(8, "little") + local_error.blob.ljust(C.sizeof(L.Error), b"\0")
        buf = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)
    src = dist.get_global_rank(group, best) if group is not None else best
    dist.broadcast(buf, src=src, group=group)
    raw = bytes(buf.cpu().numpy().tobytes())
    ShardError(orders[best], raw[8:], int.from_bytes(raw[:8], "little")).raise_()


def agree_on_error(local_error, group=None, device="cpu"):
    """Stand-alone form of the agreement (one tiny all_gather): returns when no rank has an error, else every
    rank raises the globally earliest one."""
    _all_gather_var(torch.zeros(0, dtype=torch.int64, device=device), group, local_error or ShardError())


def _keys(tab):
    """(n, 2) int64 rows (target << 32 | start, end): row-lexicographic order is the
    table's key order."""
    k0 = (tab["target"].to(torch.int64) << 32) | tab["intron_start"].to(torch.int64)
    return torch.stack([k0, tab["intron_end"].to(torch.int64)], dim=1)


def _find_rows(union, keys):
    """index of each row of `keys` in the sorted unique `union` (all present)."""
    # union rows are unique and sorted; compare through a single int128-free trick:
    # search on column 0, then resolve equal-k0 runs with column 1
    lo = torch.searchsorted(union[:, 0].contiguous(), keys[:, 0].contiguous(), right=False)
    hi = torch.searchsorted(union[:, 0].contiguous(), keys[:, 0].contiguous(), right=True)
    idx = lo.clone()
    # runs of equal (target,start) are short (alternative ends); walk them
    maxrun = int((hi - lo).max().item()) if keys.shape[0] else 0
    found = torch.zeros(keys.shape[0], dtype=torch.bool, device=keys.device)
    for step in range(maxrun):
        cand = torch.clamp(lo + step, max=union.shape[0] - 1)
        hit = (~found) & (lo + step < hi) & (union[cand, 1] == keys[:, 1])
        idx = torch.where(hit, cand, idx)
        found |= hit
    assert bool(found.all()) or keys.shape[0] == 0
    return idx


def merge_shard_tables(fc, al, shard_begin_order, group=None, local_error=None):
    """fc / al: the shard's from-creator and all-records tables as dicts of torch
    tensors (target,intron_start,intron_end,anchor_start,anchor_end,strand,motif,
    counts[n,3],creator_order,hist_begin[n+1],hist_value,hist_count).
    shard_begin_order: smallest creator_order any record of this shard can have
    (first global read index << 20).  Returns the merged table (same on all ranks),
    rows in key order, ijunc = rank of creation.
    local_error: this shard's ShardError (or None on every rank): all ranks raise the earliest."""
    dev = al["target"].device
    k_al, k_fc = _keys(al), _keys(fc)
    # 1. union of keys (the shards' errors, if any, travel with the sizes: see _all_gather_var)
    parts = _all_gather_var(k_al, group, local_error)
    allk = torch.cat(parts, dim=0)
    if allk.shape[0] == 0:
        empty = {k: v[:0] for k, v in al.items() if k not in ("hist_begin",)}
        empty["hist_begin"] = torch.zeros(1, dtype=torch.int64, device=dev)
        empty["ijunc"] = torch.zeros(0, dtype=torch.int64, device=dev)
        return empty
    union = torch.unique(allk, dim=0)                      # sorted rows
    U = union.shape[0]
    i_al = _find_rows(union, k_al)
    i_fc = _find_rows(union, k_fc)

    # 2. global creator: min of (creator_order << 8 | strand)
    packed = torch.full((U,), BIG, dtype=torch.int64, device=dev)
    packed[i_fc] = (fc["creator_order"].to(torch.int64) << 8) | fc["strand"].to(torch.int64)
    mine = packed.clone()
    dist.all_reduce(packed, op=dist.ReduceOp.MIN, group=group)
    exists = packed < BIG
    g_order = packed >> 8

    # 3. contributions
    is_creator = exists & (mine == packed)
    after = exists & (g_order < int(shard_begin_order))    # created by an earlier shard
    counts = torch.zeros((U, 3), dtype=torch.int64, device=dev)
    amin = torch.full((U,), BIG, dtype=torch.int64, device=dev)
    amax = torch.zeros((U,), dtype=torch.int64, device=dev)
    motif = torch.zeros((U,), dtype=torch.int64, device=dev)
    sel_fc = is_creator[i_fc]
    sel_al = after[i_al]
    for tab, idx, sel in ((fc, i_fc, sel_fc), (al, i_al, sel_al)):
        rows = idx[sel]
        counts[rows] = tab["counts"][sel].to(torch.int64)
        amin[rows] = tab["anchor_start"][sel].to(torch.int64)
        amax[rows] = tab["anchor_end"][sel].to(torch.int64)
    motif[i_al] = al["motif"].to(torch.int64)
    dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(amin, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(amax, op=dist.ReduceOp.MAX, group=group)
    dist.all_reduce(motif, op=dist.ReduceOp.MAX, group=group)

    # 4. histogram bins of the contributing rows
    def bins_of(tab, idx, sel):
        hb = tab["hist_begin"].to(torch.int64)
        nb = hb[1:] - hb[:-1]
        row_of_bin = torch.repeat_interleave(torch.arange(nb.shape[0], device=dev), nb)
        keep = sel[row_of_bin]
        u = idx[row_of_bin[keep]]
        return (u << 16) | tab["hist_value"][keep].to(torch.int64), tab["hist_count"][keep].to(torch.int64)
    b1, c1 = bins_of(fc, i_fc, sel_fc)
    b2, c2 = bins_of(al, i_al, sel_al)
    bk = torch.cat([b1, b2])
    bc = torch.cat([c1, c2])
    bk_all = torch.cat(_all_gather_var(bk, group))
    bc_all = torch.cat(_all_gather_var(bc, group))
    ub, inv = torch.unique(bk_all, return_inverse=True)    # sorted by (key index, offset)
    hc = torch.zeros(ub.shape[0], dtype=torch.int64, device=dev)
    hc.index_add_(0, inv, bc_all)

    # compact to the junctions that exist
    rows = torch.nonzero(exists, as_tuple=False).squeeze(1)
    new_of_old = torch.cumsum(exists.to(torch.int64), 0) - 1
    bin_row = new_of_old[ub >> 16]
    n = rows.shape[0]
    hist_begin = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    hist_begin[1:] = torch.cumsum(torch.bincount(bin_row, minlength=n), 0)
    order = g_order[rows]
    ijunc = torch.empty(n, dtype=torch.int64, device=dev)
    ijunc[torch.argsort(order)] = torch.arange(n, dtype=torch.int64, device=dev)
    return {
        "target": union[rows, 0] >> 32, "intron_start": union[rows, 0] & 0xffffffff, "intron_end": union[rows, 1],
        "anchor_start": amin[rows], "anchor_end": amax[rows], "strand": packed[rows] & 0xff,
        "motif": motif[rows], "counts": counts[rows], "creator_order": order, "ijunc": ijunc,
        "hist_begin": hist_begin, "hist_value": ub & 0xffff, "hist_count": hc,
    }


_FIELDS = (("target", torch.int32, 4), ("intron_start", torch.int32, 4), ("intron_end", torch.int32, 4),
           ("anchor_start", torch.int32, 4), ("anchor_end", torch.int32, 4), ("strand", torch.uint8, 1),
           ("motif", torch.int16, 2), ("creator_order", torch.int64, 8))


def table_to_torch(ex, t, device):
    """Copy an ipx_table (host or device memory) into torch tensors on `device`."""
    n, nh = int(t.n_junctions), int(t.n_hist)
    out = {}

    def grab(ptr, count, dtype, width):
        x = torch.empty(count, dtype=dtype, device=device)
        if count:
            ex._check(ex.lib.ipx_copy(ex.h, x.data_ptr(), ptr, count * width))
        return x
    for name, dt, w in _FIELDS:
        x = grab(getattr(t, name), n, dt, w)
        if dt == torch.int32:
            x = x.to(torch.int64) & 0xffffffff
        elif dt == torch.int16:
            x = x.to(torch.int64) & 0xffff
        out[name] = x
    out["counts"] = grab(t.read_counts, 3 * n, torch.int64, 8).reshape(n, 3)
    out["hist_begin"] = grab(t.hist_begin, n + 1, torch.int64, 8) if n else torch.zeros(1, dtype=torch.int64, device=device)
    out["hist_value"] = grab(t.hist_value, nh, torch.int16, 2).to(torch.int64) & 0xffff
    out["hist_count"] = grab(t.hist_count, nh, torch.int32, 4).to(torch.int64) & 0xffffffff
    return out


def empty_table(device):
    z = lambda dt: torch.zeros(0, dtype=dt, device=device)
    t = {k: z(torch.int64) for k in ("target", "intron_start", "intron_end", "anchor_start", "anchor_end", "strand", "motif",
                                     "creator_order", "hist_value", "hist_count")}
    t["counts"] = torch.zeros((0, 3), dtype=torch.int64, device=device)
    t["hist_begin"] = torch.zeros(1, dtype=torch.int64, device=device)
    return t


class RecordExchanger:
    """Multi-GPU by key range: every rank scans its own alignments, the junction RECORDS are
    exchanged once (all_to_all over RCCL) so that each rank owns a contiguous key range, and
    each rank sorts + reduces its range with the ordinary single-GPU finalize.  The table stays
    distributed (rank order = key order); ijunc is made global with one all_gather of the
    creation orders.  Cheaper than merging tables whenever the tables are not much smaller
    than the records (bench workload: 8 M histogram bins per 10 M records)."""

    N_SAMPLES = 512
    # Splitters are key-range boundaries; they only decide the load balance, never the result.  A stream of
    # batches from one input keeps its key distribution, so the splitters of the first batch are kept and
    # drawn again only after an exchange in which some rank received more than REBALANCE x the mean.
    REBALANCE = 1.25
    SLOT_MARGIN = 1.25                   # slots and the order gather: this much above the previous batch's need

    def __init__(self, device, group=None, comm_device=None, reuse_splitters=True):
        self.device, self.group = device, group
        # gloo rehearsal on one GPU: collectives on CPU copies
        self.comm_device = comm_device if comm_device is not None else device
        self.reuse_splitters = reuse_splitters
        self._splitters = None
        self.resampled = 0               # how often the splitters were drawn (statistics, tests)
        self.timed = False               # profiles/rehearse_exchange_timing.py: synchronise and time every stage
        self.stage_ms = {}
        self.slotted = True              # False: always the exact exchange (tests compare both)
        self._slot_records = None        # records per slot of the slotted exchange (None: not known yet)
        self._send = self._recv = None
        self._spill = self._tails = None
        self._stream = None              # the slotted exchange's own stream (see _run_slotted)
        self.slot_overflows = 0
        self.strict = True               # look at the ranks' finalize errors at the end of every run (False: check_last())
        self._last_trouble = None

    def _lap(self, name, t0):
        import time
        if not self.timed:
            return t0
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        t1 = time.perf_counter()
        self.stage_ms[name] = self.stage_ms.get(name, 0.0) + (t1 - t0) * 1e3
        return t1

    def _to_comm(self, t):
        return t if t.device == self.comm_device else t.to(self.comm_device)

    def run(self, ex, n_alignments_local=0):
        """-> (ipx_table of this rank's key range, device memory; global ijunc per row as a tensor).

        Two forms.  The EXACT exchange (the first batch, and whenever the splitters were just drawn): the ranks tell
        each other their per-peer counts, all_to_all with those splits, orders gathered at their true lengths -- five
        host round trips a step.  The SLOTTED exchange (every later batch): sizes are taken from the batch before
        with a margin, parts travel in fixed-capacity slots whose headers carry the true counts, an overflow flag and
        the ranks' errors (ipx_records_partition_slots), so ONE all_to_all with equal splits needs no counts collective
        and no read-back before it; the creation orders are gathered at a fixed length too.  A batch that outgrows
        its slots is completed by a SECOND all_to_all (round 4): the partition kept every part's tail in a spill, the
        headers carry every sender's true counts, so all ranks know the tails' lengths without asking; the slots are
        raised for the next batch.  (Rounds 2-3 raised SlotOverflow here and the caller scanned the batch again.)"""
        if (self._splitters is None or not self.reuse_splitters or self._slot_records is None or not self.slotted
                or len(self._splitters) != dist.get_world_size(self.group) - 1):
            return self._run_exact(ex, n_alignments_local)
        if os.environ.get("IPX_DEBUG_PLAN"):
            import sys
            print("ipx: exchange in slots of %d records" % self._slot_records, file=sys.stderr)
        return self._run_slotted(ex, n_alignments_local)

    def check_last(self):
        """Did some rank's finalize fail in the last slotted run?  Then every rank raises the earliest error of the run
        (one small read-back; with strict = False the caller decides when to pay for it)."""
        if self._last_trouble is None:
            return
        worst, failed = self._last_trouble
        self._last_trouble = None
        if int(worst.item()) != torch.iinfo(torch.int64).max:
            agree_on_error(failed, self.group, self.comm_device)

    def _run_slotted(self, ex, n_alignments_local):
        from . import _lib as L
        from .extractor import IpxError
        import time
        world = dist.get_world_size(self.group)
        rank = dist.get_rank(self.group)
        t0 = time.perf_counter()
        S = self._slot_records
        failed = None
        try:
            ex.records_pending()                  # completes a pending push: a data error of this rank's scan surfaces at finalize
        except IpxError as exc:
            failed = ShardError.from_ipx(exc)
        # Everything below is queued on ONE stream -- the handle's kernels, torch's ops, the collectives' waits -- and that
        # stream is the exchanger's OWN (round 4).  Round 3 took torch's current stream, which nobody had set: the legacy
        # null stream, handle 0, which ipx_use_stream reads as "the handle's own non-blocking stream again" -- so the
        # handle's kernels ran beside torch's, ordered against nothing: the all_to_all could read `send` while
        # k_part_scatter wrote it, the adoption could read `recv` before the collective had filled it (the advisor's
        # finding; no test saw it because every step resent the same batch).  A stream of our own has a handle that is
        # not 0, is ordered after what the caller queued before (wait_stream), and the caller's stream waits for it at
        # the end.
        staged = self.comm_device != self.device          # gloo rehearsal: the collectives run on CPU copies
        if self._stream is None:
            self._stream = torch.cuda.Stream(self.device)
        caller_stream = torch.cuda.current_stream(self.device)
        self._stream.wait_stream(caller_stream)
        with torch.cuda.stream(self._stream):
            try:
                return self._slotted_on_stream(ex, n_alignments_local, world, S, failed, staged, t0)
            finally:
                ex.use_stream(None)
                caller_stream.wait_stream(self._stream)

    def _slotted_on_stream(self, ex, n_alignments_local, world, S, failed, staged, t0):
        from . import _lib as L
        from .extractor import IpxError
        stream = torch.cuda.current_stream(self.device)
        assert stream.cuda_stream != 0
        ex.use_stream(stream.cuda_stream)
        need = world * S * 32
        if self._send is None or self._send.numel() < need:
            self._send = torch.empty(need, dtype=torch.uint8, device=self.device)
            self._recv = torch.empty(need, dtype=torch.uint8, device=self.device)
        send, recv = self._send[:need], self._recv[:need]
        # the spill: room for every record this rank holds (the worst case: all of them beyond their slots)
        n_local = ex.records_pending() if failed is None else 0
        if self._spill is None or self._spill.numel() < max(n_local, 1) * 32:
            self._spill = torch.empty(max(n_local, 1) * 32 * 5 // 4, dtype=torch.uint8, device=self.device)
        try:
            ex.records_partition_slots(self._splitters, S, send.data_ptr(),
                                       failed.order if failed else ERR_NONE, failed.status if failed else 0,
                                       spill_ptr=self._spill.data_ptr(), spill_records=self._spill.numel() // 32)
        except IpxError as exc:                  # (arguments, memory: not about the data) -- empty slots that say so
            failed = ShardError.from_ipx(exc)
            send.zero_()
            hdr = torch.tensor([0, 0, 0, 0, 0, failed.status, failed.order & 0xffffffff, failed.order >> 32],
                               dtype=torch.int64, device=self.device).to(torch.int32).view(torch.uint8)
            for p in range(world):
                send[p * S * 32:p * S * 32 + 32] = hdr
        t0 = self._lap("partition", t0)
        if staged:
            recv_c = torch.empty(need, dtype=torch.uint8, device=self.comm_device)
            dist.all_to_all_single(recv_c, send.to(self.comm_device), group=self.group)
            recv.copy_(recv_c)
        else:
            dist.all_to_all_single(recv, send, group=self.group)         # equal splits: no counts, no host
        t0 = self._lap("all_to_all", t0)
        info = ex.records_adopt_slots(world, S, recv.data_ptr(), n_alignments_local)
        t0 = self._lap("adopt", t0)
        # (what the headers said is the same on every rank -- every sender's flags reach every receiver -- so all
        # ranks take the same branch here without talking to each other)
        if info.error_order != ERR_NONE or (failed is not None):
            agree_on_error(failed, self.group, self.comm_device)
        if info.overflow:
            self.slot_overflows += 1
            self._exchange_tails(ex, world, S, recv)
            t0 = self._lap("tails", t0)
        # keep the slots a margin above what the data needs (they shrink slowly, grow at once)
        want = int(info.max_part * self.SLOT_MARGIN) + 64
        self._slot_records = want if want > S else max(want, int(S * 0.9))
        # the counts rows in the headers are the whole counts matrix: every rank sees the same imbalance
        if world > 1 and info.max_received > self.REBALANCE * max(info.all_received / world, 1.0):
            self._splitters = None                                     # draw new splitters next time
        table = None
        try:
            rc, table = ex.finalize_raw(L.IPX_MEM_DEVICE)
            if rc:
                ex._check(rc)
        except IpxError as exc:
            failed = ShardError.from_ipx(exc)
        t0 = self._lap("finalize", t0)
        # global sjN: creation orders gathered at a length every rank derives from the slot headers alone -- a table has
        # at most as many junctions as its rank received records, and the most any rank received is the same number
        # everywhere (info.max_received) -- so no lengths are exchanged.  Element 0 carries this rank's own trouble (a
        # data error its reduce found: the order of that error, else "none"); it is looked at after everything else of
        # the step has been queued (strict) or by the caller, later (check_last()).
        n = int(table.n_junctions) if failed is None else 0
        J = int(info.max_received) + 1
        cd = self.comm_device
        # (round 5: the sorted list and the ranking are libipx kernels, ipx_table_sorted_orders / ipx_table_global_ijunc --
        # rounds 2-4 built them here from two copies, a scatter by index, a batched searchsorted and a sum)
        mine = torch.empty(J, dtype=torch.int64, device=self.device)
        empty = L.Table()
        empty.memory = L.IPX_MEM_DEVICE
        tab = table if n else empty
        ex._check(ex.lib.ipx_table_sorted_orders(ex.h, C.byref(tab), mine.data_ptr() + 8, J - 1))
        mine[0] = min(failed.order, (1 << 62)) if failed is not None else torch.iinfo(torch.int64).max
        parts = torch.empty((world, J), dtype=torch.int64, device=cd)
        dist.all_gather_into_tensor(parts.view(-1), mine.to(cd), group=self.group)
        parts = parts.to(self.device)
        ijunc = torch.empty(n, dtype=torch.int32, device=self.device)
        if n:
            ex._check(ex.lib.ipx_table_global_ijunc(ex.h, C.byref(table), parts.data_ptr() + 8, world, J, J - 1, ijunc.data_ptr()))
        ijunc = ijunc.to(torch.int64)
        self._last_trouble = (parts[:, 0].min(), failed)
        if self.strict:
            self.check_last()
        t0 = self._lap("global_ijunc", t0)
        return table, ijunc

    def _exchange_tails(self, ex, world, S, recv):
        """The slots were too small for some part: bring over what the partition left in the spill.  Every header holds
        its sender's counts for ALL parts (ipx_exchange.hip: k_part_plan_slots), and every rank received one slot from
        every sender, so the whole counts matrix is here already -- one small read-back, no counts collective."""
        rank = dist.get_rank(self.group)
        header_records = 1 + (world + 7) // 8
        room = S - header_records
        rows = recv.view(world, S * 32)[:, 32:32 + 4 * world].contiguous().view(torch.int32).to(torch.int64).cpu()
        rows = rows & 0xffffffff                                      # (counts are unsigned 32-bit words)
        send_counts, recv_counts = tail_splits(rows, room, rank)
        n_send, n_recv = sum(send_counts), sum(recv_counts)
        if n_send * 32 > self._spill.numel():
            raise RuntimeError("record exchange: the spill is smaller than the tails (%d records)" % n_send)
        out = self._to_comm(self._spill[:n_send * 32])
        got = torch.empty(max(n_recv, 1) * 32, dtype=torch.uint8, device=self.comm_device)
        dist.all_to_all_single(got[:n_recv * 32], out, output_split_sizes=[x * 32 for x in recv_counts],
                               input_split_sizes=[x * 32 for x in send_counts], group=self.group)
        got = got if got.device == self.device else got.to(self.device)
        if n_recv:
            ex.records_adopt_more(n_recv, got.data_ptr())
        self._tails = got                    # (kept until the next run: the adoption is queued, not finished)

    def _run_exact(self, ex, n_alignments_local=0):
        from . import _lib as L
        import numpy as np
        world = dist.get_world_size(self.group)
        rank = dist.get_rank(self.group)
        import time
        t0 = time.perf_counter()
        n_local = ex.records_pending()
        # 1. common splitters from everybody's samples (kept from the previous run while that stayed balanced;
        #    every rank sees the same counts matrix, so all ranks decide alike)
        if self._splitters is None or not self.reuse_splitters or len(self._splitters) != world - 1:
            smp = ex.records_sample(self.N_SAMPLES).astype(np.int64)        # ~0 (no records) becomes -1
            t = torch.from_numpy(smp).to(self.comm_device)
            bufs = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(bufs, t, group=self.group)
            allk = torch.cat(bufs)
            allk = torch.sort(allk[allk >= 0]).values
            if allk.numel():
                q = (torch.arange(1, world, device=allk.device) * allk.numel()) // world
                self._splitters = allk[q].cpu().numpy().astype(np.uint64)
            else:
                self._splitters = np.full(world - 1, np.iinfo(np.int64).max, np.uint64)
            self.resampled += 1
        splitters = self._splitters
        t0 = self._lap("splitters", t0)
        # 2. partition my records by key range
        from .extractor import IpxError
        failed = None                    # a ShardError once something went wrong on THIS rank
        send = torch.empty(max(n_local, 1) * 32, dtype=torch.uint8, device=self.device)
        try:
            counts = ex.records_partition(splitters, send.data_ptr())
        except IpxError as exc:
            failed = ShardError.from_ipx(exc)
            counts = np.zeros(world, np.uint64)
        t0 = self._lap("partition", t0)
        # 3. who sends how much to whom (and whether everybody got this far), then the exchange itself
        c = torch.from_numpy(np.concatenate([counts.astype(np.int64), [1 if failed else 0]])).to(self.comm_device)
        cm = [torch.empty_like(c) for _ in range(world)]
        dist.all_gather(cm, c, group=self.group)
        cmat = torch.stack(cm).cpu()                                   # one read-back, not one per rank
        if int(cmat[:, world].sum()):                                  # some rank failed: all leave together
            agree_on_error(failed, self.group, self.comm_device)
        cmat = cmat[:, :world]
        recv_counts = cmat[:, rank].tolist()
        n_recv = sum(recv_counts)
        per_rank = cmat.sum(dim=0).to(torch.float64)                   # records each rank receives
        if world > 1 and float(per_rank.max()) > self.REBALANCE * max(float(per_rank.mean()), 1.0):
            self._splitters = None                                     # draw new splitters next time
        # the next batches travel in slots sized from this one's largest part
        self._slot_records = int(int(cmat.max()) * self.SLOT_MARGIN) + 64
        t0 = self._lap("counts", t0)
        send_c = self._to_comm(send[:n_local * 32])
        recv_c = torch.empty(max(n_recv, 1) * 32, dtype=torch.uint8, device=self.comm_device)
        dist.all_to_all_single(recv_c[:n_recv * 32], send_c, output_split_sizes=[x * 32 for x in recv_counts],
                               input_split_sizes=[int(x) * 32 for x in counts.tolist()], group=self.group)
        recv = recv_c if recv_c.device == self.device else recv_c.to(self.device)
        if recv.device.type == "cuda":
            torch.cuda.synchronize(recv.device)
        t0 = self._lap("all_to_all", t0)
        # 4. the key range is now local: ordinary finalize.  A data error (IPX_EDATA: an alignment the reference
        #    throws at) or a failure is not raised here but agreed on in step 5's first collective
        table = None
        try:
            ex.records_adopt(n_recv, recv.data_ptr(), n_alignments_local)
            t0 = self._lap("adopt", t0)
            rc, table = ex.finalize_raw(L.IPX_MEM_DEVICE)
            if rc:
                ex._check(rc)
        except IpxError as exc:
            failed = ShardError.from_ipx(exc)
        t0 = self._lap("finalize", t0)
        # 5. sjN numbering is global: rank of the creation order over all ranks' junctions
        n = int(table.n_junctions) if failed is None else 0
        # the local ijunc is the rank inside this table: scattering by it sorts the orders without a sort
        # (ipx_table_sorted_orders); the global rank is the number of smaller orders in every rank's (sorted) list
        # (ipx_table_global_ijunc)
        empty = L.Table()
        empty.memory = L.IPX_MEM_DEVICE
        co_sorted = torch.empty(n, dtype=torch.int64, device=self.device)
        ex._check(ex.lib.ipx_table_sorted_orders(ex.h, C.byref(table if n else empty), co_sorted.data_ptr(), n))
        torch.cuda.synchronize(self.device)      # (this form runs on the handle's own stream, torch's ops on torch's: the first batch only)
        # (rows filled up with the largest order: they stay sorted and the filling counts for nobody)
        parts, _ = _all_gather_var(self._to_comm(co_sorted), self.group, failed or ShardError(),
                                       pad_value=torch.iinfo(torch.int64).max, stacked=True)
        parts = parts.to(self.device).contiguous()
        ijunc = torch.empty(n, dtype=torch.int32, device=self.device)
        if n:
            ex._check(ex.lib.ipx_table_global_ijunc(ex.h, C.byref(table), parts.data_ptr(), int(parts.shape[0]), int(parts.shape[1]),
                                                    int(parts.shape[1]), ijunc.data_ptr()))
            torch.cuda.synchronize(self.device)
        ijunc = ijunc.to(torch.int64)
        t0 = self._lap("global_ijunc", t0)
        self._keep = (send, recv)        # the table arrays do not alias these, but keep them until the next run
        return table, ijunc


def tail_splits(counts, room, rank):
    """counts[s][p]: records sender s has for part p (what the slot headers carry); room: records a slot holds.
    -> (what `rank` sends to every peer, what it receives from every peer) in the second all_to_all of an exchange whose
    slots overflowed: the part of every count beyond `room`."""
    tails = (counts.to(torch.int64) - int(room)).clamp_(min=0)
    return [int(x) for x in tails[rank].tolist()], [int(x) for x in tails[:, rank].tolist()]


class SlotOverflow(Exception):
    """Rounds 2-3: a part did not fit its slot and the caller scanned the batch again.  No longer raised (round 4: the tails
    follow in a second all_to_all, RecordExchanger._exchange_tails); kept so that callers written against it still import."""

    def __init__(self, slot_records):
        super().__init__("record exchange: a part outgrew its slot; slots raised to %d records" % slot_records)
        self.slot_records = slot_records


class TableMerger:
    """bench.py / multi-GPU driver: local shard tables -> merged table over RCCL."""

    def __init__(self, device, group=None):
        self.device, self.group = device, group

    def merge(self, ex, shard_first_read_index):
        from . import _lib as L
        import ctypes as C
        from .extractor import IpxError
        fc, al = L.Table(), L.Table()
        failed = None
        try:
            rc = ex.lib.ipx_finalize_shard(ex.h, L.IPX_MEM_DEVICE, C.byref(fc), C.byref(al))
            if rc:
                ex._check(rc)
            t_fc = table_to_torch(ex, fc, self.device)
            same = (fc.target == al.target)
            t_al = t_fc if same else table_to_torch(ex, al, self.device)
        except IpxError as exc:
            # every rank raises the earliest error of the run in merge_shard_tables' first collective
            failed = ShardError.from_ipx(exc)
            t_fc = t_al = empty_table(self.device)
        merged = merge_shard_tables(t_fc, t_al, int(shard_first_read_index) << 20, self.group, failed or ShardError())
        return merged, fc
