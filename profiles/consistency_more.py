"""Run-to-run comparison (see profiles/consistency.py) of the paths that profiles/consistency.py does not take:
 X  the multi-GPU exchange path with a process group of one rank (slots, all_to_all over RCCL, adopt, finalize, global sjN)
 B  a BAM file decoded on the device (ipx_decode_bgzf: inflate, CRC, record chains, fields, CIGARs) and the table of its batch
 C  the command line on that file, device decode on, outputs by md5
usage (on the GPU box): python profiles/consistency_more.py [X B C] [--rounds N]"""
import argparse, hashlib, os, struct, subprocess, sys, tempfile, time
import ctypes as C
import numpy as np
import torch            # before libipx.so
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from intronprospector_amd import synth, _lib as L
import intronprospector_amd as ipa


def sha(arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def report(kind, seen, rounds, t0):
    print(kind, "%d rounds, %d distinct results %s (%.0f s)" % (rounds, len(seen), sorted(seen.values(), reverse=True), time.time() - t0), flush=True)
    return len(seen) == 1


def exchange(rounds):
    import torch.distributed as dist
    from intronprospector_amd import distributed
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    b, names, _ = bench.make_workload("S", 10_000_000, synth.SEED_S)
    _, ptrs = up = bench.upload(b, dev)
    n, nw = len(b["tid"]), len(b["cigar"])
    ex = ipa.JunctionsExtractor(targets=names, device=0)
    xch = distributed.RecordExchanger(dev, comm_device=dev)
    seen, t0 = {}, time.time()
    for r in range(rounds):
        while True:
            ex.clear()
            ex.push_device(n, nw, ptrs, base_read_index=0)
            try:
                table, ijunc = xch.run(ex, n)
                break
            except distributed.SlotOverflow:
                pass
        t = distributed.table_to_torch(ex, table, "cpu")
        d = sha([t[k].numpy() for k in sorted(t)] + [ijunc.cpu().numpy()])
        seen[d] = seen.get(d, 0) + 1
    ex.close()
    dist.destroy_process_group()
    return report("X (exchange, one rank)", seen, rounds, t0)


def make_bam(path, n_short=150_000):
    from tests import bam_encode
    from tests.test_gpu_inflate import _bgzf_file
    short, names, _ = synth.short_reads(n_short, seed=21, n_targets=3, target_len=40_000_000, n_sites=20_000)
    long_, _, _ = synth.long_reads(200, seed=22, n_targets=3, target_len=40_000_000, n_transcripts=20)
    batch = synth.concat_batches([short, long_])
    order = np.lexsort((batch["pos"], batch["tid"]))
    batch = synth.permute_batch(batch, order)
    rec, sb, sf, nw = bam_encode.records(batch, seed=23)
    n = len(batch["tid"])
    rb = np.zeros(n + 1, np.uint64)
    off = 0
    for i in range(n):
        rb[i] = off
        off += 4 + struct.unpack("<I", rec[off:off + 4].tobytes())[0]
    rb[n] = off
    text = b"@HD\tVN:1.6\tSO:coordinate\n" + b"".join(b"@SQ\tSN:%s\tLN:40000000\n" % nm.encode() for nm in names)
    header = b"BAM\1" + struct.pack("<I", len(text)) + text + struct.pack("<I", len(names)) + \
        b"".join(struct.pack("<I", len(nm) + 1) + nm.encode() + b"\0" + struct.pack("<I", 40000000) for nm in names)
    data = _bgzf_file(header, rec.tobytes(), rb)
    open(path, "wb").write(data)
    return data, len(header), names, n


def bam_decode(rounds, data, records_begin, names):
    from tests.test_gpu_inflate import _decode_on_device
    ex = ipa.JunctionsExtractor(targets=names)
    seen, t0 = {}, time.time()
    for r in range(rounds):
        ex.clear()
        rc, msg, got = _decode_on_device(ex, data, records_begin)
        assert rc == 0, msg
        dev, b = got
        n = int(b.n_alignments)
        s = L.Batch(n, b.n_cigar_words, 0, b.tid, b.pos, b.flag, b.xs, b.nh, b.cigar_off, b.n_cigar, b.cigar, L.IPX_MEM_DEVICE)
        ex._check(ex.lib.ipx_push_batch(ex.h, C.byref(s)))
        tab = ex.get_junctions()
        d = sha([dev[k] for k in sorted(dev)] + [tab[k] for k in sorted(tab) if isinstance(tab[k], np.ndarray)])
        seen[d] = seen.get(d, 0) + 1
    ex.close()
    return report("B (BAM file decoded on the device)", seen, rounds, t0)


def cli(rounds, path):
    exe = os.path.join(ROOT, "intronprospector_amd", "bin", "intronProspector")
    seen, t0 = {}, time.time()
    with tempfile.TemporaryDirectory() as d:
        for r in range(rounds):
            env = dict(os.environ, IPX_DEVICE_INFLATE="1")
            subprocess.run([exe, "-c", d + "/c.tsv", "-j", d + "/j.bed", "-n", d + "/i.bed", path], check=True, env=env)
            h = sha([np.frombuffer(open(d + "/" + f, "rb").read(), np.uint8) for f in ("c.tsv", "j.bed", "i.bed")])
            seen[h] = seen.get(h, 0) + 1
    return report("C (command line, device decode)", seen, rounds, t0)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("kinds", nargs="*", default=["B", "C", "X"])
    ap.add_argument("--rounds", type=int, default=30)
    a = ap.parse_args()
    ok = True
    if "B" in a.kinds or "C" in a.kinds:
        path = os.path.join(tempfile.mkdtemp(), "s.bam")
        data, rb, names, n = make_bam(path)
        print("BAM file: %d alignments, %d bytes" % (n, len(data)), flush=True)
        if "B" in a.kinds:
            ok = bam_decode(a.rounds, data, rb, names) and ok
        if "C" in a.kinds:
            ok = cli(max(a.rounds // 3, 5), path) and ok
    if "X" in a.kinds:
        ok = exchange(a.rounds) and ok
    sys.exit(0 if ok else 1)
