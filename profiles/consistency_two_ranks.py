"""Run-to-run comparison of the exchange path with TWO ranks sharing one GPU (gloo for the collectives): the partition
into two key ranges, the slots, the adoption and the global sjN, thirty times over the same input -- every rank's table
and numbering must be the same each time.  usage (GPU box): python profiles/consistency_two_ranks.py [rounds]"""
import hashlib, os, socket, sys
import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, port, rounds, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import intronprospector_amd as ipa
    from intronprospector_amd import synth, distributed
    batch, names, _ = synth.short_reads(4_000_000, seed=77)
    n = len(batch["tid"])
    cuts = np.linspace(0, n, world + 1).astype(int)
    lo, hi = int(cuts[rank]), int(cuts[rank + 1])
    mine = synth.slice_batch(batch, lo, hi)
    ex = ipa.JunctionsExtractor(targets=names)
    xch = distributed.RecordExchanger(torch.device("cuda", 0), comm_device=torch.device("cpu"))
    seen = {}
    for r in range(rounds):
        while True:
            ex.clear()
            ex.push(mine, base_read_index=lo)
            try:
                table, ijunc = xch.run(ex, hi - lo)
                break
            except distributed.SlotOverflow:
                pass
        t = distributed.table_to_torch(ex, table, "cpu")
        h = hashlib.sha256()
        for k in sorted(t):
            h.update(np.ascontiguousarray(t[k].numpy()).tobytes())
        h.update(np.ascontiguousarray(ijunc.cpu().numpy()).tobytes())
        d = h.hexdigest()
        seen[d] = seen.get(d, 0) + 1
    ex.close()
    q.put((rank, sorted(seen.values(), reverse=True), int(table.n_junctions)))
    dist.destroy_process_group()


if __name__ == "__main__":
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, 2, port, rounds, q)) for r in range(2)]
    for p in ps:
        p.start()
    out = sorted(q.get(timeout=600) for _ in ps)
    for p in ps:
        p.join()
    ok = all(len(o[1]) == 1 for o in out)
    for rank, counts, nj in out:
        print("rank %d: %d rounds, %d distinct results %s, %d junctions in its key range" % (rank, rounds, len(counts), counts, nj))
    sys.exit(0 if ok else 1)
