"""Run N worker processes and fail fast when one dies (instead of waiting on a queue)."""
import queue as _queue
import time

import torch.multiprocessing as mp


def run_workers(target, world, args, timeout=300):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=(r, world) + tuple(args) + (q,)) for r in range(world)]
    for p in procs:
        p.start()
    deadline = time.time() + timeout
    result = None
    try:
        while time.time() < deadline:
            try:
                result = q.get(timeout=1.0)
                break
            except _queue.Empty:
                dead = [p for p in procs if p.exitcode not in (None, 0)]
                if dead:
                    raise AssertionError("worker exited with code %s before reporting" % dead[0].exitcode)
        if result is None:
            raise AssertionError("workers did not report within %d s" % timeout)
        for p in procs:
            p.join(timeout=60)
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
    for p in procs:
        assert p.exitcode == 0, "worker exit code %s" % p.exitcode
    return result
