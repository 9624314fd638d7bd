"""A CPU model of the bucket sort's splitter selection (csrc/ipx_bsort.hip: k_bs_sample1 / k_bs_sample2 / k_bs_sort2) on
the keys of shuffled S -- 9.87 M unfolded (junction, read offset) records of 10 M synthetic short reads, 2.09 M distinct,
1 900 keys with 500 to 16 000 copies each.  Why: with IPX_BSORT_MIN=8000000 the device sorted that table by buckets in
one run (1.27 ms a step) and fell back to the radix passes every few steps in the next (1.34 ms), and the library does
not say which leaf overflowed.  The model does: the order of the records in memory varies from step to step (the scan's
stripe reservations), the sample is taken at fixed positions, so every step draws another sample; with ~1800 sample keys
per coarse range a range over the hot junctions needs 180-212 of its 255 splitter slots (a fenced key takes two), and
in one sort of eight more than 255: the rest is dropped and the range's last leaf holds 18 000 records.  With ~600 keys
a range the same ranges need 55-90.  What remains is a leaf of neighbouring keys with 500-1500 copies each that no fence
caught plus sampling noise: 3 917 records at worst in 50 sorts (the leaf's capacity is 4 096).

usage (CPU only; needs oracle/libipxo.so):  python profiles/experiments/bsort_splitter_model.py [trials]
Test infrastructure: the keys come from the oracle's table; nothing here is on the product's path."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

BS_CAP, BS_LISTCAP, BS_OVERSAMPLE, BS_MAX_LEAVES = 4096, 4096, 16, 43000


def shuffled_s_keys():
    """(junction index << 16) | read offset of every record of S: the order of the device's squeezed keys, ties included"""
    from intronprospector_amd import synth
    from tests.oracle_binding import Oracle
    b, names, _ = synth.short_reads(10_000_000, seed=20251024)
    o = Oracle(targets=names)
    o.process(b)
    tab = o.table()
    o.close()
    n = tab["n_offsets"].astype(np.int64)
    j = np.repeat(np.arange(len(n), dtype=np.uint64), n)
    return (j << np.uint64(16)) | tab["offsets"].astype(np.uint64)


def splitters(keys, per_range, leaf_avg=BS_CAP // 3, heavy_d=10, first_per_range=16):
    """ipx_launch_bsort's plan and the three sampling kernels; per_range: second-level sample keys aimed at per coarse range;
    first_per_range: first-level sample keys per coarse range (16 in the library: k_bs_sample1 sorts 16 * k1 keys in one block)"""
    n = len(keys)
    k_target = min(max(1, -(-n // leaf_avg)), BS_MAX_LEAVES)
    m2 = min(n, k_target * BS_OVERSAMPLE)
    k1 = min(max(max(-(-m2 // per_range), -(-k_target // 170)), 1), 256)
    m = min(n, min(max(first_per_range * k1, 256), BS_CAP))
    s1 = np.sort(keys[(np.arange(m, dtype=np.int64) * n) // m])
    coarse = s1[(np.arange(1, k1, dtype=np.int64) * m) // k1]
    s2 = keys[(np.arange(m2, dtype=np.int64) * n + n // 2) // m2]
    c = np.searchsorted(coarse, s2, side="right")
    order = np.argsort(c, kind="stable")
    cs, s2s = c[order], s2[order]
    starts = np.searchsorted(cs, np.arange(k1 + 1))
    bounds, dropped, most, longest = [], 0, 0, int(np.diff(starts).max())
    for r in range(k1):
        lst = np.sort(s2s[starts[r]:starts[r + 1]])[:BS_LISTCAP]
        mm = len(lst)
        f = min(max((mm + BS_OVERSAMPLE // 2) // BS_OVERSAMPLE, 1), 256)
        out = []
        for t in range(f - 1):
            rr = ((t + 1) * mm) // f
            v = lst[rr]
            if t >= 1 and v == lst[(t * mm) // f]:
                continue
            heavy = (rr + heavy_d < mm and lst[rr + heavy_d] == v) or (rr >= heavy_d and lst[rr - heavy_d] == v)
            out.append(v)
            if heavy:
                out.append(v + np.uint64(1))
        most = max(most, len(out))
        if len(out) > 255:
            dropped += len(out) - 255
            out = out[:255]
        if r > 0:
            bounds.append(coarse[r - 1:r])
        bounds.append(np.array(out, dtype=np.uint64))
    return np.concatenate(bounds), k1, dropped, most, longest


def worst_mixed_leaf(keys, bnd):
    """size of the largest leaf whose keys are not all equal (a leaf of equal keys may have any size)"""
    ids = np.searchsorted(bnd, keys, side="right")
    cnt = np.bincount(ids, minlength=len(bnd) + 1)
    worst = 0
    for b in np.argsort(cnt)[::-1][:600]:
        if cnt[b] <= worst:
            break
        kk = keys[ids == b]
        if kk.min() != kk.max():
            worst = int(cnt[b])
            break
    return worst, len(cnt)


if __name__ == "__main__":
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    keys = shuffled_s_keys()
    rng = np.random.default_rng(2)
    for per_range in (1800, 600):
        for t in range(trials):
            k = keys[rng.permutation(len(keys))]
            bnd, k1, dropped, most, longest = splitters(k, per_range)
            worst, leaves = worst_mixed_leaf(k, bnd)
            print("keys a range %4d: %3d ranges, %5d leaves, longest second-level list %4d, most splitters in a range %3d, dropped %2d, "
                  "largest mixed leaf %5d%s" % (per_range, k1, leaves, longest, most, dropped, worst,
                                                "  <- does not fit LDS" if worst > BS_CAP else ""), flush=True)
    # k_bs_sample1 with half the first-level keys: how uneven do the coarse ranges get (the list holds BS_LISTCAP keys)
    for first in (16, 8):
        longest = [splitters(keys[rng.permutation(len(keys))], 600, first_per_range=first)[4] for _ in range(trials)]
        print("first-level keys a range %2d: longest second-level list %s of %d" % (first, longest, BS_LISTCAP), flush=True)
