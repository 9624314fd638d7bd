"""Throughput of ipx_bgzf_inflate (one lane per BGZF block) on the blocks of a make-bam file; usage:
python profiles/gpu_inflate_timing.py [N_RECORDS]   (default 10 M: the file of profiles/cli_timeline.sh)"""
import ctypes as C, json, os, struct, subprocess, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from intronprospector_amd import _lib as L

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
bam = "/tmp/ipx_gpu_inflate.bam"
subprocess.run([os.path.join(ROOT, "intronprospector_amd", "bin", "ipx_host_check"), "make-bam", str(n), bam], check=True, stdout=subprocess.DEVNULL)
data = np.fromfile(bam, np.uint8)
raw = data.tobytes()
begin, end, isize = [], [], []
p = 0
while p < len(raw):
    xlen = struct.unpack_from("<H", raw, p + 10)[0]
    bsize = struct.unpack_from("<H", raw, p + 16)[0] + 1
    begin.append(p + 12 + xlen); end.append(p + bsize - 8)
    isize.append(struct.unpack_from("<I", raw, p + bsize - 4)[0])
    p += bsize
nb = len(begin)
# streams back to back (the entry point takes one offsets array)
lens = np.array(end, np.int64) - np.array(begin, np.int64)
off = np.zeros(nb + 1, np.uint64); off[1:] = np.cumsum(lens)
packed = np.empty(int(off[-1]), np.uint8)
for b in range(nb):
    packed[int(off[b]):int(off[b + 1])] = data[begin[b]:end[b]]
isz = np.array(isize, np.uint32)
out_off = np.zeros(nb, np.uint64); out_off[1:] = np.cumsum(isz[:-1].astype(np.uint64))
total = int(isz.astype(np.uint64).sum())
out = np.zeros(total, np.uint8); status = np.zeros(nb, np.uint32); ms = C.c_float(0)
lib = L.load()
res = []
for lpw in (os.environ.get("LPW_SWEEP", "").split() or [None]):
  if lpw: os.environ["IPX_INFLATE_LPW"] = lpw
  r2 = []
  for rep in range(2 if lpw else 0):
    rc = lib.ipx_bgzf_inflate(0, packed.ctypes.data, len(packed), off.ctypes.data, isz.ctypes.data, out_off.ctypes.data, nb,
                              out.ctypes.data, total, status.ctypes.data, C.byref(ms))
    r2.append(round(ms.value, 1))
  if lpw: print("lanes per wave", lpw, "kernel ms", r2, flush=True)
for rep in range(3):
    t0 = time.perf_counter()
    rc = lib.ipx_bgzf_inflate(0, packed.ctypes.data, len(packed), off.ctypes.data, isz.ctypes.data, out_off.ctypes.data, nb,
                              out.ctypes.data, total, status.ctypes.data, C.byref(ms))
    res.append((ms.value, time.perf_counter() - t0))
assert rc == 0 and (status == 0).all()
# spot check against zlib, and zlib's own rate on one core
t0 = time.perf_counter(); k = min(nb, 2000)
for b in range(k):
    assert zlib.decompress(raw[begin[b]:end[b]], -15) == out[int(out_off[b]):int(out_off[b]) + int(isz[b])].tobytes()
tz = time.perf_counter() - t0
print(json.dumps({"records": n, "blocks": nb, "compressed_MB": round(len(packed) / 1e6, 1), "inflated_MB": round(total / 1e6, 1),
                  "kernel_ms": [round(r[0], 1) for r in res], "call_s_with_copies_and_allocations": [round(r[1], 3) for r in res],
                  "kernel_inflated_GB_per_s": round(total / 1e9 / (min(r[0] for r in res) / 1e3), 2),
                  "zlib_one_core_inflated_MB_per_s_incl_compare": round(float(isz[:k].sum()) / 1e6 / tz, 1)}))
os.remove(bam)
