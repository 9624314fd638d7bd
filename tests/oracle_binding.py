"""ctypes binding of the test-only CPU oracle (oracle/libipxo.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use this.
"""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(ROOT, "oracle", "libipxo.so")
REF_SPLICE_PATH = os.path.join(ROOT, "oracle", "_ref", "libref_splice.so")


class Params(C.Structure):
    _fields_ = [
        ("min_anchor_length", C.c_uint32),
        ("max_anchor_indel_size", C.c_uint32),
        ("min_intron_length", C.c_uint32),
        ("max_intron_length", C.c_uint32),
        ("junction_filter", C.c_int32),
        ("strandness", C.c_int32),
        ("excludes", C.c_uint32),
        ("skip_missing_targets", C.c_int32),
    ]


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    lib = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    lib.ipxo_default_params.argtypes = [C.POINTER(Params)]
    lib.ipxo_create.restype = vp
    lib.ipxo_create.argtypes = [C.POINTER(Params)]
    lib.ipxo_destroy.argtypes = [vp]
    lib.ipxo_add_targets.argtypes = [vp, C.c_int, C.POINTER(C.c_char_p)]
    lib.ipxo_set_genome.argtypes = [vp, vp]
    lib.ipxo_set_warn_stream.argtypes = [vp, vp]
    lib.ipxo_errmsg.restype = C.c_char_p
    lib.ipxo_errmsg.argtypes = [vp]
    lib.ipxo_genome_load.restype = vp
    lib.ipxo_genome_load.argtypes = [C.c_char_p]
    lib.ipxo_genome_free.argtypes = [vp]
    lib.ipxo_process_batch.restype = C.c_int
    lib.ipxo_process_batch.argtypes = [vp, C.c_uint64] + [vp] * 8 + [C.POINTER(C.c_uint64)]
    lib.ipxo_clear.argtypes = [vp]
    lib.ipxo_num_junctions.restype = C.c_size_t
    lib.ipxo_num_junctions.argtypes = [vp]
    lib.ipxo_export.restype = C.c_size_t
    lib.ipxo_export.argtypes = [vp] * 12
    lib.ipxo_export_offsets.restype = C.c_uint64
    lib.ipxo_export_offsets.argtypes = [vp, vp, C.c_uint64]
    lib.ipxo_is_canonical.argtypes = [C.c_char_p]
    _lib = lib
    return lib


def load_ref_splice():
    """The reference's own splice_juncs.cc (oracle/_ref), or None if not built."""
    if not os.path.exists(REF_SPLICE_PATH):
        return None
    lib = C.CDLL(REF_SPLICE_PATH)
    lib.ref_is_canonical.argtypes = [C.c_char_p]
    lib.ref_junction_filter_check.argtypes = [C.c_int, C.c_char_p]
    lib.ref_junction_filter_parse.argtypes = [C.c_char_p]
    return lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleError(RuntimeError):
    def __init__(self, code, index, msg):
        super().__init__(msg)
        self.code, self.index, self.msg = code, index, msg


class Oracle:
    """One reference-style extractor instance over SoA batches."""

    def __init__(self, min_anchor_length=8, max_anchor_indel_size=0, min_intron_length=70,
                 max_intron_length=500000, junction_filter=0, strandness=0, excludes=0,
                 skip_missing_targets=0, genome_fasta=None, targets=()):
        self.lib = load()
        p = Params(min_anchor_length, max_anchor_indel_size, min_intron_length, max_intron_length,
                   junction_filter, strandness, excludes, skip_missing_targets)
        self.h = self.lib.ipxo_create(C.byref(p))
        self.genome = None
        if genome_fasta:
            self.genome = self.lib.ipxo_genome_load(os.fsencode(genome_fasta))
            if not self.genome:
                raise RuntimeError("can't open genome FASTA file or associated index: %s" % genome_fasta)
            self.lib.ipxo_set_genome(self.h, self.genome)
        if len(targets):
            self.add_targets(targets)

    def add_targets(self, names):
        arr = (C.c_char_p * len(names))(*[n.encode() if isinstance(n, str) else n for n in names])
        self.lib.ipxo_add_targets(self.h, len(names), arr)

    def process(self, batch):
        """batch: dict of numpy arrays tid,pos,flag,xs,nh,cigar_off,n_cigar,cigar."""
        n = len(batch["tid"])
        err = C.c_uint64(0)
        arrs = [np.ascontiguousarray(batch[k], dtype=dt) for k, dt in (
            ("tid", np.int32), ("pos", np.int32), ("flag", np.uint16), ("xs", np.uint8),
            ("nh", np.uint8), ("cigar_off", np.uint32), ("n_cigar", np.uint32), ("cigar", np.uint32))]
        rc = self.lib.ipxo_process_batch(self.h, n, *[_p(a) for a in arrs], C.byref(err))
        if rc:
            raise OracleError(rc, err.value, self.lib.ipxo_errmsg(self.h).decode())

    def clear(self):
        self.lib.ipxo_clear(self.h)

    def table(self):
        """Junction table in std::map order plus the sorted offset multisets."""
        n = self.lib.ipxo_num_junctions(self.h)
        t = {
            "target": np.zeros(n, np.int32), "intron_start": np.zeros(n, np.uint32),
            "intron_end": np.zeros(n, np.uint32), "anchor_start": np.zeros(n, np.uint32),
            "anchor_end": np.zeros(n, np.uint32), "strand": np.zeros(n, np.uint8),
            "ijunc": np.zeros(n, np.uint32), "counts": np.zeros((n, 3), np.uint64),
            "confidence": np.zeros(n, np.float32), "splice_sites": np.zeros((n, 6), np.uint8),
            "n_offsets": np.zeros(n, np.uint64),
        }
        self.lib.ipxo_export(self.h, _p(t["target"]), _p(t["intron_start"]), _p(t["intron_end"]),
                             _p(t["anchor_start"]), _p(t["anchor_end"]), _p(t["strand"]),
                             _p(t["ijunc"]), _p(t["counts"]), _p(t["confidence"]),
                             _p(t["splice_sites"]), _p(t["n_offsets"]))
        tot = int(t["n_offsets"].sum())
        offs = np.zeros(tot, np.uint16)
        got = self.lib.ipxo_export_offsets(self.h, _p(offs), tot)
        assert got == tot
        t["offsets"] = offs
        return t

    def close(self):
        if self.h:
            self.lib.ipxo_destroy(self.h)
            self.h = None
        if self.genome:
            self.lib.ipxo_genome_free(self.genome)
            self.genome = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
