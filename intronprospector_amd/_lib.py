"""ctypes view of include/ipx.h (libipx.so, built in-tree by csrc/Makefile).

There is no Python or CPU fallback: if the shared library is missing, or no
gfx950 device can be opened, the caller gets an exception.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IPX_LIB_PATH") or os.path.join(HERE, "libipx.so")     # IPX_LIB_PATH: an instrumented build (profiles/)

IPX_OK, IPX_EINVAL, IPX_ENODEVICE, IPX_ENOMEM, IPX_EDEVICE, IPX_EDATA = range(6)
IPX_MEM_HOST, IPX_MEM_DEVICE = 0, 1
IPX_SJ_NULL, IPX_SJ_ALL, IPX_SJ_CANON = 0, 1, 2
IPX_UNSTRANDED, IPX_RF_STRANDED, IPX_FR_STRANDED = 0, 1, 2
IPX_SCOPE_TARGET, IPX_SCOPE_GLOBAL = 0, 1
IPX_EXCLUDE_MULTI = 1
IPX_DATA_CIGAR, IPX_DATA_CATEGORY, IPX_DATA_GENOME_MISSING, IPX_DATA_GENOME_SHORT, IPX_DATA_TARGET = 1, 2, 3, 4, 5
MOTIF_PRESENT, MOTIF_NONACGT, MOTIF_CANON_P, MOTIF_CANON_RC = 0x8000, 0x4000, 0x2000, 0x1000

# every symbol include/ipx.h declares
SYMBOLS = [
    "ipx_abi_version", "ipx_create", "ipx_destroy", "ipx_set_targets", "ipx_set_genome_2bit",
    "ipx_reserve", "ipx_push_batch", "ipx_finalize", "ipx_clear", "ipx_last_error",
    "ipx_finalize_shard", "ipx_copy", "ipx_records_pending", "ipx_records_sample",
    "ipx_records_partition", "ipx_records_adopt", "ipx_records_export", "ipx_last_timing", "ipx_profile",
    "ipx_profile_only", "ipx_profile_read", "ipx_set_option",
    "ipx_group_create", "ipx_group_destroy", "ipx_group_set_targets", "ipx_group_set_genome_2bit",
    "ipx_group_set_option", "ipx_group_push_batch", "ipx_group_push_member", "ipx_group_finalize", "ipx_group_clear",
    "ipx_group_last_error", "ipx_group_uses_rccl",
    "ipx_push_batch_async", "ipx_sync", "ipx_host_alloc", "ipx_host_free",
    "ipx_push_bam", "ipx_push_bam_async", "ipx_bgzf_inflate",
    "ipx_records_partition_slots", "ipx_records_adopt_slots", "ipx_records_partition_slots_spill", "ipx_records_adopt_more", "ipx_use_stream", "ipx_copy_async", "ipx_decode_bgzf",
    "ipx_table_sorted_orders", "ipx_table_global_ijunc",
]
IPX_OPT_COMBINE = 1


class Params(C.Structure):
    _fields_ = [
        ("min_anchor_length", C.c_uint32), ("max_anchor_indel_size", C.c_uint32),
        ("min_intron_length", C.c_uint32), ("max_intron_length", C.c_uint32),
        ("sj_filter", C.c_int32), ("strandness", C.c_int32), ("excludes", C.c_uint32),
        ("scope", C.c_int32), ("skip_missing_targets", C.c_int32), ("device", C.c_int32),
    ]


class Batch(C.Structure):
    _fields_ = [
        ("n_alignments", C.c_uint64), ("n_cigar_words", C.c_uint64), ("base_read_index", C.c_uint64),
        ("tid", C.c_void_p), ("pos", C.c_void_p), ("flag", C.c_void_p), ("xs", C.c_void_p),
        ("nh", C.c_void_p), ("cigar_off", C.c_void_p), ("n_cigar", C.c_void_p), ("cigar", C.c_void_p),
        ("memory", C.c_int32),
    ]


class BamBatch(C.Structure):
    _fields_ = [
        ("n_alignments", C.c_uint64), ("n_cigar_words", C.c_uint64), ("base_read_index", C.c_uint64),
        ("bytes", C.c_void_p), ("n_bytes", C.c_uint64), ("span_begin", C.c_void_p), ("span_first", C.c_void_p),
        ("n_spans", C.c_uint32), ("tid_map", C.c_void_p), ("n_tid_map", C.c_uint32), ("memory", C.c_int32),
    ]


class Table(C.Structure):
    _fields_ = [
        ("n_junctions", C.c_uint64),
        ("target", C.c_void_p), ("intron_start", C.c_void_p), ("intron_end", C.c_void_p),
        ("anchor_start", C.c_void_p), ("anchor_end", C.c_void_p), ("strand", C.c_void_p),
        ("motif", C.c_void_p), ("read_counts", C.c_void_p), ("creator_order", C.c_void_p),
        ("ijunc", C.c_void_p), ("hist_begin", C.c_void_p), ("hist_value", C.c_void_p),
        ("hist_count", C.c_void_p), ("n_hist", C.c_uint64),
        ("n_alignments", C.c_uint64), ("n_records", C.c_uint64), ("memory", C.c_int32),
    ]


class Error(C.Structure):
    _fields_ = [
        ("status", C.c_int32), ("kind", C.c_int32), ("read_index", C.c_uint64), ("target", C.c_uint32),
        ("start", C.c_uint32), ("end", C.c_uint32), ("retlen", C.c_int32), ("op", C.c_char),
        ("detail", C.c_char * 235), ("order", C.c_uint64),
    ]


class BgzfFile(C.Structure):
    _fields_ = [("bytes", C.c_void_p), ("n_bytes", C.c_uint64), ("n_blocks", C.c_uint64), ("payload_begin", C.c_void_p),
                ("payload_end", C.c_void_p), ("isize", C.c_void_p), ("crc32", C.c_void_p), ("records_begin", C.c_uint64),
                ("tid_map", C.c_void_p), ("n_tid_map", C.c_uint32)]


class SlotsInfo(C.Structure):
    _fields_ = [("n_records", C.c_uint64), ("max_part", C.c_uint64), ("overflow", C.c_uint32), ("error_slot", C.c_uint32),
                ("error_order", C.c_uint64), ("error_status", C.c_int32), ("reserved", C.c_int32),
                ("max_received", C.c_uint64), ("all_received", C.c_uint64)]


class KernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("launches", C.c_uint64), ("total_ms", C.c_double)]


_lib = None


def load():
    """Load libipx.so and declare the prototypes.  Loading does not touch the GPU."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("intronprospector_amd: %s is missing; build it with "
                          "`make -C intronprospector_amd/csrc` (hipcc, gfx950). There is no CPU fallback."
                          % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    lib.ipx_abi_version.restype = C.c_uint32
    lib.ipx_create.argtypes = [C.POINTER(Params), C.POINTER(vp)]
    lib.ipx_destroy.argtypes = [vp]
    lib.ipx_set_targets.argtypes = [vp, C.c_uint32, C.POINTER(C.c_char_p), C.POINTER(C.c_int64)]
    lib.ipx_set_genome_2bit.argtypes = [vp, C.c_uint32, C.POINTER(C.c_char_p), C.POINTER(C.c_int64),
                                        C.POINTER(vp), C.POINTER(vp)]
    lib.ipx_reserve.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint64]
    lib.ipx_set_option.argtypes = [vp, C.c_int32, C.c_int64]
    lib.ipx_push_batch.argtypes = [vp, C.POINTER(Batch)]
    lib.ipx_push_batch_async.argtypes = [vp, C.POINTER(Batch)]
    lib.ipx_sync.argtypes = [vp]
    lib.ipx_push_bam.argtypes = [vp, C.POINTER(BamBatch)]
    lib.ipx_push_bam_async.argtypes = [vp, C.POINTER(BamBatch)]
    lib.ipx_bgzf_inflate.argtypes = [C.c_int32, vp, C.c_uint64, vp, vp, vp, C.c_uint64, vp, C.c_uint64, vp, C.POINTER(C.c_float)]
    lib.ipx_host_alloc.argtypes = [C.POINTER(vp), C.c_size_t]
    lib.ipx_host_free.argtypes = [vp]
    lib.ipx_finalize.argtypes = [vp, C.c_int32, C.POINTER(Table)]
    lib.ipx_clear.argtypes = [vp]
    lib.ipx_last_error.argtypes = [vp, C.POINTER(Error)]
    lib.ipx_finalize_shard.argtypes = [vp, C.c_int32, C.POINTER(Table), C.POINTER(Table)]
    lib.ipx_copy.argtypes = [vp, vp, vp, C.c_uint64]
    lib.ipx_records_pending.argtypes = [vp, C.POINTER(C.c_uint64)]
    lib.ipx_records_sample.argtypes = [vp, C.c_uint32, C.POINTER(C.c_uint64)]
    lib.ipx_records_partition.argtypes = [vp, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), vp]
    lib.ipx_records_adopt.argtypes = [vp, C.c_uint64, vp, C.c_uint64]
    lib.ipx_records_partition_slots.argtypes = [vp, C.c_uint32, C.POINTER(C.c_uint64), C.c_uint64, C.c_uint64, C.c_int32, vp]
    lib.ipx_records_adopt_slots.argtypes = [vp, C.c_uint32, C.c_uint64, vp, C.c_uint64, C.POINTER(SlotsInfo)]
    lib.ipx_records_partition_slots_spill.argtypes = [vp, C.c_uint32, C.POINTER(C.c_uint64), C.c_uint64, C.c_uint64, C.c_int32, vp, vp, C.c_uint64]
    lib.ipx_records_adopt_more.argtypes = [vp, C.c_uint64, vp]
    lib.ipx_use_stream.argtypes = [vp, vp]
    lib.ipx_decode_bgzf.argtypes = [vp, C.POINTER(BgzfFile), C.POINTER(Batch)]
    lib.ipx_copy_async.argtypes = [vp, vp, vp, C.c_uint64]
    lib.ipx_table_sorted_orders.argtypes = [vp, C.POINTER(Table), vp, C.c_uint64]
    lib.ipx_table_global_ijunc.argtypes = [vp, C.POINTER(Table), vp, C.c_uint32, C.c_uint64, C.c_uint64, vp]
    lib.ipx_records_export.argtypes = [vp, vp, C.c_uint64, C.POINTER(C.c_uint64)]
    lib.ipx_last_timing.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.ipx_profile.argtypes = [vp, C.c_int]
    lib.ipx_profile_only.argtypes = [vp, C.c_char_p]
    lib.ipx_profile_read.argtypes = [vp, C.POINTER(KernelStat), C.c_uint32, C.POINTER(C.c_uint32)]
    lib.ipx_group_create.argtypes = [C.POINTER(Params), C.c_uint32, C.POINTER(C.c_int32), C.POINTER(vp)]
    lib.ipx_group_destroy.argtypes = [vp]
    lib.ipx_group_set_targets.argtypes = [vp, C.c_uint32, C.POINTER(C.c_char_p), C.POINTER(C.c_int64)]
    lib.ipx_group_set_genome_2bit.argtypes = [vp, C.c_uint32, C.POINTER(C.c_char_p), C.POINTER(C.c_int64),
                                              C.POINTER(vp), C.POINTER(vp)]
    lib.ipx_group_set_option.argtypes = [vp, C.c_int32, C.c_int64]
    lib.ipx_group_push_batch.argtypes = [vp, C.POINTER(Batch)]
    lib.ipx_group_push_member.argtypes = [vp, C.c_uint32, C.POINTER(Batch)]
    lib.ipx_group_finalize.argtypes = [vp, C.POINTER(Table)]
    lib.ipx_group_clear.argtypes = [vp]
    lib.ipx_group_last_error.argtypes = [vp, C.POINTER(Error)]
    lib.ipx_group_uses_rccl.argtypes = [vp]
    for name in SYMBOLS:
        getattr(lib, name)
    _lib = lib
    return lib
