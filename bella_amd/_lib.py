"""ctypes loader for libbella_hip.so (the C ABI of include/bella_hip.h).

There is no fallback: if the library is missing or no gfx950 device is present the product path raises."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIBPATH = os.path.join(HERE, "libbella_hip.so")

PAIR_DT = np.dtype([("rid", "<u4"), ("cid", "<u4"), ("count", "<u2"), ("seedH", "<u2"), ("seedV", "<u2"), ("flags", "<u2")])
EXT_DT = np.dtype([("nbins", "<u2"), ("support", "<u2"), ("binov", "<u2"), ("pad", "<u2")])
ALN_DT = np.dtype([("score", "<i4"), ("begH", "<i4"), ("endH", "<i4"), ("begV", "<i4"), ("endV", "<i4"), ("ov", "<u2"),
                   ("strand", "u1"), ("passed", "u1"), ("steps", "<u4"), ("flagged", "<u4")])
SEED_DT = np.dtype([("rid", "<u4"), ("cid", "<u4"), ("seedH", "<u2"), ("seedV", "<u2")])
assert PAIR_DT.itemsize == 16 and EXT_DT.itemsize == 8 and ALN_DT.itemsize == 32 and SEED_DT.itemsize == 12


class Params(C.Structure):
    _fields_ = [("kmer_size", C.c_uint16), ("bin_size", C.c_uint16), ("xdrop", C.c_uint16), ("skip_alignment", C.c_uint16),
                ("error_rate", C.c_double), ("delta_chernoff", C.c_double)]


class WriteStats(C.Structure):
    _fields_ = [("lines", C.c_uint64), ("bytes", C.c_uint64), ("aligned_pairs", C.c_uint64), ("aligned_bases", C.c_uint64),
                ("total_read_len", C.c_uint64), ("bases_passed", C.c_uint64), ("bases_failed", C.c_uint64), ("seconds", C.c_double),
                ("format_seconds", C.c_double), ("threads", C.c_uint32), ("pad", C.c_uint32)]


class IngestStats(C.Structure):
    _fields_ = [("file_bytes", C.c_uint64), ("bases", C.c_uint64), ("reads", C.c_uint32), ("threads", C.c_uint32),
                ("index_ms", C.c_double), ("upload_ms", C.c_double)]


class Timings(C.Structure):
    _fields_ = [("assemble_ms", C.c_float), ("symbolic_ms", C.c_float), ("spgemm_ms", C.c_float), ("fold_ms", C.c_float),
                ("compact_ms", C.c_float), ("xdrop_ms", C.c_float), ("overlap_total_ms", C.c_float), ("spgemm_launches", C.c_uint32),
                ("kcount_ms", C.c_float), ("retry_columns", C.c_uint32), ("overflow_pairs", C.c_uint32), ("layout_ms", C.c_float),
                ("rows_ms", C.c_float), ("lane_order", C.c_uint32), ("expand_ms", C.c_float), ("numeric_passes", C.c_uint32),
                ("symbolic_passes", C.c_uint32), ("pad", C.c_uint32), ("numeric_columns", C.c_uint64)]


class Memory(C.Structure):
    _fields_ = [("reads_bytes", C.c_uint64), ("matrix_bytes", C.c_uint64), ("layout_A_bytes", C.c_uint64), ("layout_B_bytes", C.c_uint64),
                ("rowlist_bytes", C.c_uint64), ("pass_bytes", C.c_uint64), ("other_bytes", C.c_uint64), ("owned_nnz", C.c_uint64),
                ("layout_shared", C.c_uint64), ("live_nnz", C.c_uint64)]


# every symbol include/bella_hip.h declares: (name, restype, argtypes)
vp = C.c_void_p
SIGNATURES = [
    ("bella_hip_abi_version", C.c_int, []),
    ("bella_hip_device_count", C.c_int, []),
    ("bella_hip_init", C.c_int, [C.c_int, C.POINTER(vp)]),
    ("bella_hip_destroy", None, [vp]),
    ("bella_hip_strerror", C.c_char_p, [C.c_int]),
    ("bella_hip_last_error", C.c_char_p, [vp]),
    ("bella_hip_set_reads", C.c_int, [vp, vp, vp, C.c_uint32]),
    ("bella_hip_load_fastq", C.c_int, [vp, C.c_char_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]),
    ("bella_hip_load_fastq_list", C.c_int, [vp, C.POINTER(C.c_char_p), C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]),
    ("bella_hip_get_ingest_stats", C.c_int, [vp, C.c_void_p]),
    ("bella_hip_get_read_names", C.c_int, [vp, vp, C.c_uint64, vp, C.POINTER(C.c_uint64)]),
    ("bella_hip_get_read_lengths", C.c_int, [vp, vp]),
    ("bella_hip_count_kmers", C.c_int, [vp, C.c_uint16, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64),
                                        C.POINTER(C.c_uint64)]),
    ("bella_hip_count_syncmers", C.c_int, [vp, C.c_uint16, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64),
                                           C.POINTER(C.c_uint64)]),
    ("bella_hip_count_minimizers", C.c_int, [vp, C.c_uint16, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32),
                                             C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    ("bella_hip_get_dictionary", C.c_int, [vp, vp, vp]),
    ("bella_hip_get_tuples", C.c_int, [vp, vp, vp, vp]),
    ("bella_hip_assemble_counted", C.c_int, [vp]),
    ("bella_hip_assemble_counted_panel", C.c_int, [vp, C.c_uint32, C.c_uint32]),
    ("bella_hip_assemble_tuples", C.c_int, [vp, C.c_uint16, C.c_uint32, C.c_uint64, vp, vp, vp]),
    ("bella_hip_set_B", C.c_int, [vp, C.c_uint16, C.c_uint32, vp, vp, vp]),
    ("bella_hip_assemble_panel", C.c_int, [vp, C.c_uint16, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, vp, vp, vp]),
    ("bella_hip_set_B_panel", C.c_int, [vp, C.c_uint16, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, vp]),
    ("bella_hip_comm_id", C.c_int, [vp]),
    ("bella_hip_comm_init", C.c_int, [vp, C.c_int, C.c_int, vp]),
    ("bella_hip_comm_destroy", C.c_int, [vp]),
    ("bella_hip_comm_available", C.c_int, []),
    ("bella_hip_comm_id_local", C.c_int, [vp]),
    ("bella_hip_comm_init_local", C.c_int, [vp, C.c_int, C.c_int, vp]),
    ("bella_hip_allgather_panels", C.c_int, [vp]),
    ("bella_hip_count_kmers_dist", C.c_int, [vp, C.c_uint16, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                             C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    ("bella_hip_panel_device_ptrs", C.c_int, [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(vp),
                                               C.POINTER(vp), C.POINTER(vp)]),
    ("bella_hip_set_B_device", C.c_int, [vp, C.c_uint16, C.c_uint32, vp, vp, vp, C.c_uint64]),
    ("bella_hip_get_B", C.c_int, [vp, C.POINTER(C.c_uint64), vp, vp, vp]),
    ("bella_hip_set_partition", C.c_int, [vp, C.c_uint32, C.c_uint32]),
    ("bella_hip_set_column_range", C.c_int, [vp, C.c_uint32, C.c_uint32]),
    ("bella_hip_overlap", C.c_int, [vp, C.POINTER(Params), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    ("bella_hip_count_pairs", C.c_int, [vp, C.POINTER(Params), vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    ("bella_hip_get_pairs", C.c_int, [vp, vp, vp, vp]),
    ("bella_hip_align_pairs", C.c_int, [vp, C.POINTER(Params), C.POINTER(C.c_uint64)]),
    ("bella_hip_get_alignments", C.c_int, [vp, vp]),
    ("bella_hip_xdrop_batch", C.c_int, [vp, vp, C.c_uint64, C.POINTER(Params), vp]),
    ("bella_hip_align_pairs_exact", C.c_int, [vp, C.POINTER(Params), C.POINTER(C.c_uint64)]),
    ("bella_hip_xdrop_batch_exact", C.c_int, [vp, vp, C.c_uint64, C.POINTER(Params), vp]),
    ("bella_hip_write_output", C.c_int, [C.c_char_p, C.POINTER(Params), C.c_int, C.c_uint32, vp, vp, vp, vp, C.c_uint64, C.c_int,
                                         C.POINTER(WriteStats)]),
    ("bella_hip_get_timings", C.c_int, [vp, C.POINTER(Timings)]),
    ("bella_hip_get_memory", C.c_int, [vp, C.POINTER(Memory)]),
    ("bella_hip_get_memory_sized", C.c_int, [vp, vp, C.c_uint64]),
    ("bella_hip_set_debug", C.c_int, [vp, C.c_uint32]),
    ("bella_hip_reserve", C.c_int, [vp, C.c_uint64, C.POINTER(C.c_double)]),
    ("bella_hip_trim", C.c_int, [vp]),
    ("bella_hip_set_tuning", C.c_int, [vp, C.c_uint32, vp, C.c_uint32]),
]

_lib = None


def load():
    """Loads the HIP library; raises (never falls back) if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIBPATH):
            raise RuntimeError("libbella_hip.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        lib = C.CDLL(LIBPATH)
        for name, res, args in SIGNATURES:
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib
