"""ctypes binding of libtksmseq.so -- the C-ABI declared in include/tksmseq.h.

The shared library is the product; this module only loads it.  There is no fallback: if the
library is missing or was not built, importing the compute path raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, os.environ.get("TKSMSEQ_LIB", "libtksmseq.so"))   # TKSMSEQ_LIB: diagnostic builds

OK, EINVAL, EIO, EDEVICE, ENOMEM, ESTATE, ELIMIT = range(7)
MODE_PERFECT, MODE_BADREAD = 0, 1


class BatchDesc(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("n_intervals", C.c_uint64), ("n_mods", C.c_uint64),
                ("n_literals", C.c_uint64), ("literal_bytes", C.c_uint64), ("id_bytes", C.c_uint64),
                ("reads", C.c_void_p), ("intervals", C.c_void_p), ("mods", C.c_void_p), ("literals", C.c_void_p),
                ("literal_pool", C.c_void_p), ("ids", C.c_void_p), ("id_pool", C.c_void_p)]


class RunParams(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("first_read_index", C.c_uint64), ("read_index_stride", C.c_uint64),
                ("mode", C.c_int32), ("fastq", C.c_int32), ("compute_qual", C.c_int32), ("collect_stats", C.c_int32),
                ("perfect_of_badread", C.c_int32), ("reserved", C.c_int32)]


class Result(C.Structure):
    _fields_ = [("records", C.c_void_p), ("record_offsets", C.c_void_p), ("records_bytes", C.c_uint64),
                ("n_reads", C.c_uint64), ("bases_in", C.c_uint64), ("bases_out", C.c_uint64),
                ("kernel_ms", C.c_float * 8)]


class TailModelDesc(C.Structure):            # tksmseq_tail_model
    _fields_ = [("n_lx", C.c_uint32), ("n_ly", C.c_uint32), ("lx", C.c_void_p), ("ly", C.c_void_p), ("grid", C.c_void_p),
                ("trans", C.c_double * 16), ("ratio", C.c_double), ("bases", C.c_uint8 * 4), ("pad", C.c_uint8 * 4)]


# every symbol include/tksmseq.h declares (checked by tests/test_abi.py without a GPU)
class PcrParams(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("target_count", C.c_uint64), ("cycles", C.c_int32), ("flags", C.c_int32),
                ("error_rate", C.c_double), ("efficiency", C.c_double), ("template_begin", C.c_uint64), ("template_end", C.c_uint64)]


class TrcParams(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("first_molecule_index", C.c_uint64), ("mode", C.c_int32), ("always_end", C.c_int32),
                ("kde_models_length", C.c_int32), ("flags", C.c_int32), ("mu", C.c_double), ("sigma", C.c_double),
                ("kde_model_path", C.c_char_p)]


TRC_NORMAL, TRC_LOGNORMAL, TRC_KDE = 0, 1, 2

SYMBOLS = [
    "tksmseq_create", "tksmseq_destroy", "tksmseq_last_error", "tksmseq_version", "tksmseq_set_stream",
    "tksmseq_synchronize", "tksmseq_reference_add_fasta", "tksmseq_reference_add_contig",
    "tksmseq_reference_contig_id", "tksmseq_reference_info", "tksmseq_load_error_model",
    "tksmseq_load_qscore_model", "tksmseq_set_identity", "tksmseq_get_error_model", "tksmseq_get_qscore_model",
    "tksmseq_get_identity", "tksmseq_batch_create", "tksmseq_batch_from_mdf_text", "tksmseq_batch_info",
    "tksmseq_batch_free", "tksmseq_run", "tksmseq_set_output_buffer", "tksmseq_set_timing",
    "tksmseq_prefetch_model", "tksmseq_prefetch_identity", "tksmseq_result_download", "tksmseq_result_download_range", "tksmseq_result_copy_device", "tksmseq_stats_download", "tksmseq_interleave_records", "tksmseq_sequence_main",
    "tksmseq_clone", "tksmseq_host_alloc", "tksmseq_host_free", "tksmseq_load_tail_model", "tksmseq_set_tail_model",
    "tksmseq_set_host_threads", "tksmseq_model_available",
    "tksmseq_device_alloc", "tksmseq_device_free", "tksmseq_copy_to_host", "tksmseq_pcr_preset", "tksmseq_pcr", "tksmseq_pcr_template_counts", "tksmseq_truncate", "tksmseq_batch_to_mdf_text", "tksmseq_text_free",
    "tksmseq_molecules_from_mdf_text", "tksmseq_pcr_main", "tksmseq_truncate_main", "tksmseq_run_diagnostics",
]

_lib = None


def load():
    """Loads libtksmseq.so (raises if it has not been built: there is no CPU fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(make -C tksm_amd/csrc).  tksm_amd has no CPU fallback.")
    # (Several contexts in flight want a hardware queue each -- GPU_MAX_HW_QUEUES=16, read when the HIP runtime starts: the CLI and
    # bench.py set it for their own processes; a library import does not touch the environment of the application that embeds it --
    # INTEGRATION.md)
    lib = C.CDLL(LIB_PATH)
    vp, u64, i32 = C.c_void_p, C.c_uint64, C.c_int32
    P = C.POINTER
    sig = {
        "tksmseq_create": (C.c_int, [C.c_int, P(vp)]),
        "tksmseq_destroy": (None, [vp]),
        "tksmseq_last_error": (C.c_char_p, [vp]),
        "tksmseq_version": (C.c_char_p, []),
        "tksmseq_set_stream": (C.c_int, [vp, vp]),
        "tksmseq_clone": (C.c_int, [vp, C.POINTER(vp)]),
        "tksmseq_host_alloc": (C.c_int, [C.c_uint64, C.POINTER(vp)]),
        "tksmseq_host_free": (None, [vp]),
        "tksmseq_synchronize": (C.c_int, [vp]),
        "tksmseq_reference_add_fasta": (C.c_int, [vp, C.c_char_p]),
        "tksmseq_reference_add_contig": (C.c_int, [vp, C.c_char_p, vp, u64, C.c_int]),
        "tksmseq_reference_contig_id": (C.c_int, [vp, C.c_char_p]),
        "tksmseq_reference_info": (C.c_int, [vp, P(u64), P(u64), P(u64)]),
        "tksmseq_load_error_model": (C.c_int, [vp, C.c_char_p]),
        "tksmseq_load_qscore_model": (C.c_int, [vp, C.c_char_p]),
        "tksmseq_load_tail_model": (C.c_int, [vp, C.c_char_p]),
        "tksmseq_set_tail_model": (C.c_int, [vp, vp]),
        "tksmseq_set_host_threads": (C.c_int, [vp, C.c_int]),
        "tksmseq_pcr_preset": (C.c_int, [C.c_char_p, P(C.c_double), P(C.c_double)]),
        "tksmseq_pcr": (C.c_int, [vp, vp, vp, P(vp)]),
        "tksmseq_pcr_template_counts": (C.c_int, [vp, vp, vp, P(C.c_uint64)]),
        "tksmseq_truncate": (C.c_int, [vp, vp, vp, P(vp)]),
        "tksmseq_batch_to_mdf_text": (C.c_int, [vp, vp, P(vp), P(u64)]),
        "tksmseq_text_free": (None, [vp]),
        "tksmseq_molecules_from_mdf_text": (C.c_int, [vp, C.c_char_p, u64, P(vp)]),
        "tksmseq_model_available": (C.c_int, [C.c_char_p, C.c_char_p]),
        "tksmseq_set_identity": (C.c_int, [vp, C.c_double, C.c_double, C.c_double]),
        "tksmseq_get_error_model": (C.c_int, [vp, P(i32), P(i32), P(i32), vp, vp, vp]),
        "tksmseq_get_qscore_model": (C.c_int, [vp, P(i32), P(i32), P(u64), vp, vp, vp, vp, vp]),
        "tksmseq_get_identity": (C.c_int, [vp, P(i32), P(C.c_double), P(C.c_double), P(C.c_double), vp]),
        "tksmseq_batch_create": (C.c_int, [vp, P(BatchDesc), P(vp)]),
        "tksmseq_batch_from_mdf_text": (C.c_int, [vp, C.c_char_p, u64, P(vp)]),
        "tksmseq_batch_info": (C.c_int, [vp, P(u64), P(u64), P(u64)]),
        "tksmseq_batch_free": (None, [vp, vp]),
        "tksmseq_run": (C.c_int, [vp, vp, P(RunParams), P(Result)]),
        "tksmseq_set_output_buffer": (C.c_int, [vp, vp, u64]),
        "tksmseq_set_timing": (C.c_int, [vp, C.c_int]),
        "tksmseq_run_diagnostics": (C.c_int, [vp, P(C.c_uint32)]),
        "tksmseq_result_download": (C.c_int, [vp, vp, vp]),
        "tksmseq_prefetch_model": (C.c_int, [C.c_char_p, C.c_char_p]),
        "tksmseq_prefetch_identity": (C.c_int, [C.c_double, C.c_double, C.c_double]),
        "tksmseq_result_download_range": (C.c_int, [vp, vp, C.c_uint64, C.c_uint64, C.c_int]),
        "tksmseq_result_copy_device": (C.c_int, [vp, vp, vp]),
        "tksmseq_stats_download": (C.c_int, [vp, vp, vp]),
        "tksmseq_interleave_records": (C.c_int, [vp, C.c_int, P(vp), P(vp), P(u64), vp, u64, P(u64)]),
        "tksmseq_sequence_main": (C.c_int, [C.c_int, P(C.c_char_p)]),
    }
    for name, (res, args) in sig.items():
        f = getattr(lib, name)
        f.restype, f.argtypes = res, args
    _lib = lib
    return lib
