"""ctypes binding of libaha_hip.so (include/aha_hip.h).

The library is built in-tree (aha_amd/csrc/Makefile, or
__graft_entry__.build()); there is no pure-Python or CPU fallback: if the
shared object is missing, importing the product path fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# AHA_HIP_LIB: another build of the same library (the CPU suite points it at the sanitizer build)
LIB_PATH = os.environ.get("AHA_HIP_LIB") or os.path.join(_HERE, "libaha_hip.so")
SYNTH_PATH = os.path.join(_HERE, "libaha_synth.so")

AHA_ABI_VERSION = 8
AHA_OK = 0
AHA_E_INVALID = -1
AHA_E_EMPTY_KEY = -2
AHA_E_ZERO_BYTE = -3
AHA_E_DUP_KEY = -4
AHA_E_SEP_SIZE = -5
AHA_E_CAPACITY = -6
AHA_E_NO_DEVICE = -7
AHA_E_HIP = -8
AHA_E_TOO_LONG = -9
AHA_E_NOT_FOUND = -10
AHA_E_TOO_LARGE = -11
AHA_E_NOMEM = -12

AHA_OPT_HOST_ONLY = 1
AHA_OPT_FORCE_WIDE = 2
AHA_IMG_SLOTS, AHA_IMG_END_KEY, AHA_IMG_KEY_LN, AHA_IMG_KEY_CNT, AHA_IMG_KEY_KC = 0, 1, 2, 3, 4
AHA_IMG_STALE_ENDS = 5
AHA_IMG_UNIT_SLOTS, AHA_IMG_UNIT_ROOT, AHA_IMG_UNIT_END_KEY, AHA_IMG_UNIT_TABLES = 6, 7, 8, 9
AHA_IMG_UNIT_MARKS, AHA_IMG_UNIT_PAIRS, AHA_IMG_UNIT_PAIR_DISP = 10, 11, 12


class aha_options(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device", C.c_int32), ("flags", C.c_uint32),
                ("reserved", C.c_uint32)]


class aha_match_params(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("char_offsets", C.c_int32), ("sep_size", C.c_int32),
                ("sep_bits", C.c_uint8 * 32), ("longest", C.c_int32)]


class aha_ac_info_t(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("n_keys", C.c_uint32), ("n_states", C.c_uint64),
                ("n_slots", C.c_uint64), ("image_bytes", C.c_uint64), ("max_key_len", C.c_uint32),
                ("slot_bytes", C.c_uint32), ("lds_slots", C.c_uint32), ("device", C.c_int32),
                ("fail_s1_lo", C.c_uint32), ("fail_s2_lo", C.c_uint32), ("fail_hdr_lo", C.c_uint32),
                ("unit_header_beside", C.c_uint32), ("unit_enabled", C.c_uint32), ("unit_slots", C.c_uint32),
                ("unit_syms", C.c_uint32), ("unit_multi_permille", C.c_uint32), ("unit_big_lo", C.c_uint32),
                ("unit_big_block", C.c_uint32), ("unit_n_low", C.c_uint32), ("unit_n_big", C.c_uint32),
                ("unit_base_bits", C.c_uint32), ("unit_headers", C.c_uint32),
                ("filter_prefix_bytes", C.c_uint32), ("filter_words", C.c_uint32),
                ("skip_filter_words", C.c_uint32), ("skip_pairs", C.c_uint32),
                ("pair_hash_k1", C.c_uint32), ("pair_table_log2", C.c_uint32), ("pair_groups", C.c_uint32),
                ("pair_engine", C.c_uint32)]


class aha_timing(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("n_kernels", C.c_uint32), ("ms_total", C.c_float),
                ("ms_count", C.c_float), ("ms_scan", C.c_float), ("ms_write", C.c_float),
                ("ms_aux", C.c_float), ("n_chunks", C.c_uint64), ("n_hits", C.c_uint64), ("engine", C.c_uint32),
                ("chunk_bytes", C.c_uint32), ("repeats", C.c_uint32), ("reserved", C.c_uint32)]


class aha_stream_seg(C.Structure):
    _fields_ = [("word_offset", C.c_uint64), ("n_hits", C.c_uint64), ("out_offset", C.c_uint64)]


class aha_group_timing(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("n_devices", C.c_uint32), ("ms_match", C.c_float),
                ("ms_match_max_shard", C.c_float), ("ms_exchange", C.c_float), ("ms_download", C.c_float),
                ("n_hits", C.c_uint64), ("exchange", C.c_uint32), ("packed", C.c_uint32), ("wire_bytes", C.c_uint64)]


# every symbol include/aha_hip.h declares: name -> (restype, argtypes)
_vp, _i32, _u32, _u64 = C.c_void_p, C.c_int32, C.c_uint32, C.c_uint64
SIGNATURES = {
    "aha_strerror": (C.c_char_p, [_i32]),
    "aha_last_error": (C.c_char_p, [_vp]),
    "aha_abi_version": (_u32, []),
    "aha_device_count": (_i32, []),
    "aha_ac_compile": (_i32, [_vp, _vp, _u32, C.POINTER(aha_options), C.POINTER(_vp), C.POINTER(_u32)]),
    "aha_ac_free": (None, [_vp]),
    "aha_ac_info": (_i32, [_vp, C.POINTER(aha_ac_info_t)]),
    "aha_ac_key": (_i32, [_vp, _i32, _vp, _i32]),
    "aha_ac_id": (_i32, [_vp, _vp, _i32]),
    "aha_ac_match_bytes": (_i32, [_vp, _vp, _u64, C.POINTER(aha_match_params), _vp, _u64, C.POINTER(_u64)]),
    "aha_ac_match_batch": (_i32, [_vp, _vp, _vp, _u64, C.POINTER(aha_match_params), _vp, _u64, _vp,
                                  C.POINTER(_u64)]),
    "aha_ac_match_batch_device": (_i32, [_vp, _vp, _vp, _u64, _u64, C.POINTER(aha_match_params), _vp, _u64,
                                         _vp, C.POINTER(_u64), _vp]),
    "aha_ac_export": (C.c_int64, [_vp, _i32, _vp, _u64]),
    "aha_ac_hits_pack_device": (_i32, [_vp, _vp, _u64, _vp, _vp]),
    "aha_ac_hits_unpack_device": (_i32, [_vp, _vp, _u64, _i32, _vp, _vp]),
    "aha_ac_stream_format": (_i32, [_vp, _vp, _vp]),
    "aha_ac_hits_pack4_device": (_i32, [_vp, _vp, _u64, _vp, _u64, _vp, _vp]),
    "aha_ac_hits_unpack4_device": (_i32, [_vp, _vp, _u64, _i32, _vp, _vp]),
    "aha_ac_hits_unpack4_segs_device": (_i32, [_vp, _vp, _vp, _u32, _i32, _vp, _vp]),
    "aha_ac_save": (C.c_int64, [_vp, _vp, _u64]),
    "aha_ac_load": (_i32, [_vp, _u64, C.POINTER(aha_options), C.POINTER(_vp)]),
    "aha_ac_release_scratch": (_i32, [_vp]),
    "aha_ac_scratch_bytes": (C.c_int64, [_vp]),
    "aha_ac_match_batch_keep": (_i32, [_vp, _vp, _vp, _u64, C.POINTER(aha_match_params), _vp, _u64, _vp,
                                C.POINTER(_u64)]),
    "aha_ac_replicate": (_i32, [_vp, _i32, C.POINTER(_vp)]),
    "aha_ac_match_batch_device_stream": (_i32, [_vp, _vp, _vp, _u64, _u64, C.POINTER(aha_match_params), _vp, _u64, _vp,
                                                 C.POINTER(_u64), _vp, _u64, _vp, _vp]),
    "aha_group_compile": (_i32, [_vp, _vp, _u32, _vp, _i32, _u32, C.POINTER(_vp), C.POINTER(_u32)]),
    "aha_group_free": (None, [_vp]),
    "aha_group_size": (_i32, [_vp]),
    "aha_group_last_error": (C.c_char_p, [_vp]),
    "aha_group_partition": (_i32, [_vp, _u64, _i32, _vp]),
    "aha_group_match_batch": (_i32, [_vp, _vp, _vp, _u64, C.POINTER(aha_match_params), _vp, _u64, _vp,
                                     C.POINTER(_u64)]),
    "aha_group_last_timing": (_i32, [_vp, C.POINTER(aha_group_timing)]),
    "aha_group_download_shard": (_i32, [_vp, _i32, _vp, _u64, C.POINTER(_u64)]),
    "aha_group_corpus_upload": (_i32, [_vp, _vp, _vp, _u64, C.POINTER(_vp)]),
    "aha_group_corpus_free": (None, [_vp]),
    "aha_group_match_batch_device": (_i32, [_vp, _vp, C.POINTER(aha_match_params), _vp, C.POINTER(_u64)]),
    "aha_buffer_alloc": (_i32, [_i32, _u64, C.POINTER(_vp)]),
    "aha_buffer_free": (_i32, [_i32, _vp]),
    "aha_buffer_upload": (_i32, [_i32, _vp, _vp, _u64]),
    "aha_buffer_download": (_i32, [_i32, _vp, _vp, _u64]),
    "aha_corpus_upload": (_i32, [_i32, _vp, _vp, _u64, C.POINTER(_vp)]),
    "aha_corpus_free": (None, [_vp]),
    "aha_corpus_bytes": (_vp, [_vp]),
    "aha_corpus_doc_offsets": (_vp, [_vp]),
    "aha_corpus_n_docs": (_u64, [_vp]),
    "aha_corpus_n_bytes": (_u64, [_vp]),
    "aha_corpus_device": (_i32, [_vp]),
    "aha_ac_set_profiling": (_i32, [_vp, _i32]),
    "aha_ac_last_timing": (_i32, [_vp, C.POINTER(aha_timing)]),
}

_lib = None
_synth = None


def lib():
    """Loads libaha_hip.so; raises (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C aha_amd/csrc` (there is no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        if L.aha_abi_version() != AHA_ABI_VERSION:  # structs and entry points below are this version's
            raise ImportError(f"{LIB_PATH} has ABI {L.aha_abi_version()}, this binding is written for ABI {AHA_ABI_VERSION}")
        _lib = L
    return _lib


def synth():
    global _synth
    if _synth is None:
        L = C.CDLL(SYNTH_PATH)
        L.aha_synth_keys.restype = C.c_int64
        L.aha_synth_keys.argtypes = [C.c_int, _u64, _u32, _vp, _u64, _vp, C.POINTER(_u32)]
        L.aha_synth_corpus.restype = C.c_int64
        L.aha_synth_corpus.argtypes = [C.c_int, _u64, _vp, _vp, _u32, _u32, _u64, _u64, _vp, _vp, _u64]
        _synth = L
    return _synth
