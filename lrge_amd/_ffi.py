"""ctypes binding of include/lrge_hip.h.  There is no CPU fallback: if liblrge_hip.so is missing
or no HIP device is usable, every compute entry point raises."""
import ctypes as C
import os

import numpy as np

from .build import LIB_PATH

OK = 0
ERR_IO, ERR_PARSE, ERR_TOO_MANY, ERR_TOO_FEW, ERR_DEVICE, ERR_MAP, ERR_DUPLICATE_ID, ERR_PAF_WRITE, ERR_INVALID = \
    -1, -2, -3, -4, -5, -6, -7, -8, -9
PRESET_AVA_ONT, PRESET_AVA_PB = 0, 1

T_NAMES = ["pack", "sketch", "index_sort", "index_table", "qfilter", "lookup", "expand", "anchor_sort", "group",
           "chain", "chain_glb", "count", "total", "chain_lpg", "rs_scatter", "k_lookup", "index_restrict", "k_sketch"]
C_NAMES = ["query_bases", "query_minimizers", "anchors", "groups", "groups_chained", "chain_launches", "batches",
           "chain_anchors", "chain_glb_launches", "chain_glb_anchors", "lpg_launches", "lpg_anchors",
           "rs_scatter_launches", "rs_scatter_items", "rs_scatter_bytes", "lpg_split", "lookup_launches", "table_disp_sum", "anchors_kept", "index_parts", "sketch_launches", "sketch_wave_launches"]

EXPORTS = [
    "lrge_hip_device_count", "lrge_hip_ctx_create", "lrge_hip_ctx_destroy", "lrge_hip_last_error", "lrge_hip_ctx_set_option",
    "lrge_hip_seqset_upload", "lrge_hip_seqset_upload_async", "lrge_hip_seqset_wait", "lrge_hip_host_alloc",
    "lrge_hip_host_free", "lrge_hip_seqset_free", "lrge_hip_seqset_size", "lrge_hip_seqset_presketch", "lrge_hip_seqset_presketch_sharded", "lrge_hip_pack_choice", "lrge_hip_read_records",
    "lrge_hip_index_build", "lrge_hip_index_build_for", "lrge_hip_index_build_sharded", "lrge_hip_index_build_tsharded", "lrge_hip_last_shard_stats", "lrge_hip_index_free",
    "lrge_hip_comm_alltoallv", "lrge_hip_comm_rccl_ranks", "lrge_hip_comm_rccl_ops", "lrge_hip_comm_local_group_serialize", "lrge_hip_comm_local_turn",
    "lrge_hip_comm_busy_ms", "lrge_hip_comm_standin_ms",
    "lrge_hip_comm_unique_id", "lrge_hip_comm_create", "lrge_hip_comm_local_group_create", "lrge_hip_comm_local_group_destroy",
    "lrge_hip_comm_create_local", "lrge_hip_comm_create_host", "lrge_hip_comm_destroy", "lrge_hip_comm_abort", "lrge_hip_comm_rank", "lrge_hip_comm_world",
    "lrge_hip_comm_allreduce_u32", "lrge_hip_comm_allgather", "lrge_hip_index_stats",
    "lrge_hip_overlap_twoset", "lrge_hip_overlap_inverse", "lrge_hip_overlap_ava", "lrge_hip_chains",
    "lrge_hip_estimates", "lrge_hip_median", "lrge_hip_paf_stats",
    "lrge_hip_unique_random_set", "lrge_hip_chacha_block",
    "lrge_hip_sketch_dump", "lrge_hip_index_dump", "lrge_hip_anchors_dump",
    "lrge_hip_set_timer_level", "lrge_hip_last_timings", "lrge_hip_last_counters", "lrge_hip_version",
]


class Params(C.Structure):
    _fields_ = [("remove_internal", C.c_int32), ("max_overhang_ratio", C.c_float)]


CHAIN = np.dtype([("query", "<u4"), ("target", "<u4"), ("rev", "<i4"), ("score", "<i4"), ("cnt", "<i4"),
                  ("qs", "<i4"), ("qe", "<i4"), ("rs", "<i4"), ("re", "<i4"), ("mlen", "<i4"), ("blen", "<i4"),
                  ("n_seeds", "<i4")])

_lib = None


class LrgeHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("lrge_hip error %d: %s" % (code, msg))
        self.code = code


def lib():
    """Load liblrge_hip.so (built in-tree by lrge_amd.build / __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("LRGE_HIP_LIB_AB") or LIB_PATH     # LRGE_HIP_LIB_AB: another build of the same library (tools/ab.sh)
    if not os.path.exists(path):
        raise RuntimeError("liblrge_hip.so not built (%s): run `python -c 'import __graft_entry__ as g; g.build()'`; "
                           "there is no CPU fallback" % path)
    L = C.CDLL(path)
    vp = C.c_void_p
    L.lrge_hip_version.restype = C.c_char_p
    L.lrge_hip_last_error.restype = C.c_char_p
    L.lrge_hip_last_error.argtypes = [vp]
    L.lrge_hip_device_count.argtypes = [C.POINTER(C.c_int)]
    L.lrge_hip_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.lrge_hip_ctx_destroy.argtypes = [vp]
    L.lrge_hip_ctx_destroy.restype = None
    L.lrge_hip_ctx_set_option.argtypes = [vp, C.c_char_p, C.c_char_p]
    L.lrge_hip_seqset_upload.argtypes = [vp, vp, vp, C.c_uint32, vp, C.POINTER(vp)]
    L.lrge_hip_seqset_upload_async.argtypes = [vp, vp, vp, C.c_uint32, vp, C.POINTER(vp)]
    L.lrge_hip_seqset_wait.argtypes = [vp]
    L.lrge_hip_host_alloc.argtypes = [C.c_size_t, C.POINTER(vp)]
    L.lrge_hip_host_free.argtypes = [vp]
    L.lrge_hip_host_free.restype = None
    L.lrge_hip_seqset_free.argtypes = [vp]
    L.lrge_hip_seqset_free.restype = None
    L.lrge_hip_seqset_size.argtypes = [vp]
    L.lrge_hip_seqset_size.restype = C.c_uint32
    L.lrge_hip_seqset_presketch.argtypes = [vp, vp, C.c_int]
    L.lrge_hip_pack_choice.argtypes = [C.c_int, C.POINTER(C.c_double)]
    L.lrge_hip_seqset_presketch_sharded.argtypes = [vp, vp, C.c_int, vp]
    L.lrge_hip_index_build.argtypes = [vp, vp, C.c_int, C.POINTER(vp)]
    L.lrge_hip_index_build_for.argtypes = [vp, vp, C.c_int, vp, vp, C.POINTER(vp)]
    L.lrge_hip_index_build_sharded.argtypes = [vp, vp, vp, C.c_uint32, vp, C.c_uint32, C.c_int, vp, vp, C.POINTER(vp)]
    L.lrge_hip_index_build_tsharded.argtypes = [vp, vp, C.c_int, vp, C.POINTER(vp)]
    L.lrge_hip_last_shard_stats.argtypes = [vp, C.POINTER(C.c_uint64 * 8)]
    L.lrge_hip_comm_alltoallv.argtypes = [vp, vp, vp, vp, vp, C.c_size_t]
    L.lrge_hip_comm_rccl_ranks.argtypes = [vp, C.POINTER(C.c_int)]
    L.lrge_hip_comm_rccl_ops.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.lrge_hip_comm_local_group_serialize.argtypes = [vp, C.c_int]
    L.lrge_hip_comm_local_turn.argtypes = [vp, C.c_int]
    L.lrge_hip_comm_busy_ms.argtypes = [vp, C.c_int]
    L.lrge_hip_comm_busy_ms.restype = C.c_double
    L.lrge_hip_comm_standin_ms.argtypes = [vp, C.c_int]
    L.lrge_hip_comm_standin_ms.restype = C.c_double
    L.lrge_hip_comm_unique_id.argtypes = [vp]
    L.lrge_hip_comm_create.argtypes = [vp, C.c_int, C.c_int, vp, C.POINTER(vp)]
    L.lrge_hip_comm_local_group_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.lrge_hip_comm_local_group_destroy.argtypes = [vp]
    L.lrge_hip_comm_local_group_destroy.restype = None
    L.lrge_hip_comm_create_local.argtypes = [vp, C.c_int, vp, C.POINTER(vp)]
    L.lrge_hip_comm_create_host.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, C.POINTER(vp)]
    L.lrge_hip_comm_destroy.argtypes = [vp]
    L.lrge_hip_comm_destroy.restype = None
    L.lrge_hip_comm_rank.argtypes = [vp]
    L.lrge_hip_comm_world.argtypes = [vp]
    L.lrge_hip_comm_allreduce_u32.argtypes = [vp, vp, C.c_size_t]
    L.lrge_hip_comm_allgather.argtypes = [vp, vp, C.c_size_t, vp]
    L.lrge_hip_index_free.argtypes = [vp]
    L.lrge_hip_index_free.restype = None
    L.lrge_hip_index_stats.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int32)]
    L.lrge_hip_overlap_twoset.argtypes = [vp, vp, vp, C.POINTER(Params), vp, vp]
    L.lrge_hip_overlap_inverse.argtypes = [vp, vp, vp, C.POINTER(Params), vp]
    L.lrge_hip_overlap_ava.argtypes = [vp, vp, vp, C.POINTER(Params), vp]
    L.lrge_hip_chains.argtypes = [vp, vp, vp, C.c_int, vp, C.c_uint64, C.POINTER(C.c_uint64)]
    L.lrge_hip_paf_stats.argtypes = [vp, vp, vp, vp, vp, vp]
    L.lrge_hip_estimates.argtypes = [vp, vp, vp, C.c_uint32, C.c_float, C.c_uint64, C.c_uint32, vp]
    L.lrge_hip_median.argtypes = [vp, C.c_uint64, C.c_int, C.c_int, C.c_float, C.c_int, C.c_float,
                                  C.POINTER(C.c_float * 3), C.POINTER(C.c_int * 3)]
    L.lrge_hip_unique_random_set.argtypes = [C.c_uint64, C.c_uint32, C.c_int, C.c_uint64, vp]
    L.lrge_hip_chacha_block.argtypes = [vp, C.c_uint64, C.c_int, vp]
    L.lrge_hip_sketch_dump.argtypes = [vp, vp, C.c_int, vp, vp, C.c_uint64, C.POINTER(C.c_uint64)]
    L.lrge_hip_index_dump.argtypes = [vp, vp, vp, vp, C.c_uint64, C.POINTER(C.c_uint64)]
    L.lrge_hip_anchors_dump.argtypes = [vp, vp, vp, C.c_int, C.c_uint32, vp, vp, C.c_uint64, C.POINTER(C.c_uint64)]
    L.lrge_hip_set_timer_level.argtypes = [vp, C.c_int]
    L.lrge_hip_last_timings.argtypes = [vp, C.POINTER(C.c_float * len(T_NAMES))]
    L.lrge_hip_last_counters.argtypes = [vp, C.POINTER(C.c_uint64 * len(C_NAMES))]
    _lib = L
    return L
