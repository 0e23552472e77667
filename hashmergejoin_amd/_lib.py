"""ctypes binding of include/hmj.h.  Loads hashmergejoin_amd/libhmj_hip.so (built in-tree by
`make`); raises loudly if it is missing -- there is no fallback path."""
import ctypes as C
import os

HMJ_MATERIALIZE = 0x01
HMJ_ORDERED = 0x02
HMJ_FIRST_WINS = 0x04
HMJ_CHECKSUM = 0x08
HMJ_SUM_PROBE = 0x10
HMJ_E_RCCL, HMJ_E_PEER, HMJ_E_TIMEOUT = -6, -7, -8
# hmj_timing.path bits
HMJ_PATH_SLAB, HMJ_PATH_EXACT, HMJ_PATH_UNIQ_WRITE, HMJ_PATH_SPLIT, HMJ_PATH_WINDOW = 0x1, 0x2, 0x4, 0x8, 0x10
HMJ_PATH_ORDER_DEFERRED, HMJ_PATH_ORDER_BY_KEY, HMJ_PATH_PREPARED, HMJ_PATH_CHUNKED_BUILD = 0x20, 0x40, 0x80, 0x100
HMJ_PATH_HOT_KEY_HINT = 0x200
HMJ_PATH_SLAB_PROBE = 0x400
HMJ_PATH_SORTED_WRITE = 0x800
HMJ_PATH_SORTED_FK = 0x1000
HMJ_PATH_DENSE_BUILD = 0x2000
HMJ_PATH_GLOBAL_TABLE = 0x100000
HMJ_PATH_ORDER_BY_RANK_SORT = 0x200000
HMJ_PATH_ORDERED_EXPANSION = 0x400000
HMJ_PATH_LDS_TABLE = 0x800000
HMJ_PATH_RANK_RUNS = 0x1000000
HMJ_PATH_RANK_LOOKUP_IN_PASS = 0x2000000
HMJ_PATH_SORT_MSD = 0x4000000
HMJ_PATH_KEY_RANGES = 0x8000000
HMJ_PATH_SLAB_ONE_PASS = 0x80000
HMJ_PATH_HOST_PIPELINE = 0x4000
HMJ_PATH_SORTED_FK_HALF = 0x8000
HMJ_PATH_LOOKBACK_TIMEOUT = 0x10000
HMJ_PATH_SORTED_FK_WIDE = 0x20000
HMJ_PATH_PRESORTED = 0x40000

_HERE = os.path.dirname(os.path.abspath(__file__))
_U64P = C.POINTER(C.c_uint64)


class HmjError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("hmj error %d: %s" % (code, msg))
        self.code = code


class JoinResult(C.Structure):
    _fields_ = [("n_matches", C.c_uint64), ("sum_r", C.c_uint64), ("sum_s", C.c_uint64),
                ("xor_fold", C.c_uint64), ("mix_sum", C.c_uint64), ("sum_probe_all", C.c_uint64),
                ("key", C.c_void_p), ("rval", C.c_void_p), ("sval", C.c_void_p)]

    def checks(self):
        return {k: int(getattr(self, k)) for k in ("n_matches", "sum_r", "sum_s", "xor_fold", "mix_sum")}


class Timing(C.Structure):
    _fields_ = [("ms_total", C.c_float), ("ms_h2d", C.c_float), ("ms_d2h", C.c_float),
                ("ms_partition_build", C.c_float), ("ms_partition_probe", C.c_float),
                ("ms_hist", C.c_float), ("ms_scan", C.c_float), ("ms_scatter", C.c_float),
                ("ms_offsets", C.c_float), ("ms_probe_count", C.c_float),
                ("ms_out_scan", C.c_float), ("ms_probe_write", C.c_float), ("ms_order", C.c_float),
                ("radix_bits", C.c_int), ("radix_passes", C.c_int),
                ("n_scatter_launches", C.c_int), ("n_split_retries", C.c_int),
                ("bytes_scatter", C.c_uint64), ("bytes_hist", C.c_uint64),
                ("bytes_probe_count", C.c_uint64), ("bytes_probe_write", C.c_uint64),
                ("path", C.c_uint32), ("key_prefix_bits", C.c_int32), ("key_window_low", C.c_int32),
                ("n_probe_items", C.c_uint32), ("ms_scatter_pass0", C.c_float), ("ms_scatter_pass1", C.c_float)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class PlanDesc(C.Structure):
    """hmj_plan_desc: how the last join was planned and why (include/hmj.h)."""
    _fields_ = [("struct_size", C.c_uint32), ("path", C.c_uint32), ("radix_bits", C.c_int32), ("radix_passes", C.c_int32),
                ("pass_bits", C.c_int32 * 4), ("key_prefix_bits", C.c_int32), ("key_window_low", C.c_int32),
                ("n_partitions", C.c_uint32), ("probe_items", C.c_uint32), ("attempts", C.c_uint32), ("refused", C.c_uint32),
                ("cooling", C.c_uint32), ("workload", C.c_uint64)]

    def as_dict(self):
        d = {n: getattr(self, n) for n, _ in self._fields_}
        d["pass_bits"] = list(self.pass_bits)
        return d


# hmj_plan_desc.refused / .cooling bits
HMJ_REFUSED_GTABLE_SHAPE, HMJ_REFUSED_GTABLE_COOLING, HMJ_REFUSED_GTABLE_GAVE_UP = 0x1, 0x2, 0x4
HMJ_REFUSED_RANK_SORT_MODEL, HMJ_REFUSED_RANK_SORT_COOLING, HMJ_REFUSED_RANK_SORT_GAVE_UP = 0x8, 0x10, 0x20
HMJ_REFUSED_SLAB_SHAPE, HMJ_REFUSED_SLAB_COOLING, HMJ_REFUSED_SLAB_SORTED_INPUT, HMJ_REFUSED_SLAB_OVERFLOW = 0x40, 0x80, 0x100, 0x200
HMJ_REFUSED_FAST_WRITE_COOLING, HMJ_REFUSED_FAST_WRITE_GAVE_UP = 0x400, 0x800
HMJ_REFUSED_SLAB_PROBE_COOLING, HMJ_REFUSED_SLAB_PROBE_OVERFLOW = 0x1000, 0x2000
HMJ_REFUSED_PREFIX_VIOLATED, HMJ_REFUSED_EXPANSION_GAVE_UP = 0x4000, 0x8000
HMJ_COOL_UNIQ_WRITE, HMJ_COOL_SORTED_WRITE, HMJ_COOL_GTABLE, HMJ_COOL_GTABLE_WRITE = 0x1, 0x2, 0x4, 0x8
HMJ_COOL_RANK_SORT, HMJ_COOL_RANK_SORT_SLAB, HMJ_COOL_EXPANSION, HMJ_COOL_SORT_SLAB = 0x10, 0x20, 0x40, 0x80
HMJ_COOL_SLAB, HMJ_COOL_SLAB_PROBE, HMJ_COOL_ONE_PASS_WRITE, HMJ_COOL_EXACT_PREFIX = 0x100, 0x200, 0x400, 0x800
HMJ_COOL_RANK_RUNS = 0x1000


class PlaceInfo(C.Structure):
    _fields_ = [("name", C.c_char * 16), ("bytes", C.c_uint64), ("fill_TBps", C.c_float), ("candidates", C.c_int),
                ("ms_search", C.c_float), ("searched", C.c_int), ("aborted", C.c_int), ("budget_ms", C.c_float),
                ("cand_ms_alloc", C.c_float * 4), ("cand_ms_fill", C.c_float * 4), ("cand_TBps", C.c_float * 4)]


def lib_path():
    # HMJ_LIB: developer override to A/B an alternative build of the same library
    return os.environ.get("HMJ_LIB") or os.path.join(_HERE, "libhmj_hip.so")


class ExchangeInfo(C.Structure):
    _fields_ = [("n_ranks", C.c_int), ("owner_mode", C.c_int), ("rounds_build", C.c_uint32), ("rounds_probe", C.c_uint32),
                ("recv_build", C.c_uint64), ("recv_probe", C.c_uint64), ("ms_split", C.c_float),
                ("ms_exchange_build", C.c_float), ("ms_exchange_probe", C.c_float), ("ms_local", C.c_float),
                ("ms_total", C.c_float), ("digit_bits", C.c_int32), ("digit_low", C.c_int32), ("n_subjoins", C.c_uint32),
                ("fallback", C.c_uint32), ("sample_max_share", C.c_float), ("ms_kernels", C.c_float),
                ("ms_exposed", C.c_float)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


HMJ_MAX_RANKS, HMJ_MAX_ROUNDS = 16, 16


class DigitPlan(C.Structure):
    _fields_ = [("usable", C.c_int32), ("digit_bits", C.c_int32), ("digit_low", C.c_int32), ("n_rounds", C.c_uint32),
                ("owner_first", C.c_uint32 * (HMJ_MAX_RANKS + 1)),
                ("round_first", (C.c_uint32 * (HMJ_MAX_ROUNDS + 1)) * HMJ_MAX_RANKS), ("max_share", C.c_float)]


# hmj_transport: the two collectives a host may supply instead of RCCL (include/hmj.h)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_int)
ALLTOALLV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64),
                           C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.c_void_p)


class Transport(C.Structure):
    _fields_ = [("user", C.c_void_p), ("n_ranks", C.c_int), ("rank", C.c_int), ("allgather_u64", ALLGATHER_FN),
                ("alltoallv", ALLTOALLV_FN)]


_LIB = None


def load_library():
    """Load libhmj_hip.so.  torch (if used by the caller) must be imported BEFORE this so that both
    share one HIP runtime (the wheel bundles libamdhip64 under the same SONAME)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise HmjError(-100, "HIP extension not built: %s is missing (run `make` at the repo root)" % path)
    L = C.CDLL(path)
    u, i, vp, cp = C.c_uint64, C.c_int, C.c_void_p, C.c_char_p
    L.hmj_create.restype = i
    L.hmj_create.argtypes = [C.POINTER(vp), i]
    L.hmj_destroy.restype = None
    L.hmj_destroy.argtypes = [vp]
    L.hmj_set_stream.restype = i
    L.hmj_set_stream.argtypes = [vp, vp]
    L.hmj_reserve.restype = i
    L.hmj_reserve.argtypes = [vp, u, u, u, C.c_uint32]
    L.hmj_set_radix_bits.restype = i
    L.hmj_set_radix_bits.argtypes = [vp, i]
    L.hmj_autotune_radix_bits.restype = i
    L.hmj_autotune_radix_bits.argtypes = [vp, u, u, i, C.POINTER(C.c_int), C.POINTER(C.c_double)]
    L.hmj_set_key_prefix_bits.restype = i
    L.hmj_set_key_prefix_bits.argtypes = [vp, i]
    L.hmj_plan.restype = i
    L.hmj_plan.argtypes = [u, C.POINTER(i), C.POINTER(i), C.POINTER(i * 4)]
    L.hmj_set_profiling.restype = i
    L.hmj_set_profiling.argtypes = [vp, i]
    L.hmj_last_timing.restype = i
    L.hmj_last_timing.argtypes = [vp, C.POINTER(Timing)]
    L.hmj_last_plan.restype = i
    L.hmj_last_plan.argtypes = [vp, C.POINTER(PlanDesc)]
    L.hmj_forget_workloads.restype = i
    L.hmj_forget_workloads.argtypes = [vp]
    L.hmj_placement_info.restype = i
    L.hmj_placement_info.argtypes = [vp, C.POINTER(PlaceInfo), i]
    L.hmj_strerror.restype = cp
    L.hmj_strerror.argtypes = [i]
    L.hmj_last_error.restype = cp
    L.hmj_last_error.argtypes = [vp]
    L.hmj_version.restype = cp
    L.hmj_version.argtypes = []
    L.hmj_abi_version.restype = i
    L.hmj_abi_version.argtypes = []
    L.hmj_join_u64_device.restype = i
    L.hmj_join_u64_device.argtypes = [vp, vp, u, vp, u, C.c_uint32, C.POINTER(JoinResult)]
    L.hmj_prepare_build_u64_device.restype = i
    L.hmj_prepare_build_u64_device.argtypes = [vp, vp, u, u]
    L.hmj_join_u64.restype = i
    L.hmj_join_u64.argtypes = [vp, vp, u, vp, u, C.c_uint32, C.POINTER(JoinResult)]
    L.hmj_join_u64_rows.restype = i
    L.hmj_join_u64_rows.argtypes = [vp, vp, u, vp, u, C.c_uint32, C.POINTER(JoinResult), C.POINTER(vp)]
    L.hmj_rows_free.restype = None
    L.hmj_rows_free.argtypes = [vp]
    L.hmj_host_pool_trim.restype = u
    L.hmj_host_pool_trim.argtypes = [u]
    L.hmj_host_pool_bytes.restype = u
    L.hmj_host_pool_bytes.argtypes = []
    L.hmj_set_host_threads.restype = i
    L.hmj_set_host_threads.argtypes = [vp, i]
    L.hmj_release_result.restype = None
    L.hmj_release_result.argtypes = [vp]
    L.hmj_comm_unique_id.restype = i
    L.hmj_comm_unique_id.argtypes = [vp]
    L.hmj_comm_init_rank.restype = i
    L.hmj_comm_init_rank.argtypes = [vp, i, i, vp]
    L.hmj_comm_set_transport.restype = i
    L.hmj_comm_set_transport.argtypes = [vp, C.POINTER(Transport)]
    L.hmj_comm_destroy.restype = i
    L.hmj_comm_destroy.argtypes = [vp]
    L.hmj_comm_set_message_bytes.restype = i
    L.hmj_comm_set_message_bytes.argtypes = [vp, u, u]
    L.hmj_exchange_join_u64_device.restype = i
    L.hmj_exchange_join_u64_device.argtypes = [vp, vp, u, vp, u, C.c_uint32, C.POINTER(JoinResult), C.POINTER(JoinResult)]
    L.hmj_owner_split_u64_device.restype = i
    L.hmj_owner_split_u64_device.argtypes = [vp, vp, u, i, _U64P, vp, vp]
    L.hmj_last_exchange_info.restype = i
    L.hmj_last_exchange_info.argtypes = [vp, C.POINTER(ExchangeInfo)]
    L.hmj_comm_set_timeout_ms.restype = i
    L.hmj_comm_set_timeout_ms.argtypes = [vp, u]
    L.hmj_comm_get_timeout_ms.restype = i
    L.hmj_comm_get_timeout_ms.argtypes = [vp, _U64P]
    L.hmj_comm_set_owner_path.restype = i
    L.hmj_comm_set_owner_path.argtypes = [vp, i]
    L.hmj_comm_set_self_exchange.restype = i
    L.hmj_comm_set_self_exchange.argtypes = [vp, i]
    L.hmj_exchange_digit_plan.restype = i
    L.hmj_exchange_digit_plan.argtypes = [i, _U64P, u, C.c_uint32, C.POINTER(DigitPlan)]
    L.hmj_exchange_digit_layout.restype = i
    L.hmj_exchange_digit_layout.argtypes = [i, i, C.POINTER(DigitPlan), _U64P, _U64P, _U64P, _U64P, _U64P, _U64P]
    L.hmj_exchange_rounds.restype = C.c_uint32
    L.hmj_exchange_rounds.argtypes = [i, _U64P, u]
    L.hmj_exchange_layout.restype = i
    L.hmj_exchange_layout.argtypes = [i, i, _U64P, C.c_uint32, i, _U64P, _U64P, _U64P, _U64P, _U64P]
    L.hmj_partition_u64_device.restype = i
    L.hmj_partition_u64_device.argtypes = [vp, vp, u, i, i, vp, vp]
    L.hmj_sort_u64_device.restype = i
    L.hmj_sort_u64_device.argtypes = [vp, vp, u, vp]
    L.hmj_gen_build_u64_device.restype = i
    L.hmj_gen_build_u64_device.argtypes = [vp, vp, u, u, u]
    L.hmj_gen_probe_u64_device.restype = i
    L.hmj_gen_probe_u64_device.argtypes = [vp, vp, u, u, u, u, u]
    L.hmj_gen_from_cdf_u64_device.restype = i
    L.hmj_gen_from_cdf_u64_device.argtypes = [vp, vp, u, u, vp, u, u, u]
    L.hmj_gen_uniform_domain_u64_device.restype = i
    L.hmj_gen_uniform_domain_u64_device.argtypes = [vp, vp, u, u, u, u, u]
    _LIB = L
    return L
