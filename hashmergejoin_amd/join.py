"""Host-side mirror of the reference's join operator over the C ABI.

`Executor` owns one hmj_ctx (one per GPU / per process).  `HashMergeJoin` mirrors
HashMergeJoin<RIter,SIter> (reference hashjoin.h:33-199): construct from two relations, iterate
(key, rval, sval) in ascending-key order, `clear()`.  torch is used only to hold device memory
and to name the current stream.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import (HMJ_CHECKSUM, HMJ_FIRST_WINS, HMJ_MATERIALIZE, HMJ_ORDERED, HMJ_SUM_PROBE,
                   HmjError, JoinResult, Timing)

SEED_B = 0x243F6A8885A308D3


def plan(n_build):
    """(total_bits, [bits per LSD pass]) the executor will use for a build side of n_build rows."""
    L = _lib.load_library()
    tb, npass, pb = C.c_int(0), C.c_int(0), (C.c_int * 4)()
    rc = L.hmj_plan(n_build, C.byref(tb), C.byref(npass), C.byref(pb))
    if rc:
        raise HmjError(rc, "hmj_plan")
    return tb.value, [pb[i] for i in range(npass.value)]


def _dev_ptr(t):
    """(pointer, rows) of a CUDA/HIP torch tensor holding n x {key,val} as int64/uint64 [n,2]."""
    if t is None:
        return None, 0
    if not t.is_cuda:
        raise ValueError("expected a device tensor")
    if not t.is_contiguous() or t.element_size() != 8 or t.dim() != 2 or t.shape[1] != 2:
        raise ValueError("relation must be a contiguous [n,2] 64-bit tensor {key,val}")
    return t.data_ptr(), t.shape[0]


class Executor:
    """One hmj_ctx.  All methods raise HmjError on a nonzero status."""

    def __init__(self, device=None, use_torch_stream=True):
        import torch  # noqa: F401  (must be loaded before the HIP library, see _lib.load_library)

        self._torch = torch
        self.L = _lib.load_library()
        if device is None:
            device = torch.cuda.current_device()
        self.device = int(device)
        h = C.c_void_p()
        rc = self.L.hmj_create(C.byref(h), self.device)
        if rc:
            raise HmjError(rc, self.L.hmj_strerror(rc).decode())
        self.h = h
        self.use_torch_stream = use_torch_stream
        self._keep = None

    def close(self):
        if getattr(self, "h", None):
            self.L.hmj_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc:
            raise HmjError(rc, "%s (%s)" % (self.L.hmj_strerror(rc).decode(), self.L.hmj_last_error(self.h).decode()))

    def _sync_stream(self):
        # run on torch's current stream so the join is ordered after whatever produced its inputs
        # (handle 0 = the HIP default stream, which hmj_set_stream takes as such)
        if self.use_torch_stream:
            s = self._torch.cuda.current_stream(self.device).cuda_stream
            self._check(self.L.hmj_set_stream(self.h, C.c_void_p(s)))

    # ---- configuration -----------------------------------------------------------------------
    def reserve(self, n_build, n_probe, max_matches=0, flags=0):
        self._check(self.L.hmj_reserve(self.h, n_build, n_probe, max_matches, flags))

    def set_radix_bits(self, bits):
        self._check(self.L.hmj_set_radix_bits(self.h, -1 if bits is None else bits))

    def set_key_prefix_bits(self, bits):
        """Top `bits` key bits are constant in both relations (consumed by an outer split); -1 = sample."""
        self._check(self.L.hmj_set_key_prefix_bits(self.h, bits))

    def set_profiling(self, on=True):
        self._check(self.L.hmj_set_profiling(self.h, int(bool(on))))

    def last_timing(self):
        t = Timing()
        self._check(self.L.hmj_last_timing(self.h, C.byref(t)))
        return t.as_dict()

    def last_plan(self):
        """hmj_last_plan: path, bits per pass, attempts, why faster formulations were refused, what the workload is
        skipping from now on (dict of hmj_plan_desc's fields)."""
        p = _lib.PlanDesc()
        p.struct_size = C.sizeof(_lib.PlanDesc)
        self._check(self.L.hmj_last_plan(self.h, C.byref(p)))
        return p.as_dict()

    def forget_workloads(self):
        self._check(self.L.hmj_forget_workloads(self.h))

    def placement_info(self):
        """One dict per big partition buffer this executor probed (hmj_placement_info; empty with HMJ_PLACE=0):
        name, bytes, fill rate of the allocation kept, candidates tried, whether a search ran (hmj_reserve or
        HMJ_PLACE=n) and was cut short by its budget, and per candidate the hipMalloc ms, fill ms and fill rate."""
        arr = (_lib.PlaceInfo * 16)()
        n = self.L.hmj_placement_info(self.h, arr, 16)
        out = []
        for i in range(n):
            e, k = arr[i], int(arr[i].candidates)
            out.append({"name": e.name.decode(), "bytes": int(e.bytes), "fill_TBps": round(float(e.fill_TBps), 3),
                        "candidates": k, "ms_search": round(float(e.ms_search), 2), "searched": bool(e.searched),
                        "aborted": bool(e.aborted), "budget_ms": round(float(e.budget_ms), 1),
                        "cand_ms_alloc": [round(float(x), 2) for x in e.cand_ms_alloc[:k]],
                        "cand_ms_fill": [round(float(x), 2) for x in e.cand_ms_fill[:k]],
                        "cand_TBps": [round(float(x), 3) for x in e.cand_TBps[:k]]})
        return out

    # ---- joins -------------------------------------------------------------------------------
    def join_device(self, build, probe, flags=0):
        """build/probe: device tensors [n,2] {key,val}.  Returns JoinResult (columns are device
        pointers owned by the executor; see `columns`)."""
        self._sync_stream()
        bp, nb = _dev_ptr(build)
        pp, np_ = _dev_ptr(probe)
        res = JoinResult()
        self._check(self.L.hmj_join_u64_device(self.h, C.c_void_p(bp), nb, C.c_void_p(pp), np_, flags, C.byref(res)))
        return res

    def prepare_build(self, build, n_probe_hint):
        """Partition the build side now; the next matching plain-count join_device skips that work."""
        self._sync_stream()
        bp, nb = _dev_ptr(build)
        self._check(self.L.hmj_prepare_build_u64_device(self.h, C.c_void_p(bp), nb, n_probe_hint))

    def join_host(self, build, probe, flags=0):
        """build/probe: numpy uint64 [n,2] in host memory (the reference ctor's situation)."""
        self._sync_stream()
        b = np.ascontiguousarray(build, np.uint64).reshape(-1, 2)
        p = np.ascontiguousarray(probe, np.uint64).reshape(-1, 2)
        res = JoinResult()
        self._check(self.L.hmj_join_u64(self.h, b.ctypes.data_as(C.c_void_p), len(b), p.ctypes.data_as(C.c_void_p),
                                        len(p), flags, C.byref(res)))
        return res

    def columns_to_numpy(self, res, host):
        """Copy the result columns out as an [n,3] uint64 array of (key, rval, sval)."""
        n = int(res.n_matches)
        out = np.empty((n, 3), np.uint64)
        if n == 0 or not res.key:
            return out[:0] if not res.key else out
        if host:
            for c, ptr in enumerate((res.key, res.rval, res.sval)):
                out[:, c] = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint64)), shape=(n,))
        else:
            torch = self._torch
            tmp = torch.empty(n, dtype=torch.int64, device="cuda:%d" % self.device)
            for c, ptr in enumerate((res.key, res.rval, res.sval)):
                _memcpy_d2d(torch, tmp, ptr, n * 8)
                out[:, c] = tmp.cpu().numpy().view(np.uint64)
        return out

    def release_result(self):
        self.L.hmj_release_result(self.h)

    # ---- multi-GPU exchange (hmj_comm_* / hmj_exchange_join_u64_device) -------------------------
    def comm_init_rccl(self, n_ranks, rank, unique_id):
        """unique_id: the 128 bytes rank 0 got from hashmergejoin_amd.dist.new_unique_id()."""
        buf = (C.c_char * 128).from_buffer_copy(bytes(unique_id))
        self._check(self.L.hmj_comm_init_rank(self.h, n_ranks, rank, C.cast(buf, C.c_void_p)))

    def comm_set_transport(self, transport):
        """transport: a _lib.Transport whose callbacks stay alive as long as this executor uses them."""
        self._keep = transport
        self._check(self.L.hmj_comm_set_transport(self.h, C.byref(transport.struct)))

    def comm_set_self_exchange(self, on=True):
        """One-rank communicators: run the whole exchange path (tests / rehearsals) instead of the plain local join."""
        self._check(self.L.hmj_comm_set_self_exchange(self.h, int(bool(on))))

    def comm_set_owner_path(self, split=False):
        """Non-ordered distributed joins: digit-range owners with per-round joins (default) or, split=True, round 2's
        hash owner with its separate owner split and one local join (hmj_comm_set_owner_path)."""
        self._check(self.L.hmj_comm_set_owner_path(self.h, 1 if split else 0))

    def comm_set_timeout_ms(self, timeout_ms):
        """Deadline of one exchange step (hmj_comm_set_timeout_ms; 0 = wait for ever): past it the step returns
        HMJ_E_TIMEOUT and the communicator is unusable."""
        self._check(self.L.hmj_comm_set_timeout_ms(self.h, int(timeout_ms)))

    def comm_get_timeout_ms(self):
        v = C.c_uint64(0)
        self._check(self.L.hmj_comm_get_timeout_ms(self.h, C.byref(v)))
        return int(v.value)

    def comm_set_message_bytes(self, max_message_bytes=0, probe_round_bytes=0):
        self._check(self.L.hmj_comm_set_message_bytes(self.h, max_message_bytes, probe_round_bytes))

    def owner_split(self, rel, n_ranks, splitters=None):
        """hmj_owner_split_u64_device: (rows grouped by owner, [G'+1] offsets) with G' = 2^ceil(log2 n_ranks)."""
        torch = self._torch
        self._sync_stream()
        ptr, n = _dev_ptr(rel)
        out = torch.empty_like(rel)
        nd = 1
        while nd < n_ranks:
            nd *= 2
        off = torch.empty(nd + 1, dtype=torch.int64, device=rel.device)
        sp = None
        if splitters is not None:
            sp = (C.c_uint64 * (n_ranks - 1))(*[int(x) for x in splitters])
        self._check(self.L.hmj_owner_split_u64_device(self.h, C.c_void_p(ptr), n, n_ranks, sp, C.c_void_p(out.data_ptr()),
                                                      C.c_void_p(off.data_ptr())))
        return out, off

    def exchange_join(self, build_shard, probe_shard, flags=0):
        """Distributed join of row shards (collective).  Returns (local JoinResult, global JoinResult)."""
        self._sync_stream()
        bp, nb = _dev_ptr(build_shard)
        pp, np_ = _dev_ptr(probe_shard)
        loc, glob = JoinResult(), JoinResult()
        self._check(self.L.hmj_exchange_join_u64_device(self.h, C.c_void_p(bp), nb, C.c_void_p(pp), np_, flags,
                                                        C.byref(loc), C.byref(glob)))
        return loc, glob

    def last_exchange_info(self):
        info = _lib.ExchangeInfo()
        self._check(self.L.hmj_last_exchange_info(self.h, C.byref(info)))
        return info.as_dict()

    # ---- single radix pass (hmj_partition_u64_device) ------------------------------------------
    def partition_device(self, rel, shift, bits):
        torch = self._torch
        self._sync_stream()
        ptr, n = _dev_ptr(rel)
        out = torch.empty_like(rel)
        off = torch.empty((1 << bits) + 1, dtype=torch.int64, device=rel.device)
        self._check(self.L.hmj_partition_u64_device(self.h, C.c_void_p(ptr), n, shift, bits, C.c_void_p(out.data_ptr()),
                                                    C.c_void_p(off.data_ptr())))
        return out, off

    def autotune(self, n_build, n_probe, apply=True):
        """Time the join of synthetic relations with B-1, B, B+1 radix bits (hmj_autotune_radix_bits).
        Returns (best_bits, {bits: ms}); apply=True keeps the fastest as this executor's plan."""
        self._sync_stream()
        best, ms = C.c_int(0), (C.c_double * 3)()
        self._check(self.L.hmj_autotune_radix_bits(self.h, n_build, n_probe, 1 if apply else 0, C.byref(best), ms))
        b0 = plan(n_build)[0]
        return best.value, {b0 - 1 + k: ms[k] for k in range(3) if ms[k] >= 0}

    def sort_device(self, rel, inplace=False):
        """Full ascending-key sort of a device relation (hmj_sort_u64_device); returns a new tensor, or
        sorts `rel` itself with inplace=True (the radix_int_inplace replacement)."""
        torch = self._torch
        self._sync_stream()
        ptr, n = _dev_ptr(rel)
        out = rel if inplace else torch.empty_like(rel)
        self._check(self.L.hmj_sort_u64_device(self.h, C.c_void_p(ptr), n, C.c_void_p(out.data_ptr())))
        return out

    # ---- generators ----------------------------------------------------------------------------
    def _alloc(self, n):
        torch = self._torch
        return torch.empty((n, 2), dtype=torch.int64, device="cuda:%d" % self.device)

    def gen_build(self, n, start=0, seed=SEED_B):
        self._sync_stream()
        t = self._alloc(n)
        self._check(self.L.hmj_gen_build_u64_device(self.h, C.c_void_p(t.data_ptr()), n, start, seed))
        return t

    def gen_probe(self, n, n_build, start=0, seed=SEED_B, miss_mod=0):
        self._sync_stream()
        t = self._alloc(n)
        self._check(self.L.hmj_gen_probe_u64_device(self.h, C.c_void_p(t.data_ptr()), n, start, n_build, seed, miss_mod))
        return t

    def gen_from_cdf(self, n, thr_dev, start=0, seed=SEED_B, zseed=0x1234567):
        self._sync_stream()
        t = self._alloc(n)
        self._check(self.L.hmj_gen_from_cdf_u64_device(self.h, C.c_void_p(t.data_ptr()), n, start,
                                                       C.c_void_p(thr_dev.data_ptr()), thr_dev.numel(), seed, zseed))
        return t

    def gen_uniform_domain(self, n, domain, start=0, seed=SEED_B, zseed=0x7654321):
        self._sync_stream()
        t = self._alloc(n)
        self._check(self.L.hmj_gen_uniform_domain_u64_device(self.h, C.c_void_p(t.data_ptr()), n, start, domain, seed, zseed))
        return t


def _memcpy_d2d(torch, dst_tensor, src_ptr, nbytes):
    """Copy nbytes from a raw device pointer into a torch tensor (plumbing only)."""
    hip = C.CDLL(None)  # the HIP runtime torch already loaded
    fn = getattr(hip, "hipMemcpy", None)
    if fn is None:
        import ctypes.util

        hip = C.CDLL("libamdhip64.so")
        fn = hip.hipMemcpy
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    torch.cuda.synchronize()
    rc = fn(C.c_void_p(dst_tensor.data_ptr()), C.c_void_p(src_ptr), nbytes, 3)  # hipMemcpyDeviceToDevice
    if rc:
        raise HmjError(-4, "hipMemcpy d2d failed: %d" % rc)


class HashMergeJoin:
    """Python mirror of the reference operator (hashjoin.h:33-199).

        hmj = HashMergeJoin(r, s, num_threads=1)     # hashjoin.h:56-58
        for key, rval, sval in hmj: ...              # begin()/end(), hashjoin.h:183-191
        hmj.clear()                                  # hashjoin.h:192-195

    r, s: [n,2] uint64 relations {key,val}, numpy (host) or torch (device).  `num_threads` is
    accepted for signature compatibility; the work runs on the GPU.  Iteration order is ascending
    key, as the reference's.  Keys must be unique within each relation for bit-exact parity with
    the reference iterator (its behaviour on duplicates is not relational).
    """

    def __init__(self, r=None, s=None, num_threads=1, executor=None, flags=0):
        self.num_threads = num_threads
        self._rows = None
        self.result = None
        if r is None and s is None:  # HashMergeJoin() = default, hashjoin.h:55
            return
        self._ex = executor or Executor()
        fl = HMJ_MATERIALIZE | HMJ_ORDERED | flags
        host = isinstance(r, np.ndarray)
        res = self._ex.join_host(r, s, fl) if host else self._ex.join_device(r, s, fl)
        self.result = res
        self._rows = self._ex.columns_to_numpy(res, host)

    def __iter__(self):
        if self._rows is None:
            return iter(())
        return (tuple(int(x) for x in row) for row in self._rows)

    def __len__(self):
        return 0 if self._rows is None else len(self._rows)

    def rows(self):
        """All result rows as an [n,3] uint64 array (key, rval, sval)."""
        return np.empty((0, 3), np.uint64) if self._rows is None else self._rows

    def clear(self):
        self._rows = None
        self.result = None
