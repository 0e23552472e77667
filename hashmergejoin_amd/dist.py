"""Multi-GPU use of the executor: one process per GPU, the partition exchange itself is inside
libhmj_hip.so (hmj_exchange_join_u64_device, include/hmj.h; csrc/exchange.hip).  This module only

  * sets up the communicator of an Executor from a torch.distributed process group -- RCCL over xGMI when the
    group's backend is "nccl" (rank 0's unique id is broadcast through the group), otherwise the library's
    callback transport over the group's own collectives (gloo: CPU tests, several ranks sharing one GPU);
  * mirrors, in numpy, the owner function and the round plan for tests that have no GPU.

Nothing here computes a join.  Reference counterpart: the fork-join over threads inside the ctor
(hashjoin.h:56-68 -> radix_hash.h:375-405); here the fork-join is over GPUs.
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import _lib

_M64 = (1 << 64) - 1


def mix64(x):
    """numpy uint64 mirror of hmj_dev.h mix64 (same constants as the oracle)."""
    x = np.asarray(x, dtype=np.uint64).copy()
    x ^= x >> np.uint64(30)
    x *= np.uint64(0xBF58476D1CE4E5B9)
    x ^= x >> np.uint64(27)
    x *= np.uint64(0x94D049BB133111EB)
    x ^= x >> np.uint64(31)
    return x


def owner_of(keys, n_ranks, splitters=None):
    """Owner rank of every key, as radix.hip owner_digit computes it: floor(mix64(key) * G / 2^64), or with
    splitters (ordered mode) the number of splitters <= key."""
    keys = np.asarray(keys, dtype=np.uint64)
    if splitters is not None:
        return np.searchsorted(np.asarray(splitters, dtype=np.uint64), keys, side="right").astype(np.int64)
    m = mix64(keys)
    hi, lo = m >> np.uint64(32), m & np.uint64(0xFFFFFFFF)
    g = np.uint64(n_ranks)
    return ((hi * g + ((lo * g) >> np.uint64(32))) >> np.uint64(32)).astype(np.int64)


def exchange_plan(counts, rank, max_msg_rows, layout):
    """hmj_exchange_rounds + hmj_exchange_layout (pure host arithmetic inside the library).
    counts: [G,G] matrix, counts[src][dst].  Returns dict of [n_rounds,G] uint64 arrays + round_end."""
    L = _lib.load_library()
    m = np.ascontiguousarray(counts, dtype=np.uint64)
    G = m.shape[0]
    p = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint64))
    R = int(L.hmj_exchange_rounds(G, p(m), max_msg_rows))
    out = {k: np.zeros((R, G), np.uint64) for k in ("send_off", "send_rows", "recv_off", "recv_rows")}
    out["round_end"] = np.zeros(R, np.uint64)
    rc = L.hmj_exchange_layout(G, rank, p(m), R, layout, p(out["send_off"]), p(out["send_rows"]), p(out["recv_off"]),
                               p(out["recv_rows"]), p(out["round_end"]))
    if rc:
        raise _lib.HmjError(rc, "hmj_exchange_layout")
    out["n_rounds"] = R
    return out


def digit_plan(sample_keys, n_ranks, n_rounds=1):
    """hmj_exchange_digit_plan: the digit window, the digit ranges the ranks own and the digit ranges of the rounds,
    from a pooled key sample (pure host arithmetic inside the library).  Returns the ctypes struct."""
    L = _lib.load_library()
    k = np.ascontiguousarray(sample_keys, dtype=np.uint64)
    plan = _lib.DigitPlan()
    rc = L.hmj_exchange_digit_plan(n_ranks, k.ctypes.data_as(C.POINTER(C.c_uint64)), len(k), n_rounds, C.byref(plan))
    if rc:
        raise _lib.HmjError(rc, "hmj_exchange_digit_plan")
    return plan


def digit_of(keys, plan):
    """The first radix pass's digit of every key under `plan` (radix.hip digit_of)."""
    keys = np.asarray(keys, dtype=np.uint64)
    return ((keys >> np.uint64(plan.digit_low)) & np.uint64((1 << plan.digit_bits) - 1)).astype(np.int64)


def digit_layout(plan, counts, rank):
    """hmj_exchange_digit_layout.  counts: [G, 2^digit_bits] rows per (source rank, digit).  Returns a dict of
    [n_rounds, G] uint64 arrays send_off / send_rows / recv_off / recv_rows and round_off [n_rounds + 1]."""
    L = _lib.load_library()
    m = np.ascontiguousarray(counts, dtype=np.uint64)
    G, R = m.shape[0], plan.n_rounds
    p = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint64))
    out = {k: np.zeros((R, G), np.uint64) for k in ("send_off", "send_rows", "recv_off", "recv_rows")}
    out["round_off"] = np.zeros(R + 1, np.uint64)
    rc = L.hmj_exchange_digit_layout(G, rank, C.byref(plan), p(m), p(out["send_off"]), p(out["send_rows"]), p(out["recv_off"]),
                                     p(out["recv_rows"]), p(out["round_off"]))
    if rc:
        raise _lib.HmjError(rc, "hmj_exchange_digit_layout")
    return out


def _hip():
    hip = C.CDLL(None)
    if not hasattr(hip, "hipMemcpy"):
        hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.restype = C.c_int
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipStreamSynchronize.restype = C.c_int
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    return hip


class GroupTransport:
    """hmj_transport over a torch.distributed group's point-to-point operations.  device=True: the pointers the
    library passes are device memory and are staged through host tensors (rehearsal of several ranks on one GPU,
    where RCCL refuses duplicate devices); device=False: host pointers (CPU tests of the exchange plan)."""

    def __init__(self, group=None, device=True, timeout_s=None):
        self.group, self.device = group, device
        # every wait on a peer is bounded by this (hmj_comm_set_timeout_ms tells the library the same number): a peer
        # that never takes part makes the callback return HMJ_E_TIMEOUT instead of blocking for ever
        self.timeout_s = timeout_s
        self.n_ranks, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.hip = _hip() if device else None
        self.rounds_seen = 0
        self.struct = _lib.Transport(None, self.n_ranks, self.rank, _lib.ALLGATHER_FN(self._allgather),
                                     _lib.ALLTOALLV_FN(self._alltoallv))

    def _peer(self, g):
        return g if self.group is None else dist.get_global_rank(self.group, g)

    def _allgather(self, user, send, recv, count):
        try:
            mine = torch.from_numpy(np.ctypeslib.as_array(send, shape=(count,)).view(np.int64).copy())
            outs = [torch.empty(count, dtype=torch.int64) for _ in range(self.n_ranks)]
            work = dist.all_gather(outs, mine, group=self.group, async_op=True)
            if not self._wait([work]):
                return _lib.HMJ_E_TIMEOUT
            np.ctypeslib.as_array(recv, shape=(count * self.n_ranks,))[:] = torch.cat(outs).numpy().view(np.uint64)
            return 0
        except Exception as e:  # an exception must not unwind through the C frames
            print("GroupTransport.allgather failed:", repr(e), flush=True)
            return 1

    def _wait(self, works):
        """Wait for all of `works` under one deadline; False = it passed (the peers never showed up)."""
        import datetime
        import time

        if not self.timeout_s:
            for w in works:
                w.wait()
            return True
        until = time.monotonic() + self.timeout_s
        for w in works:
            left = until - time.monotonic()
            try:
                if left <= 0 or not w.wait(datetime.timedelta(seconds=max(left, 0.001))):
                    raise RuntimeError("timed out")
            except RuntimeError as e:
                print("GroupTransport rank %d: gave up waiting for a peer after %.1f s (%s)" % (self.rank, self.timeout_s, str(e)[:80]), flush=True)
                return False
        return True

    def _copy(self, dst, src, nbytes, kind):
        if self.device:
            rc = self.hip.hipMemcpy(C.c_void_p(dst), C.c_void_p(src), nbytes, kind)
            if rc:
                raise RuntimeError("hipMemcpy failed: %d" % rc)
        else:
            C.memmove(dst, src, nbytes)

    def _alltoallv(self, user, rnd, sp, sb, rp, rb, stream):
        try:
            G, me = self.n_ranks, self.rank
            if self.device and self.hip.hipStreamSynchronize(C.c_void_p(stream)):
                raise RuntimeError("hipStreamSynchronize failed")
            self.rounds_seen += 1
            reqs, outs, keep = [], {}, []
            for g in range(G):
                if g == me:
                    if sb[g]:
                        self._copy(rp[g], sp[g], sb[g], 3)  # device to device (host: memmove)
                    continue
                if sb[g]:
                    t = torch.empty(sb[g], dtype=torch.uint8)
                    self._copy(t.data_ptr(), sp[g], sb[g], 2)  # device to host
                    keep.append(t)
                    reqs.append(dist.isend(t, dst=self._peer(g), group=self.group))
                if rb[g]:
                    outs[g] = torch.empty(rb[g], dtype=torch.uint8)
                    reqs.append(dist.irecv(outs[g], src=self._peer(g), group=self.group))
            if not self._wait(reqs):
                return _lib.HMJ_E_TIMEOUT
            for g, t in outs.items():
                self._copy(rp[g], t.data_ptr(), rb[g], 1)  # host to device
            return 0
        except Exception as e:
            print("GroupTransport.alltoallv failed:", repr(e), flush=True)
            return 1


def new_unique_id():
    """128-byte RCCL unique id (rank 0 calls this and distributes it)."""
    L = _lib.load_library()
    buf = (C.c_char * 128)()
    rc = L.hmj_comm_unique_id(C.cast(buf, C.c_void_p))
    if rc:
        raise _lib.HmjError(rc, "hmj_comm_unique_id (is librccl available?)")
    return bytes(buf)


def init_comm(ex, group=None, force_transport=None, timeout_s=120.0):
    """Give `ex` (an Executor) the communicator of a torch.distributed group.  backend nccl -> RCCL inside the
    library (its own communicator over xGMI; torch's group only carries the 128-byte id); anything else -> the
    callback transport over the group.  force_transport: "rccl" | "group".
    timeout_s: the deadline of one exchange step (hmj_comm_set_timeout_ms; None / 0 = wait for ever): a step that a
    peer never joins returns HMJ_E_TIMEOUT on every other rank instead of blocking."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    kind = force_transport or ("rccl" if dist.get_backend(group) == "nccl" else "group")
    if kind == "rccl":
        dev = torch.device("cuda", ex.device)
        on_dev = dist.get_backend(group) == "nccl"
        t = torch.zeros(128, dtype=torch.uint8, device=dev if on_dev else "cpu")
        if rank == 0:
            t.copy_(torch.frombuffer(bytearray(new_unique_id()), dtype=torch.uint8))
        if world > 1:
            dist.broadcast(t, src=0 if group is None else dist.get_global_rank(group, 0), group=group)
        ex.comm_init_rccl(world, rank, bytes(t.cpu().numpy().tobytes()))
    else:
        ex.comm_set_transport(GroupTransport(group, device=True, timeout_s=timeout_s))
    ex.comm_set_timeout_ms(int((timeout_s or 0) * 1000))
    return kind


def init_comm_single(ex, self_exchange=True, timeout_s=None):
    """One rank, RCCL transport.  self_exchange=True: the whole exchange path with self send/recv (tests,
    HMJ_FORCE_DIST); False: what a one-rank job does by default -- the plain local join, nothing crosses RCCL."""
    ex.comm_init_rccl(1, 0, new_unique_id())
    ex.comm_set_self_exchange(self_exchange)
    if timeout_s is not None:
        ex.comm_set_timeout_ms(int(timeout_s * 1000))


def distributed_join(ex, r_shard, s_shard, flags=0):
    """Each rank holds a row shard of R (build) and S (probe) on its GPU; init_comm was called.
    Returns (local JoinResult for the keys this rank owns, global checks dict)."""
    loc, glob = ex.exchange_join(r_shard, s_shard, flags)
    return loc, glob.checks()
