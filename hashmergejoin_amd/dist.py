"""Multi-GPU radix partition exchange.  One process per GPU; the radix fan-out
shards the join: the owner of a row is the top log2(G) bits of its key -- the same
most-significant-bits rule the reference partitions with (radix_hash.h:369, partitioned_hash.h:102)
-- so rank g ends up with key range g and the per-rank results concatenate in key order.

Exchange = counts all-to-all + one all_to_all_single (grouped send/recv) per relation over
torch.distributed (backend "nccl" is RCCL over xGMI on ROCm; "gloo" on CPU for tests).  The local
split is the HIP radix pass (hmj_partition_u64_device); the local join is hmj_join_u64_device.
Nothing here computes a join on the CPU.
"""
import torch
import torch.distributed as dist


def owner_bits(world_size):
    b = world_size.bit_length() - 1
    if world_size < 1 or (1 << b) != world_size:
        raise ValueError("world size must be a power of two (owner = top log2(G) key bits)")
    return b


def split_counts_from_offsets(offsets):
    """offsets: [G+1] bucket starts -> python list of G row counts."""
    o = offsets.to("cpu", torch.int64)
    return [int(x) for x in (o[1:] - o[:-1]).tolist()]


# RCCL (2.26) truncates a single all-to-all message of 2 GiB or more (seen on MI355X: a 2^27-row bucket
# arrives half empty).  Buckets are therefore sent in rounds of at most MAX_MSG_BYTES per peer; the
# receive side places every round's slice at its final position, so the received rows stay grouped by
# source rank in shard order (= global input order, which HMJ_FIRST_WINS relies on).
MAX_MSG_BYTES = 1 << 30


class _Works:
    """wait() for every queued round (they run in order on the communicator's stream)."""

    def __init__(self, works):
        self.works = works

    def wait(self):
        for w in self.works:
            w.wait()


def _starts(counts):
    out, acc = [], 0
    for c in counts:
        out.append(acc)
        acc += c
    return out


def _exchange_counts(parted, send_counts, group):
    """All ranks learn the whole G x G count matrix (one tiny all-gather): returns (this rank's receive
    counts, the largest single message of the exchange in rows) -- the latter so that every rank splits
    the exchange into the same number of rounds."""
    world = dist.get_world_size(group)
    sc = torch.tensor(send_counts, dtype=torch.int64, device=parted.device)
    rows = [torch.empty(world, dtype=torch.int64, device=parted.device) for _ in range(world)]
    dist.all_gather(rows, sc, group=group)
    m = torch.stack(rows).cpu()  # m[src][dst]
    me = dist.get_rank(group)
    return [int(x) for x in m[:, me].tolist()], int(m.max().item())


def _list_all_to_all(outs, ins, group, async_op):
    """dist.all_to_all on tensor lists; gloo has none, so CPU tests get point-to-point transfers."""
    if dist.get_backend(group) == "nccl":
        return dist.all_to_all(outs, ins, group=group, async_op=async_op)
    me, reqs = dist.get_rank(group), []
    for g in range(dist.get_world_size(group)):
        if g == me:
            outs[g].copy_(ins[g])
            continue
        peer = g if group is None else dist.get_global_rank(group, g)
        reqs.append(dist.isend(ins[g].contiguous(), dst=peer, group=group))
        reqs.append(dist.irecv(outs[g], src=peer, group=group))
    for q in reqs:
        q.wait()
    return None


def _exchange_data(parted, send_counts, recv_counts, biggest_rows, group, async_op):
    """Queue the data exchange; returns (rows, work or None)."""
    parted = parted.contiguous()
    row_bytes = parted.shape[1] * parted.element_size()
    out = torch.empty((sum(recv_counts), parted.shape[1]), dtype=parted.dtype, device=parted.device)
    rounds = max(1, -(-(biggest_rows * row_bytes) // MAX_MSG_BYTES))
    if rounds == 1:
        work = dist.all_to_all_single(out, parted, output_split_sizes=list(recv_counts),
                                      input_split_sizes=list(send_counts), group=group, async_op=async_op)
        return out, work
    s0, r0 = _starts(send_counts), _starts(recv_counts)
    works = []
    for r in range(rounds):
        ins = [parted[s0[g] + r * c // rounds: s0[g] + (r + 1) * c // rounds] for g, c in enumerate(send_counts)]
        outs = [out[r0[g] + r * c // rounds: r0[g] + (r + 1) * c // rounds] for g, c in enumerate(recv_counts)]
        works.append(_list_all_to_all(outs, ins, group, async_op))
    return out, (_Works(works) if async_op and works[0] is not None else None)


def exchange_rows(parted, send_counts, group=None):
    """parted: [n,2] int64 rows already grouped by owner (owner-major); send_counts[g] rows go to
    rank g.  Returns ([m,2] rows received, recv_counts).  Works on CPU tensors (gloo) and device
    tensors (RCCL)."""
    world = dist.get_world_size(group)
    assert len(send_counts) == world and sum(send_counts) == parted.shape[0]
    if parted.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal only (several ranks sharing one GPU, where RCCL refuses duplicate devices):
        # stage through host memory.  The production path is RCCL on device tensors, below.
        rows, rc = exchange_rows(parted.cpu(), send_counts, group)
        return rows.to(parted.device), rc
    recv_counts, biggest = _exchange_counts(parted, send_counts, group)
    out, _ = _exchange_data(parted, send_counts, recv_counts, biggest, group, async_op=False)
    return out, recv_counts


def exchange_rows_async(parted, send_counts, group=None):
    """Like exchange_rows but returns (rows, work) right after queueing the data exchange: the caller
    overlaps other GPU work and calls work.wait() (a stream wait, not a host wait) before using rows.
    The small counts exchange stays synchronous (the receive size must be known to allocate)."""
    world = dist.get_world_size(group)
    assert len(send_counts) == world and sum(send_counts) == parted.shape[0]
    if dist.get_backend(group) != "nccl" or not parted.is_cuda:
        rows, _ = exchange_rows(parted, send_counts, group)
        return rows, None
    recv_counts, biggest = _exchange_counts(parted, send_counts, group)
    return _exchange_data(parted, send_counts, recv_counts, biggest, group, async_op=True)


def allreduce_checks(local, device, group=None):
    """local: dict n_matches/sum_r/sum_s/xor_fold/mix_sum (python ints mod 2^64) -> global dict.
    Sums wrap mod 2^64 (two's complement int64 add); the xor is folded after an all_gather."""
    world = dist.get_world_size(group)

    def s64(x):
        x &= (1 << 64) - 1
        return x - (1 << 64) if x >= (1 << 63) else x

    keys = ["n_matches", "sum_r", "sum_s", "mix_sum"]
    # split each 64-bit value in 32-bit halves so the reduction cannot overflow a signed add
    halves = []
    for k in keys:
        v = local[k] & ((1 << 64) - 1)
        halves += [v & 0xFFFFFFFF, v >> 32]
    if dist.get_backend(group) == "gloo":
        device = torch.device("cpu")
    t = torch.tensor(halves, dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    h = t.tolist()
    out = {}
    for i, k in enumerate(keys):
        out[k] = (h[2 * i] + (h[2 * i + 1] << 32)) & ((1 << 64) - 1)
    x = torch.tensor([s64(local["xor_fold"])], dtype=torch.int64, device=device)
    xs = [torch.empty_like(x) for _ in range(world)]
    dist.all_gather(xs, x, group=group)
    acc = 0
    for v in xs:
        acc ^= int(v.item()) & ((1 << 64) - 1)
    out["xor_fold"] = acc
    return out


def pipelined_exchange_join(ex, r_shard, s_shard, b, flags=0, group=None):
    """Owner split + exchange + local join with the exchange hidden behind GPU work:
         split R | exchange R  || split S | exchange S || partition received R | ... probe
    (|| = runs concurrently: RCCL on its stream, the HIP kernels on the current stream).
    The caller has set ex.set_key_prefix_bits(b)."""
    parted_r, off_r = ex.partition_device(r_shard, 64 - b, b)
    rows_r, work_r = exchange_rows_async(parted_r, split_counts_from_offsets(off_r), group)
    parted_s, off_s = ex.partition_device(s_shard, 64 - b, b)   # overlaps the exchange of R
    rows_s, work_s = exchange_rows_async(parted_s, split_counts_from_offsets(off_s), group)
    if work_r is not None:
        work_r.wait()
    if flags == 0:
        ex.prepare_build(rows_r, rows_s.shape[0])               # overlaps the exchange of S
    if work_s is not None:
        work_s.wait()
    return ex.join_device(rows_r, rows_s, flags)


def distributed_join(ex, r_shard, s_shard, flags=0, group=None):
    """Each rank holds a row shard of R (build) and S (probe) on its GPU.  Returns
    (local JoinResult for this rank's key range, global checks dict)."""
    world = dist.get_world_size(group)
    if world == 1:
        res = ex.join_device(r_shard, s_shard, flags)
        return res, res.checks()
    b = owner_bits(world)
    ex.set_key_prefix_bits(b)  # every row received here carries this rank's owner bits on top
    try:
        res = pipelined_exchange_join(ex, r_shard, s_shard, b, flags, group)
    finally:
        ex.set_key_prefix_bits(-1)
    return res, allreduce_checks(res.checks(), r_shard.device, group)
