"""N>1 path with the real kernels, through the C ABI (hmj_exchange_join_u64_device, csrc/exchange.hip).

  * `world` processes share the one GPU of the box; the library's callback transport carries the exchange over
    gloo (RCCL refuses two ranks on one device).  The first radix pass as the owner (digit pre-pass, digit-range
    plan, rounds of digit ranges, per-round joins), the hash / key-range owner split paths, counts, layouts and the
    collective error handling are the production code; only the byte transport differs.
  * one rank with the RCCL transport (self send/recv inside ncclGroupStart/End), up to BASELINE configs[3]'s
    per-rank shape: a 2^28-row shard in 16 rounds.
  * two ranks over RCCL on two GPUs where the box has them (skipped on a one-GPU box).
Checked against the CPU oracle, or at full size against the generator's closed forms."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
M64 = (1 << 64) - 1
VAL_XOR = 0x9E3779B97F4A7C15

WORKER = r'''
import os, sys, json
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["HMJ_ROOT"])
import hashmergejoin_amd as H
from hashmergejoin_amd import dist as hdist
from oracle.pyoracle import Oracle

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
backend = os.environ.get("BACKEND", "gloo")
dev = rank if backend == "nccl" else 0
torch.cuda.set_device(dev)
if backend == "nccl":
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
else:
    dist.init_process_group("gloo", rank=rank, world_size=world)
o = Oracle()
nb, npb, miss, dup = int(os.environ["NB"]), int(os.environ["NP"]), int(os.environ["MISS"]), int(os.environ["DUP"])
b0, b1 = rank * nb // world, (rank + 1) * nb // world
p0, p1 = rank * npb // world, (rank + 1) * npb // world
Bs = o.gen_build(b1 - b0, start=b0)
if dup == 1:  # duplicate build keys across shards: global row i and i + nb/2 share a key
    Bs[:, 0] = o.gen_build(b1 - b0, start=b0 % (nb // 2))[:, 0] if b0 >= nb // 2 else Bs[:, 0]
Ps = o.gen_probe(p1 - p0, nb // 2 if dup == 1 else nb, start=p0, miss_mod=miss)
if dup == 2:  # dense integer keys: all top bits zero; the digit window must sit under them
    Bs[:, 0] = np.arange(b0, b1, dtype=np.uint64)
    Ps[:, 0] = (np.arange(p0, p1, dtype=np.uint64) * np.uint64(7)) % np.uint64(nb + nb // 4)
if dup == 3:  # two far-apart clusters of keys (3/4 and 1/4 of the rows): no contiguous digit ranges balance the ranks -> hash owner
    clus = lambda x: x | (((x & np.uint64(3)) == np.uint64(3)).astype(np.uint64) << np.uint64(62))
    Bs[:, 0] = clus(np.arange(b0, b1, dtype=np.uint64))
    Ps[:, 0] = clus((np.arange(p0, p1, dtype=np.uint64) * np.uint64(5)) % np.uint64(nb))
to_dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64).copy()).cuda()
ex = H.Executor(dev)
kind = hdist.init_comm(ex)
assert kind == ("rccl" if backend == "nccl" else "group")
if os.environ.get("MAXMSG"):
    ex.comm_set_message_bytes(int(os.environ["MAXMSG"]), int(os.environ["MAXMSG"]) // 4)
if os.environ.get("OWNER_SPLIT") == "1":
    ex.comm_set_owner_path(split=True)
bd, pd = to_dev(Bs), to_dev(Ps)
out, infos = {}, {}
ex.set_profiling(True)
for name, fl in [("count", 0), ("checksum", H.HMJ_CHECKSUM), ("first", H.HMJ_FIRST_WINS | H.HMJ_CHECKSUM | H.HMJ_SUM_PROBE),
                 ("ordered", H.HMJ_ORDERED | H.HMJ_CHECKSUM), ("count_again", 0)]:
    res, glob = hdist.distributed_join(ex, bd, pd, fl)
    out[name] = glob
    infos[name] = dict(ex.last_exchange_info(), path=ex.last_timing()["path"])
    if fl & H.HMJ_ORDERED:
        np.save(os.path.join(os.environ["OUT"], "rows%d.npy" % rank), ex.columns_to_numpy(res, host=False))
    ex.release_result()
json.dump({"glob": out, "info": infos}, open(os.path.join(os.environ["OUT"], "out%d.json" % rank), "w"))
ex.close()
dist.destroy_process_group()
'''


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_world(tmp_path, world, nb, npb, miss, dup, maxmsg, backend="gloo", owner_split=False):
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HMJ_ROOT=ROOT, NB=str(nb), NP=str(npb), MISS=str(miss), DUP=str(dup), OUT=str(tmp_path),
                   MAXMSG=str(maxmsg) if maxmsg else "", OMP_NUM_THREADS="1", HMJ_SLAB_MIN_LOG2="22", BACKEND=backend,
                   OWNER_SPLIT="1" if owner_split else "0")
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=500)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    return [json.load(open(tmp_path / ("out%d.json" % r))) for r in range(world)]


def check_world(oracle, tmp_path, res, world, nb, npb, miss, dup, maxmsg, owner_split=False):
    B = oracle.gen_build(nb)
    if dup == 1:
        B[nb // 2:, 0] = B[: nb - nb // 2, 0]
    P = oracle.gen_probe(npb, nb // 2 if dup == 1 else nb, miss_mod=miss)
    if dup == 2:
        B[:, 0] = np.arange(nb, dtype=np.uint64)
        P[:, 0] = (np.arange(npb, dtype=np.uint64) * np.uint64(7)) % np.uint64(nb + nb // 4)
    if dup == 3:
        clus = lambda x: x | (((x & np.uint64(3)) == np.uint64(3)).astype(np.uint64) << np.uint64(62))
        B[:, 0] = clus(np.arange(nb, dtype=np.uint64))
        P[:, 0] = clus((np.arange(npb, dtype=np.uint64) * np.uint64(5)) % np.uint64(nb))
    ck, rows = oracle.equijoin(B, P)
    ckf, _ = oracle.equijoin(B, P, first_wins=True, cap=0)
    for o in res:  # every rank reports the same global reduction
        glob = o["glob"]
        assert glob["count"] == glob["count_again"]
        for k in ("n_matches", "sum_r", "sum_s"):
            assert glob["count"][k] == ck[k]
        assert glob["checksum"] == ck and glob["ordered"] == ck
        assert glob["first"] == ckf  # first build row in GLOBAL input order: sources arrive in rank order
    # ordered mode: rank g owns the g-th key range, so the per-rank ordered rows concatenate to the global order
    cat = np.concatenate([np.load(tmp_path / ("rows%d.npy" % r)) for r in range(world)])
    assert np.array_equal(cat, rows)
    # ownership: every rank owns about 1/world of the rows -- for ANY keys, dense integers included.  Count modes:
    # ranges of the first radix pass's digit (owner_mode 3), no separate owner split; keys in a few clusters fall
    # back to the hash owner (1); ordered: key ranges between sample quantiles (2)
    for o in res:
        i = o["info"]["count"]
        assert o["info"]["ordered"]["owner_mode"] == 2 and i["n_ranks"] == world
        if owner_split:  # the caller asked for round 2's path (hmj_comm_set_owner_path): hash owner, not a fallback
            assert i["owner_mode"] == 1 and i["fallback"] == (1 if dup == 3 else 0) and i["n_subjoins"] == 1, i
            for name in ("checksum", "first"):
                assert o["info"][name]["owner_mode"] == 1
        elif dup == 3:
            assert i["owner_mode"] == 1 and i["fallback"] == 1 and i["sample_max_share"] > 1.3, i
        else:
            assert i["owner_mode"] == 3 and i["fallback"] == 0 and i["digit_bits"] == 8, i
            assert i["n_subjoins"] >= 1 and i["rounds_build"] == i["rounds_probe"], i
            for name in ("checksum", "first"):
                assert o["info"][name]["owner_mode"] == 3
        assert 0.75 * nb / world <= i["recv_build"] <= 1.25 * nb / world, i
        assert 0.7 * npb / world <= i["recv_probe"] <= 1.3 * npb / world, i
        if maxmsg:
            assert i["rounds_build"] > 1 and i["rounds_probe"] > 1, i
    assert sum(o["info"]["count"]["recv_build"] for o in res) == nb
    return res


@pytest.mark.parametrize("world,nb,npb,miss,dup,maxmsg", [(2, 300000, 200000, 3, 0, 0), (4, 1 << 20, (1 << 20) + 777, 0, 0, 0),
                                                            (2, 1 << 21, 1 << 22, 5, 1, 1 << 20), (2, 9 << 20, (9 << 20) + 10, 0, 0, 1 << 24),
                                                            (2, 300000, 250000, 0, 2, 0), (3, 700000, 500000, 2, 0, 1 << 19),
                                                            (2, 300000, 270000, 0, 3, 0), (4, 1 << 20, 1 << 21, 3, 3, 1 << 20),
                                                            (5, 900000, 700001, 4, 0, 1 << 19),  # (five ranks + this process: the box allows six on its GPU)
                                                            (2, 34 << 19, 36 << 20, 0, 0, 0)])
def test_distributed_join_on_one_gpu(oracle, tmp_path, world, nb, npb, miss, dup, maxmsg):
    res = run_world(tmp_path, world, nb, npb, miss, dup, maxmsg)
    check_world(oracle, tmp_path, res, world, nb, npb, miss, dup, maxmsg)
    if nb >= 34 << 19:  # one round of 8.9 M build and 18.9 M probe rows per rank: its join takes the slab path (threshold lowered to 2^22 for the tests)
        import hashmergejoin_amd as H

        for o in res:
            i = o["info"]["count"]
            assert i["path"] & H.HMJ_PATH_SLAB and i["n_subjoins"] == i["rounds_probe"] >= 1, i


@pytest.mark.parametrize("world,nb,npb,miss,dup,maxmsg", [(3, 300000, 200000, 3, 0, 0), (2, 1 << 20, (1 << 20) + 777, 0, 1, 1 << 20)])  # (dup = 1 shares keys across the two halves: 2 or 4 ranks)
def test_owner_split_path_stays_selectable(oracle, tmp_path, world, nb, npb, miss, dup, maxmsg):
    # the digit-owner path is the default for non-ordered joins but has never crossed real links; round 2's
    # owner-split path (hash owner, separate split, one local join) must stay selectable -- hmj_comm_set_owner_path /
    # HMJ_EXCHANGE_OWNER=split -- and give the same results (first-wins across shards included).
    res = run_world(tmp_path, world, nb, npb, miss, dup, maxmsg, owner_split=True)
    check_world(oracle, tmp_path, res, world, nb, npb, miss, dup, maxmsg, owner_split=True)


ERR_WORKER = r"""
import os, sys, json
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["HMJ_ROOT"])
import hashmergejoin_amd as H
from hashmergejoin_amd import dist as hdist
from hashmergejoin_amd._lib import HmjError

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
ex = H.Executor(0)
hdist.init_comm(ex)
n = 200000
bd, pd = ex.gen_build(n, start=rank * n), ex.gen_probe(n, world * n, start=rank * n)
codes = []
for step in range(3):
    bad = step == 1 and rank == 1
    try:
        if bad:  # this rank alone passes a shard beyond the 2^32-1 row limit: rejected before any memory is touched
            import ctypes as C
            loc, glob = H._lib.JoinResult(), H._lib.JoinResult()
            rc = ex.L.hmj_exchange_join_u64_device(ex.h, C.c_void_p(bd.data_ptr()), (1 << 32) + 5, C.c_void_p(pd.data_ptr()), n, 0, C.byref(loc), C.byref(glob))
            codes.append(rc)
        else:
            loc, glob = ex.exchange_join(bd, pd, 0)
            assert int(glob.n_matches) == world * n
            codes.append(0)
    except HmjError as e:
        codes.append(e.code)
json.dump(codes, open(os.path.join(os.environ["OUT"], "codes%d.json" % rank), "w"))
ex.close()
dist.destroy_process_group()
"""


def test_an_error_on_one_rank_ends_the_collective_on_all_ranks(tmp_path):
    # early returns decided by one rank alone left its peers blocked in the next collective.  Now a rank's
    # error travels with its message of the next all-gather and every rank returns -- the failing one its own code,
    # the others HMJ_E_PEER -- and the communicator stays usable (the step before and the step after succeed).
    world, port = 3, free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HMJ_ROOT=ROOT,
                   OUT=str(tmp_path), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, "-c", ERR_WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    codes = [json.load(open(tmp_path / ("codes%d.json" % r))) for r in range(world)]
    assert codes[1] == [0, -1, 0], codes           # HMJ_E_ARG on the rank that made the mistake
    assert codes[0] == codes[2] == [0, -7, 0], codes  # HMJ_E_PEER on the others; nobody hangs


def test_two_ranks_over_rccl(oracle, tmp_path):
    # the RCCL transport between two GPUs: owner split, all-gathers, grouped ncclSend / ncclRecv rounds
    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the driver's multi-GPU node); the one-rank RCCL tests below run everywhere")
    world, nb, npb = 2, 1 << 22, (1 << 22) + 12345
    res = run_world(tmp_path, world, nb, npb, 3, 0, 1 << 22, backend="nccl")
    check_world(oracle, tmp_path, res, world, nb, npb, 3, 0, 1 << 22)


def sum_xor_range(n, c):
    """sum over j in [0, n) of (j ^ c) mod 2^64, bit by bit."""
    total = 0
    for b in range(64):
        period, half = 1 << (b + 1), 1 << b
        ones = (n // period) * half + max(0, (n % period) - half)  # j in [0,n) with bit b set
        cb = (c >> b) & 1
        cnt = (n - ones) if cb else ones
        total += cnt << b
    return total & M64


@pytest.fixture(scope="module")
def ex1():
    import hashmergejoin_amd as H
    from hashmergejoin_amd import dist as hdist

    e = H.Executor(0)
    hdist.init_comm_single(e)  # one rank, RCCL transport
    yield e
    e.close()


@pytest.mark.parametrize("log2n,maxmsg", [(22, 1 << 22), (24, 1 << 20), (26, 1 << 28), (28, 1 << 30)])
def test_one_rank_rccl_exchange_at_shard_size(ex1, log2n, maxmsg):
    # BASELINE configs[3]'s per-rank shape (2^28-row shard at log2n = 28): the whole exchange path through RCCL
    # (forced with hmj_comm_set_self_exchange: a one-rank job's default is the plain join) -- digit pre-pass, rounds
    # of digit ranges as self send/recv, one join per arrived round -- checked by the generator's closed forms
    # (every probe row matches one build row).  [24-2^20]: 16 MiB messages under a 1 MiB limit: every message goes as
    # sixteen consecutive ncclSend / ncclRecv pieces (what keeps a message under RCCL's 2 GiB truncation).
    import hashmergejoin_amd as H

    ex = ex1
    n = 1 << log2n
    bd, pd = ex.gen_build(n), ex.gen_probe(n, n)
    ex.comm_set_message_bytes(maxmsg, maxmsg // 8)
    ex.set_profiling(True)
    loc, glob = ex.exchange_join(bd, pd, 0)
    info, t = ex.last_exchange_info(), ex.last_timing()
    for r in (loc, glob):
        assert int(r.n_matches) == n
        assert int(r.sum_r) == (n * (n - 1) // 2) & M64
        assert int(r.sum_s) == sum_xor_range(n, VAL_XOR)
    assert info["n_ranks"] == 1 and info["recv_build"] == n and info["recv_probe"] == n
    assert info["owner_mode"] == 3 and info["digit_bits"] == 8 and info["digit_low"] == 56, info
    if log2n >= 26:
        assert info["rounds_build"] == info["rounds_probe"] == info["n_subjoins"] >= 8, info
        assert t["path"] & (H.HMJ_PATH_SLAB | H.HMJ_PATH_EXACT), t
        assert info["ms_kernels"] > 0 and info["ms_exposed"] >= 0, info
    ck = ex.exchange_join(bd, pd, H.HMJ_CHECKSUM)[1].checks()
    plain = ex.join_device(bd, pd, H.HMJ_CHECKSUM).checks()
    assert ck == plain  # same multiset of result rows as the plain single-GPU join
    if log2n <= 26:
        loc, glob = ex.exchange_join(bd, pd, H.HMJ_ORDERED | H.HMJ_CHECKSUM)
        assert glob.checks() == plain and int(loc.n_matches) == n
        import torch

        from hashmergejoin_amd.join import _memcpy_d2d

        k = torch.empty(n, dtype=torch.int64, device="cuda")
        _memcpy_d2d(torch, k, loc.key, n * 8)
        ks = k ^ torch.tensor(-(1 << 63), dtype=torch.int64, device="cuda")
        assert bool((ks[1:] > ks[:-1]).all())
    ex.set_profiling(False)
    ex.release_result()


def test_owner_split_matches_the_numpy_mirror(ex1, oracle):
    # the HIP owner_digit places every row exactly where dist.owner_of says, stably (owner-major, input order
    # within an owner) -- hash owner for any rank count, key-range owner with splitters
    import torch

    from hashmergejoin_amd import dist as hdist

    rng = np.random.default_rng(4)
    for n in (1, 63, 4096, 4097, 300001):
        B = oracle.gen_build(n)
        if n > 1000:
            B[: n // 3, 0] = np.arange(n // 3, dtype=np.uint64)  # dense integers mixed in
        bd = torch.from_numpy(B.view(np.int64).copy()).cuda()
        for G in (2, 3, 8, 16):
            for spl in (None, np.sort(rng.integers(0, 1 << 63, size=G - 1, dtype=np.uint64) * np.uint64(2))):
                own = hdist.owner_of(B[:, 0], G, spl)
                order = np.argsort(own, kind="stable")
                out, off = ex1.owner_split(bd, G, spl)
                assert np.array_equal(out.cpu().numpy().view(np.uint64), B[order]), (n, G, spl is None)
                cnt = np.bincount(own, minlength=G)
                assert np.array_equal(np.diff(off.cpu().numpy())[:G], cnt), (n, G)


def test_bench_launches_its_own_ranks():
    # the driver runs `python3 bench.py --gpus N` without a launcher.  The parent starts N fresh
    # rank processes before anything touches the GPU and relays rank 0's JSON line; on a one-GPU box the ranks share
    # the GPU over the gloo callback transport.  The line must say how many ranks the exchange really saw.
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--log2n", "22", "--steps", "2", "--warmup", "1"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout.decode()
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["rows_per_relation_total"] == 2 << 22
    x = d["exchange"]
    assert x["n_ranks"] == 2 and x["owner"].startswith("ranges of the first radix pass") and x["digit_bits"] == 8
    assert sum(x["recv_rows_rank0"]) > 0 and x["joins_per_step"] >= 1
    import torch

    if torch.cuda.device_count() >= 2:
        assert x["transport"] == "rccl" and x["rccl_ranks"] == 2
    else:
        assert x["transport"] == "callbacks over gloo" and x["rccl_ranks"] == 0
    for k in ("split", "exchange_build", "exchange_probe", "local", "kernels", "exposed", "total"):
        assert k in x, x
    assert 0 <= x["exposed"] <= x["total"] + 1e-3
    assert d["probe_phase"]["probe_tuples_per_s_all_ranks"] > 0 and 0 < d["probe_phase"]["per_gpu"]["frac"] < 1
    assert "roofline" in d and "placement" in d


# ---- a step cannot hang: deadline inside the library, wall-clock limit in bench.py ------------------------
STALL_WORKER = r"""
import os, sys, json, time
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["HMJ_ROOT"])
import hashmergejoin_amd as H
from hashmergejoin_amd import dist as hdist
from hashmergejoin_amd._lib import HmjError

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
ex = H.Executor(0)
hdist.init_comm(ex, timeout_s=float(os.environ["STEP_TIMEOUT_S"]))
assert ex.comm_get_timeout_ms() == int(float(os.environ["STEP_TIMEOUT_S"]) * 1000)
n = 200000
bd, pd = ex.gen_build(n, start=rank * n), ex.gen_probe(n, world * n, start=rank * n)
loc, glob = ex.exchange_join(bd, pd, 0)          # step 0: everybody takes part
assert int(glob.n_matches) == world * n
rec = {"codes": [0], "secs": []}
if rank == 1:                                    # step 1: this rank never enters it
    time.sleep(float(os.environ["STALL_S"]))
    json.dump(rec, open(os.path.join(os.environ["OUT"], "stall%d.json" % rank), "w"))
    os._exit(3)
for _ in range(2):                               # the others: the step times out; the communicator is unusable afterwards
    t0 = time.time()
    try:
        ex.exchange_join(bd, pd, 0)
        rec["codes"].append(0)
    except HmjError as e:
        rec["codes"].append(e.code)
        rec["msg"] = str(e)
    rec["secs"].append(time.time() - t0)
# the local join of the same context still works: only the communicator is gone
rec["local_matches"] = int(ex.join_device(bd, ex.gen_probe(n, n, start=0), 0).n_matches) if rank == 0 else None
json.dump(rec, open(os.path.join(os.environ["OUT"], "stall%d.json" % rank), "w"))
ex.close()                                       # (tears the aborted communicator down without blocking)
os._exit(5)
"""


def test_a_rank_that_never_enters_a_step_times_the_others_out(tmp_path):
    # the reference's workers all return from the call that started them (hashjoin.h:56-68 -> radix_hash.h:375-405); ranks in
    # different processes can lose a peer: every other rank must come back with HMJ_E_TIMEOUT within the deadline, and
    # every process must end (non-zero) instead of waiting in a collective for ever
    import time

    world, port, step_timeout = 3, free_port(), 4.0
    procs, t0 = [], time.time()
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HMJ_ROOT=ROOT,
                   OUT=str(tmp_path), OMP_NUM_THREADS="1", STEP_TIMEOUT_S=str(step_timeout), STALL_S="20")
        procs.append(subprocess.Popen([sys.executable, "-c", STALL_WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert [p.returncode for p in procs] == [5, 3, 5], "\n".join(outs)
    for r in (0, 2):
        rec = json.load(open(tmp_path / ("stall%d.json" % r)))
        assert rec["codes"] == [0, -8, -8], (rec, outs[r])       # HMJ_E_TIMEOUT, then "communicator unusable"
        assert rec["secs"][0] < step_timeout + 6 and rec["secs"][1] < 1.0, rec
        assert "no progress within 4000 ms" in rec["msg"] or "unusable" in rec["msg"] or "aborted" in rec["msg"], rec
    assert json.load(open(tmp_path / "stall0.json"))["local_matches"] == 200000
    assert time.time() - t0 < 120


def test_bench_rehearsal_ends_in_bounded_time_when_a_rank_is_killed_mid_run():
    # `python bench.py --gpus 2` on this one-GPU box (gloo callbacks); rank 1 is killed before its third step: rank 0's step
    # times out inside the library, rank 0 prints the error line, the launcher relays it and returns non-zero
    import time

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["HMJ_BENCH_FAULT"] = "kill:1:2"
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--log2n", "22", "--steps", "3", "--warmup", "1",
                        "--step-timeout-s", "5", "--timeout-s", "200"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=400)
    dt = time.time() - t0
    assert p.returncode != 0 and dt < 150, (p.returncode, dt, p.stderr.decode()[-2000:])
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout.decode()
    d = json.loads(lines[0])
    assert d["value"] is None and d["n_gpus"] == 2 and "error" in d, d
    assert "exchange step" in d["error"] or "rank(s) 1 failed" in d["error"], d


LOST_SEND_WORKER = r"""
import os, sys, json, time
import torch
sys.path.insert(0, os.environ["HMJ_ROOT"])
import hashmergejoin_amd as H
from hashmergejoin_amd import dist as hdist
from hashmergejoin_amd._lib import HmjError

ex = H.Executor(0)
hdist.init_comm_single(ex, timeout_s=3.0)        # one rank, RCCL transport, the whole exchange path
n = 1 << 22
bd, pd = ex.gen_build(n), ex.gen_probe(n, n)
rec = {}
t0 = time.time()
try:
    ex.exchange_join(bd, pd, 0)
    rec["code"] = 0
except HmjError as e:
    rec["code"], rec["msg"] = e.code, str(e)
rec["secs"] = time.time() - t0
t0 = time.time()
rec["local_matches"] = int(ex.join_device(bd, pd, 0).n_matches)   # the GPU and the context are fine
ex.close()
rec["close_secs"] = time.time() - t0
json.dump(rec, open(os.path.join(os.environ["OUT"], "lost.json"), "w"))
"""


def test_a_stalled_rccl_round_ends_in_timeout_and_teardown_does_not_block(tmp_path):
    # RCCL transport on the one GPU of the box.  A peer that stops sending looks, from this rank, like a kernel on the
    # communication stream that does not finish (RCCL's receive spinning on data that does not come).  One rank cannot make
    # RCCL itself wait (a self receive without its send is refused as invalid usage), so the test hook HMJ_FAULT_STALL holds
    # the stream with a kernel that spins for 14 s and exits.  The step must give up at its 3 s deadline with
    # HMJ_E_TIMEOUT, the context's local join must still run, and close() must come back (ncclCommAbort, which waits for the
    # communicator's own kernels -- here queued behind the 14 s stand-in; RCCL's kernels leave at the abort flag), exit code 0.
    env = dict(os.environ, HMJ_ROOT=ROOT, OUT=str(tmp_path), HMJ_FAULT_STALL="1:14000")
    p = subprocess.run(["timeout", "-k", "10", "120", sys.executable, "-c", LOST_SEND_WORKER], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=200)
    assert p.returncode == 0, p.stdout.decode()[-3000:]
    rec = json.load(open(tmp_path / "lost.json"))
    assert rec["code"] == -8 and 2.5 < rec["secs"] < 8, rec
    assert "no progress within 3000 ms" in rec["msg"], rec
    assert rec["local_matches"] == 1 << 22 and rec["close_secs"] < 25, rec
