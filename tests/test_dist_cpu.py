"""N>1 path on CPU: gloo ranks drive the product's exchange protocol without a GPU.

What runs here is the part of the N>1 path that is host arithmetic and host plumbing:
  * the round plan and receive layouts of the exchange -- hmj_exchange_rounds / hmj_exchange_layout, exported by
    libhmj_hip.so and used unchanged by hmj_exchange_join_u64_device (csrc/exchange.hip);
  * the callback transport (hashmergejoin_amd.dist.GroupTransport, the hmj_transport a host hands to
    hmj_comm_set_transport) moving real bytes between processes over gloo, round by round, on host pointers;
  * the owner function's numpy mirror (dist.owner_of; the HIP owner_digit is checked against it on the GPU).
The HIP steps either side (owner split kernel, local join) cannot run without a GPU, so HERE the oracle stands
in for them as the checker.  The same protocol with the real kernels: tests/test_dist_gpu.py."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import ctypes as C, json, os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["HMJ_ROOT"])
from hashmergejoin_amd import dist as hdist
from oracle.pyoracle import Oracle

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
o = Oracle()
nb, npb, miss = int(os.environ["NB"]), int(os.environ["NP"]), int(os.environ["MISS"])
max_rows = int(os.environ["MAXROWS"])
dense = os.environ.get("DENSE") == "1"
ordered = os.environ.get("ORDERED") == "1"
# this rank's row shards of the global relations (rows [rank*n/world, (rank+1)*n/world))
b0, b1 = rank * nb // world, (rank + 1) * nb // world
p0, p1 = rank * npb // world, (rank + 1) * npb // world
Bs = o.gen_build(b1 - b0, start=b0)
Ps = o.gen_probe(p1 - p0, nb, start=p0, miss_mod=miss)
if dense:  # dense integer keys: their top bits are all zero -- the hash owner must still spread them
    Bs[:, 0] = np.arange(b0, b1, dtype=np.uint64)
    Ps[:, 0] = (np.arange(p0, p1, dtype=np.uint64) * np.uint64(7)) % np.uint64(nb + nb // 4)
tr = hdist.GroupTransport(None, device=False)
U64P = C.POINTER(C.c_uint64)
splitters = None
if ordered:  # any splitters all ranks agree on give key ranges; take quantiles of rank 0's build keys
    s = np.sort(o.gen_build(min(nb, 4096))[:, 0])
    splitters = np.array([s[len(s) * (i + 1) // world] for i in range(world - 1)], dtype=np.uint64)

def split(rel):  # stand-in for the HIP owner split: stable partition by owner
    own = hdist.owner_of(rel[:, 0], world, splitters)
    order = np.argsort(own, kind="stable")
    return np.ascontiguousarray(rel[order]), np.bincount(own, minlength=world).astype(np.uint64)

pr, cr = split(Bs)
ps, cs = split(Ps)
send = np.concatenate([cr, cs]).astype(np.uint64)
allc = np.zeros(2 * world * world, np.uint64)
assert tr.struct.allgather_u64(None, send.ctypes.data_as(U64P), allc.ctypes.data_as(U64P), 2 * world) == 0
allc = allc.reshape(world, 2 * world)
MR, MS = np.ascontiguousarray(allc[:, :world]), np.ascontiguousarray(allc[:, world:])

def exchange(parted, M, layout):
    plan = hdist.exchange_plan(M, rank, max_rows, layout)
    total = int(M[:, rank].sum())
    out = np.zeros((total, 2), np.uint64)
    R = plan["n_rounds"]
    assert int(plan["send_rows"].max()) <= max_rows and int(plan["recv_rows"].max()) <= max_rows
    assert np.array_equal(plan["send_rows"].sum(0), M[rank]) and np.array_equal(plan["recv_rows"].sum(0), M[:, rank])
    for r in range(R):
        sp = (C.c_void_p * world)(*[parted.ctypes.data + 16 * int(plan["send_off"][r, g]) for g in range(world)])
        rp = (C.c_void_p * world)(*[out.ctypes.data + 16 * int(plan["recv_off"][r, g]) for g in range(world)])
        sb = (C.c_uint64 * world)(*[16 * int(x) for x in plan["send_rows"][r]])
        rb = (C.c_uint64 * world)(*[16 * int(x) for x in plan["recv_rows"][r]])
        assert tr.struct.alltoallv(None, r, sp, sb, rp, rb, None) == 0
        if layout == 1:  # round-major: rows [0, round_end[r]) are complete after round r
            done = int(plan["round_end"][r])
            assert bool(np.all(hdist.owner_of(out[:done, 0], world, splitters) == rank))
    assert int(plan["round_end"][-1]) == total
    return out, R

recv_r, rounds_r = exchange(pr, MR, 0)   # build side: source-major
recv_s, rounds_s = exchange(ps, MS, 1)   # probe side: round-major
for rows in (recv_r, recv_s):  # every received row belongs to this rank
    assert bool(np.all(hdist.owner_of(rows[:, 0], world, splitters) == rank))
# source-major: sources in rank order, each source's rows in its input order = global input order
if len(recv_r) and not dense:
    assert bool(np.all(np.diff(recv_r[:, 1].astype(np.int64)) > 0))  # build payload = global row index
ck, rows = o.equijoin(recv_r, recv_s)                              # stand-in for the local HIP join
mine = np.array([ck[k] for k in ("n_matches", "sum_r", "sum_s", "xor_fold", "mix_sum")], dtype=np.uint64)
allv = np.zeros(5 * world, np.uint64)
assert tr.struct.allgather_u64(None, mine.ctypes.data_as(U64P), allv.ctypes.data_as(U64P), 5) == 0
allv = allv.reshape(world, 5)
M = (1 << 64) - 1
glob = {"n_matches": sum(int(x) for x in allv[:, 0]) & M, "sum_r": sum(int(x) for x in allv[:, 1]) & M,
        "sum_s": sum(int(x) for x in allv[:, 2]) & M, "xor_fold": int(np.bitwise_xor.reduce(allv[:, 3])),
        "mix_sum": sum(int(x) for x in allv[:, 4]) & M}
np.save(os.path.join(os.environ["OUT"], "rows%d.npy" % rank), rows)
json.dump({"glob": glob, "recv": [len(recv_r), len(recv_s)], "rounds": [rounds_r, rounds_s]},
          open(os.path.join(os.environ["OUT"], "info%d.json" % rank), "w"))
dist.destroy_process_group()
'''


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,nb,npb,miss,maxrows,dense,ordered", [
    (2, 5000, 7000, 3, 1 << 30, 0, 0), (4, 1 << 14, 1 << 14, 0, 1 << 30, 0, 0), (2, 5000, 7000, 3, 256, 0, 0),
    (4, 1 << 14, 3000, 2, 100, 0, 0), (4, 20000, 15000, 0, 1000, 1, 0), (3, 9000, 9000, 4, 700, 0, 0),
    (4, 1 << 14, 1 << 14, 3, 1500, 0, 1)])
def test_exchange_protocol_over_gloo(oracle, tmp_path, world, nb, npb, miss, maxrows, dense, ordered):
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HMJ_ROOT=ROOT, NB=str(nb), NP=str(npb), MISS=str(miss), OUT=str(tmp_path), OMP_NUM_THREADS="1",
                   MAXROWS=str(maxrows), DENSE=str(dense), ORDERED=str(ordered))
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    infos = [json.load(open(tmp_path / ("info%d.json" % r))) for r in range(world)]
    B, P = oracle.gen_build(nb), oracle.gen_probe(npb, nb, miss_mod=miss)
    if dense:
        B[:, 0] = np.arange(nb, dtype=np.uint64)
        P[:, 0] = (np.arange(npb, dtype=np.uint64) * np.uint64(7)) % np.uint64(nb + nb // 4)
    ck, rows = oracle.equijoin(B, P)
    assert all(i["glob"] == ck for i in infos)
    assert sum(i["recv"][0] for i in infos) == nb and sum(i["recv"][1] for i in infos) == npb
    if maxrows < 1000:
        assert all(i["rounds"][0] > 1 for i in infos)  # the multi-round exchange really ran
    if not ordered:  # the hash owner spreads ANY key set: dense integer keys too (the top-bits owner sent them all to rank 0)
        for i in infos:
            assert 0.8 * nb / world <= i["recv"][0] <= 1.2 * nb / world, infos
    per_rank = [np.load(tmp_path / ("rows%d.npy" % r)) for r in range(world)]
    if ordered:  # rank g holds the g-th key range: per-rank ordered results concatenate to the global order
        assert np.array_equal(np.concatenate(per_rank), rows)
    else:
        cat = np.concatenate(per_rank)
        order = np.lexsort((cat[:, 2], cat[:, 1], cat[:, 0]))
        assert np.array_equal(cat[order], rows)


def test_round_plan_properties():
    # hmj_exchange_rounds / hmj_exchange_layout: pure host arithmetic inside the product library
    from hashmergejoin_amd import dist as hdist

    rng = np.random.default_rng(5)
    for G in (1, 2, 3, 8):
        M = rng.integers(0, 5000, size=(G, G)).astype(np.uint64)
        M[rng.integers(0, G), rng.integers(0, G)] = 0
        for max_rows in (1 << 30, 1000, 37):
            plans = [[hdist.exchange_plan(M, r, max_rows, lay) for r in range(G)] for lay in (0, 1)]
            for lay in (0, 1):
                R = plans[lay][0]["n_rounds"]
                assert R == max(1, -(-int(M.max()) // max_rows))
                for r in range(G):
                    p = plans[lay][r]
                    assert p["n_rounds"] == R  # every rank computes the same number of rounds
                    assert int(p["send_rows"].max()) <= max_rows
                    # what rank r sends to g in round q is what g expects from r in round q
                    for g in range(G):
                        assert np.array_equal(p["send_rows"][:, g], plans[lay][g]["recv_rows"][:, r])
                    # send slices tile the owner-major split buffer, receive slices tile the receive buffer
                    segs = sorted((int(o), int(n)) for o, n in zip(p["recv_off"].ravel(), p["recv_rows"].ravel()) if n)
                    pos = 0
                    for o, n in segs:
                        assert o == pos
                        pos += n
                    assert pos == int(M[:, r].sum()) == int(p["round_end"][-1])
                    if lay == 1:
                        assert bool(np.all(np.diff(p["round_end"].astype(np.int64)) >= 0))


def test_owner_function_mirror():
    from hashmergejoin_amd import dist as hdist

    keys = np.array([0, 1, 2, 12345, (1 << 63), (1 << 64) - 1], dtype=np.uint64)
    for G in (1, 2, 3, 8, 16):
        own = hdist.owner_of(keys, G)
        for k, g in zip(keys, own):  # exact: floor(mix64(key) * G / 2^64) in big-integer arithmetic
            m = int(hdist.mix64(np.array([k], np.uint64))[0])
            assert int(g) == (m * G) >> 64
    spl = np.array([10, 20, 20, 1 << 40], dtype=np.uint64)
    assert hdist.owner_of(np.array([0, 9, 10, 19, 20, 21, (1 << 40) - 1, 1 << 40], np.uint64), 5, spl).tolist() == [0, 0, 1, 1, 3, 3, 3, 4]
