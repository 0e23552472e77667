"""N>1 path with the real kernels: `world` processes share the one GPU of the box (gloo carries the exchange,
as RCCL refuses two ranks on one device) and run hashmergejoin_amd.dist.distributed_join -- owner split with
the HIP radix pass, exchange, prepared build side / key-prefix plan, local join -- on row shards of the same
relations.  Checked against the CPU oracle: all-reduced checksums, and the per-rank ordered rows concatenated
in rank order (rank g owns key range g)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

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
dist.init_process_group("gloo", rank=rank, world_size=world)
o = Oracle()
nb, npb, miss, dup = int(os.environ["NB"]), int(os.environ["NP"]), int(os.environ["MISS"]), int(os.environ["DUP"])
if os.environ.get("MAXMSG"):
    hdist.MAX_MSG_BYTES = int(os.environ["MAXMSG"])
b0, b1 = rank * nb // world, (rank + 1) * nb // world
p0, p1 = rank * npb // world, (rank + 1) * npb // world
Bs = o.gen_build(b1 - b0, start=b0)
if dup == 1:  # duplicate build keys across shards: global row i and i + nb/2 share a key
    Bs[:, 0] = o.gen_build(b1 - b0, start=b0 % (nb // 2))[:, 0] if b0 >= nb // 2 else Bs[:, 0]
Ps = o.gen_probe(p1 - p0, nb // 2 if dup == 1 else nb, start=p0, miss_mod=miss)
if dup == 2:  # dense integer keys: every row's owner bits are 0 -> rank 0 receives everything, the others nothing
    Bs[:, 0] = np.arange(b0, b1, dtype=np.uint64)
    Ps[:, 0] = (np.arange(p0, p1, dtype=np.uint64) * np.uint64(7)) % np.uint64(nb + nb // 4)
to_dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64).copy()).cuda()
ex = H.Executor(0)
bd, pd = to_dev(Bs), to_dev(Ps)
out = {}
for name, fl in [("count", 0), ("checksum", H.HMJ_CHECKSUM), ("first", H.HMJ_FIRST_WINS | H.HMJ_CHECKSUM | H.HMJ_SUM_PROBE),
                 ("ordered", H.HMJ_ORDERED | H.HMJ_CHECKSUM), ("count_again", 0)]:
    res, glob = hdist.distributed_join(ex, bd, pd, fl)
    out[name] = glob
    if fl & H.HMJ_ORDERED:
        np.save(os.path.join(os.environ["OUT"], "rows%d.npy" % rank), ex.columns_to_numpy(res, host=False))
    ex.release_result()
if rank == 0:
    json.dump(out, open(os.path.join(os.environ["OUT"], "glob.json"), "w"))
dist.destroy_process_group()
'''


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,nb,npb,miss,dup,maxmsg", [(2, 300000, 200000, 3, 0, 0), (4, 1 << 20, (1 << 20) + 777, 0, 0, 0),
                                                            (2, 1 << 21, 1 << 22, 5, 1, 1 << 20), (2, (1 << 23) + 10, 1 << 23, 0, 0, 0),
                                                            (2, 300000, 250000, 0, 2, 0)])
def test_distributed_join_on_one_gpu(oracle, tmp_path, world, nb, npb, miss, dup, maxmsg):
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HMJ_ROOT=ROOT, NB=str(nb), NP=str(npb), MISS=str(miss), DUP=str(dup), OUT=str(tmp_path),
                   MAXMSG=str(maxmsg) if maxmsg else "", OMP_NUM_THREADS="1", HMJ_SLAB_MIN_LOG2="22")
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=500)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    glob = json.load(open(tmp_path / "glob.json"))
    B = oracle.gen_build(nb)
    if dup == 1:
        B[nb // 2:, 0] = B[: nb - nb // 2, 0]
    P = oracle.gen_probe(npb, nb // 2 if dup == 1 else nb, miss_mod=miss)
    if dup == 2:
        B[:, 0] = np.arange(nb, dtype=np.uint64)
        P[:, 0] = (np.arange(npb, dtype=np.uint64) * np.uint64(7)) % np.uint64(nb + nb // 4)
    ck, rows = oracle.equijoin(B, P)
    ckf, _ = oracle.equijoin(B, P, first_wins=True, cap=0)
    assert glob["count"] == glob["count_again"]
    for k in ("n_matches", "sum_r", "sum_s"):
        assert glob["count"][k] == ck[k]
    assert glob["checksum"] == ck and glob["ordered"] == ck
    assert glob["first"] == ckf  # first build row in GLOBAL input order: shards arrive in rank order
    cat = np.concatenate([np.load(tmp_path / ("rows%d.npy" % r)) for r in range(world)])
    assert np.array_equal(cat, rows)
