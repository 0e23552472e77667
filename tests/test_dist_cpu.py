"""N>1 path on CPU: two gloo ranks run the product's partition-exchange host code
(hashmergejoin_amd/dist.py) on CPU tensors.  The HIP steps either side of the exchange (owner
split, local join) cannot run without a GPU, so HERE the oracle stands in for them as the checker
(tests may call the oracle); what is under test is the exchange: split counts, all_to_all_single
plumbing, receive layout, and the 64-bit modular all-reduce of the result checksums."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
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
# this rank's row shards of the global relations (rows [rank*n/world, (rank+1)*n/world))
b0, b1 = rank * nb // world, (rank + 1) * nb // world
p0, p1 = rank * npb // world, (rank + 1) * npb // world
Bs = o.gen_build(b1 - b0, start=b0)
Ps = o.gen_probe(p1 - p0, nb, start=p0, miss_mod=miss)
bits = hdist.owner_bits(world)
if os.environ.get("MAXMSG"):  # force the multi-round exchange (production: messages of 1 GiB and more)
    hdist.MAX_MSG_BYTES = int(os.environ["MAXMSG"])
recv = []
for rel in (Bs, Ps):
    parted, off = o.stable_partition(rel, 64 - bits, bits)        # stand-in for hmj_partition_u64_device
    counts = hdist.split_counts_from_offsets(torch.from_numpy(off.astype(np.int64)))
    t = torch.from_numpy(parted.view(np.int64).copy())
    rows, rc = hdist.exchange_rows(t, counts)
    rows = rows.numpy().view(np.uint64)
    # every received row belongs to this rank's key range
    assert len(rows) == 0 or bool(np.all((rows[:, 0] >> np.uint64(64 - bits)) == np.uint64(rank)))
    # and the rows arrive grouped by source rank in shard order = global input order
    assert len(rows) == sum(rc)
    if rel is Bs and len(rows):
        assert bool(np.all(np.diff(rows[:, 1].astype(np.int64)) > 0))  # build payload = global row index
    recv.append(rows)
ck, rows = o.equijoin(recv[0], recv[1])                            # stand-in for hmj_join_u64_device
glob = hdist.allreduce_checks(ck, torch.device("cpu"))
np.save(os.path.join(os.environ["OUT"], "rows%d.npy" % rank), rows)
if rank == 0:
    import json
    json.dump(glob, open(os.path.join(os.environ["OUT"], "glob.json"), "w"))
dist.destroy_process_group()
'''


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,nb,npb,miss,maxmsg", [(2, 5000, 7000, 3, 0), (4, 1 << 14, 1 << 14, 0, 0),
                                                        (2, 5000, 7000, 3, 4096), (4, 1 << 14, 3000, 2, 1000)])
def test_exchange_two_ranks_gloo(oracle, tmp_path, world, nb, npb, miss, maxmsg):
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HMJ_ROOT=ROOT, NB=str(nb), NP=str(npb), MISS=str(miss), OUT=str(tmp_path), OMP_NUM_THREADS="1",
                   MAXMSG=str(maxmsg) if maxmsg else "")
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    import json

    glob = json.load(open(tmp_path / "glob.json"))
    B, P = oracle.gen_build(nb), oracle.gen_probe(npb, nb, miss_mod=miss)
    ck, rows = oracle.equijoin(B, P)
    assert glob == ck
    # rank g holds key range g: concatenating the per-rank ordered results gives the global order
    cat = np.concatenate([np.load(tmp_path / ("rows%d.npy" % r)) for r in range(world)])
    assert np.array_equal(cat, rows)


def test_owner_bits_and_counts():
    import torch

    from hashmergejoin_amd import dist as hdist

    assert [hdist.owner_bits(w) for w in (1, 2, 4, 8)] == [0, 1, 2, 3]
    with pytest.raises(ValueError):
        hdist.owner_bits(6)
    assert hdist.split_counts_from_offsets(torch.tensor([0, 3, 3, 10])) == [3, 0, 7]
