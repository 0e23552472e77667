#!/usr/bin/env python3
"""Third cliff sweep: every flag combination on unique keys, on a foreign-key shape and on duplicate build keys."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hashmergejoin_amd as H

ex = H.Executor(0)
dev = torch.device("cuda", 0)
FL = [(0, "count"), (H.HMJ_CHECKSUM, "cks"), (H.HMJ_SUM_PROBE, "sump"), (H.HMJ_FIRST_WINS, "first"),
      (H.HMJ_FIRST_WINS | H.HMJ_SUM_PROBE, "first+sump"), (H.HMJ_MATERIALIZE, "mat"), (H.HMJ_MATERIALIZE | H.HMJ_CHECKSUM, "mat+cks"),
      (H.HMJ_ORDERED, "ord"), (H.HMJ_ORDERED | H.HMJ_CHECKSUM, "ord+cks"), (H.HMJ_FIRST_WINS | H.HMJ_MATERIALIZE, "first+mat"),
      (H.HMJ_FIRST_WINS | H.HMJ_ORDERED, "first+ord")]


def run(tag, R, S):
    row = []
    for fl, name in FL:
        ex.set_profiling(False)
        for _ in range(3):
            ex.join_device(R, S, fl)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            ex.join_device(R, S, fl)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / 5 * 1e3
        ex.set_profiling(True)
        ex.join_device(R, S, fl)
        t = ex.last_timing()
        row.append("%s %.2f %#x" % (name, wall, t["path"]))
    print("%-22s | %s" % (tag, " | ".join(row)), flush=True)
    ex.release_result()


n = (1 << 25) + 5
run("unique 2^25", ex.gen_build(n), ex.gen_probe(n, n))
run("unique 2^25 half miss", ex.gen_build(n), ex.gen_probe(n, n, miss_mod=2))
run("fk 2^21 x 2^25", ex.gen_build(1 << 21), ex.gen_uniform_domain(n, 1 << 21))
g = torch.Generator(device=dev)
g.manual_seed(6)
m = 1 << 24
ar = torch.arange(m, device=dev, dtype=torch.int64)
kb = torch.randint(0, m // 2, (m,), device=dev, generator=g, dtype=torch.int64) * 0x9E3779B97F4A7C15 % (1 << 62)
kp = torch.randint(0, m // 2, (m,), device=dev, generator=g, dtype=torch.int64) * 0x9E3779B97F4A7C15 % (1 << 62)
run("dups x2 2^24", torch.stack([kb, ar], 1).contiguous(), torch.stack([kp, ar], 1).contiguous())
