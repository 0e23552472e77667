#!/usr/bin/env python3
"""Ordered foreign-key joins under each plan (HMJ_FK_PLAN = auto | wide | half | narrow): 2^b unique build keys x 2^p probe rows.
usage: exp_fk_plans.py p b0 b1"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import hashmergejoin_amd as H

p, b0, b1 = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
gen = H.Executor(0)
rels = {b: (gen.gen_build(1 << b), gen.gen_uniform_domain(1 << p, 1 << b)) for b in range(b0, b1 + 1)} if p <= 26 else None
ref = {}
for plan in ("auto", "wide", "half", "narrow"):  # one executor at a time: four 2^28-row workspaces do not fit side by side
    os.environ["HMJ_FK_PLAN"] = plan
    ex = H.Executor(0)
    ex.set_profiling(True)
    for b in range(b0, b1 + 1):
        R, S = rels[b] if rels else (gen.gen_build(1 << b), gen.gen_uniform_domain(1 << p, 1 << b))
        ts = []
        for i in range(5):
            r = ex.join_device(R, S, H.HMJ_ORDERED | (H.HMJ_CHECKSUM if i == 0 else 0))
            assert int(r.n_matches) == 1 << p
            if i == 0:
                ck = r.checks()
                ref.setdefault(b, ck)
                assert ck == ref[b], (plan, ck, ref[b])
            if i >= 2:
                ts.append(ex.last_timing())
        m = lambda k: sum(t[k] for t in ts) / len(ts)
        print("2^%d x 2^%d ordered %-6s %.2f ms (b%d part %.2f write %.2f%s path %#x)" % (
            b, p, plan, m("ms_total"), ts[-1]["radix_bits"], m("ms_partition_build") + m("ms_partition_probe"), m("ms_probe_write"),
            " order %.2f" % m("ms_order") if m("ms_order") > 0 else "", ts[-1]["path"]), flush=True)
        ex.release_result()
        del R, S
    ex.close()
