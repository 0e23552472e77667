#!/usr/bin/env python3
"""CPU-only: run every function of the C restatement (oracle/hmj_oracle.c) built with
-fsanitize=address,undefined.  Usage: tools/oracle_asan.sh"""
import os
import sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.pyoracle import Oracle
o = Oracle('/tmp/libhmj_oracle_asan.so')
rng = np.random.default_rng(1)
for nb, npb, miss in [(0, 0, 0), (1, 1, 0), (5, 7, 2), (1000, 1200, 3), (50000, 40000, 5)]:
    B, P = o.gen_build(nb), o.gen_probe(npb, max(nb, 1), miss_mod=miss)
    o.equijoin(B, P); o.equijoin(B, P, first_wins=True); o.hashmergejoin(B, P, 1); o.hashmergejoin(B, P, 3)
    rh = np.stack([B[:, 0], B[:, 0], B[:, 1]], 1); sh = np.stack([P[:, 0], P[:, 0], P[:, 1]], 1)
    o.hashmergejoin2(rh, sh)
    for bits in (1, 4, 10): o.partitioned_join_sum(P, B, bits); o.partition_sizes(B, bits); o.partitioned_table_sizes(B, bits)
    o.radix_int_non_inplace(B, 3, -1); o.radix_int_inplace_t1(B); o.radix_non_inplace_par(B, 2, -1)
    if nb: o.stable_partition(B, 56, 8, 3); o.radix_inplace_seq(rh); o.radix_inplace_par_t1(rh)
dup = np.stack([rng.integers(0, 50, 3000, dtype=np.uint64), np.arange(3000, dtype=np.uint64)], 1)
o.equijoin(dup, dup[::-1].copy()); o.hashmergejoin(dup, dup[::-1].copy(), 2)
print("oracle under ASan/UBSan: clean")
