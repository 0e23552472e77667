#!/usr/bin/env python3
"""Large-scale cross-check against an INDEPENDENT implementation (torch sort / unique / searchsorted on the GPU):
count, sums and first-wins sums of joins far beyond what the CPU oracle checks in seconds, over several key
distributions.  Exercises the default plan at full size: slab partitioning, key windows, split partitions."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hashmergejoin_amd as H

dev = torch.device("cuda:0")
M63 = (1 << 63) - 1
def as_u64_key_order(k):  # int64 view of unsigned keys -> order-preserving signed keys
    return k ^ torch.tensor(-(1 << 63), dtype=torch.int64, device=dev)

def reference(B, P):
    """B, P: int64 [n,2] (key,val as two's complement).  Returns dict of expected values (mod 2^64 as int64)."""
    kb, vb = as_u64_key_order(B[:, 0]), B[:, 1]
    kp, vp = as_u64_key_order(P[:, 0]), P[:, 1]
    sk, order = torch.sort(kb, stable=True)
    sv = vb[order]
    uk, inv, cnt = torch.unique_consecutive(sk, return_inverse=True, return_counts=True)
    sumv = torch.zeros_like(uk).scatter_add_(0, inv, sv)               # wraps mod 2^64
    first_idx = torch.cumsum(cnt, 0) - cnt                              # first row of each key (input order: stable sort)
    firstv = sv[first_idx]
    pos = torch.searchsorted(uk, kp).clamp_(max=uk.numel() - 1)
    hit = uk[pos] == kp
    c = cnt[pos] * hit
    return {
        "n": int(c.sum().item()),
        "sum_r": int((sumv[pos] * hit).sum().item()),
        "sum_s": int((vp * c).sum().item()),
        "fw_n": int(hit.sum().item()),
        "fw_sum_r": int((firstv[pos] * hit).sum().item()),
        "fw_sum_s": int((vp * hit).sum().item()),
        "sum_p": int(vp.sum().item()),
    }

def s64(x):
    x &= (1 << 64) - 1
    return x - (1 << 64) if x >= (1 << 63) else x

def gen(kind, n, g):
    r = lambda hi, size: torch.randint(0, hi, (size,), dtype=torch.int64, device=dev, generator=g)
    if kind == "uniform":
        k = r(M63, n) * 2 + r(2, n)
    elif kind == "dup2":       # every key about twice
        k = (r(n // 2 + 1, n) * 0x9E3779B97F4A7C15) & -1
    elif kind == "dense":
        k = r(n, n)
    elif kind == "tagged":
        k = (r(3, n) << 61) | r(1 << 40, n)
    elif kind == "hot":        # 5 % of the rows share one key, the rest uniform
        k = r(M63, n) * 2
        k[r(n, n // 20)] = 0x1234567890ABCDEF
    elif kind == "sorted":
        k = torch.sort(r(M63, n))[0]
    else:
        raise ValueError(kind)
    return torch.stack([k, r(M63, n)], 1).contiguous()

def main():
    log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 26
    n = 1 << log2n
    ex = H.Executor(0)
    g = torch.Generator(device=dev); g.manual_seed(42)
    bad = 0
    for kb_kind, kp_kind in [("uniform", "uniform"), ("dup2", "dup2"), ("dense", "dense"), ("tagged", "tagged"), ("uniform", "hot"),
                             ("hot", "uniform"), ("sorted", "uniform"), ("fk", "uniform")]:
        B = gen(kb_kind, n if kb_kind != "fk" else n // 8, g) if kb_kind != "fk" else gen("uniform", n // 8, g)
        P = gen(kp_kind, n, g)
        if kb_kind == "fk":  # foreign-key shape: every probe row carries one of the (n/8) build keys
            P[:, 0] = B[torch.randint(0, B.shape[0], (n,), device=dev, generator=g), 0]
        if kp_kind in ("uniform", "hot", "sorted") and kb_kind in ("uniform", "hot", "sorted"):
            # make about half of the probe rows match something
            idx = torch.randint(0, n, (n // 2,), device=dev, generator=g)
            P[: n // 2, 0] = B[idx, 0]
        want = reference(B, P)
        ex.set_profiling(True)
        r = ex.join_device(B, P, 0); t = ex.last_timing()
        ok1 = (int(r.n_matches), s64(int(r.sum_r)), s64(int(r.sum_s))) == (want["n"], want["sum_r"], want["sum_s"])
        f = ex.join_device(B, P, H.HMJ_FIRST_WINS | H.HMJ_SUM_PROBE); tf = ex.last_timing()
        ok2 = (int(f.n_matches), s64(int(f.sum_r)), s64(int(f.sum_s)), s64(int(f.sum_probe_all))) == (want["fw_n"], want["fw_sum_r"], want["fw_sum_s"], want["sum_p"])
        print("2^%d build=%-8s probe=%-8s matches %12d  count %s (%.2f ms, hist %.2f)  first-wins %s (%.2f ms)" % (
            log2n, kb_kind, kp_kind, want["n"], "OK" if ok1 else "WRONG", t["ms_total"], t["ms_hist"], "OK" if ok2 else "WRONG", tf["ms_total"]), flush=True)
        bad += (not ok1) + (not ok2)
        if want["n"] <= (1 << 27) and kb_kind != "hot":
            m = ex.join_device(B, P, H.HMJ_ORDERED | H.HMJ_CHECKSUM); tm = ex.last_timing()
            c = ex.join_device(B, P, H.HMJ_CHECKSUM)
            ok3 = m.checks() == c.checks() and int(m.n_matches) == want["n"]
            import ctypes
            from hashmergejoin_amd.join import _memcpy_d2d
            k = torch.empty(int(m.n_matches), dtype=torch.int64, device=dev)
            if k.numel():
                _memcpy_d2d(torch, k, m.key, k.numel() * 8)
                ks = as_u64_key_order(k)
                ok3 = ok3 and bool((ks[1:] >= ks[:-1]).all())
            print("      ordered: checksums equal count-mode's and keys ascend: %s (%.2f ms)" % ("OK" if ok3 else "WRONG", tm["ms_total"]), flush=True)
            bad += not ok3
            ex.release_result()
        del B, P
    print("ALL OK" if bad == 0 else "%d MISMATCHES" % bad)
    sys.exit(1 if bad else 0)

main()
