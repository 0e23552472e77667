"""Independent full-size check of BASELINE configs[4] (Zipf build side x uniform probe side over one key domain).

Both generators draw a RANK in [0, domain) per row and emit key = mix64(rank + seed) (csrc/gen.hip,
gen_from_cdf_kernel / gen_uniform_domain_kernel); mix64 is a bijection, so joining on the key is joining on the rank.
This module recomputes the ranks with plain torch integer ops (no library call, no join kernel, no oracle) and derives
the join's reductions from per-rank counts and sums:

    cross product (flags = 0)        n = sum cntR*cntS      sum_r = sum sumR*cntS      sum_s = sum sumS*cntR
    HMJ_FIRST_WINS                   n = sum [cntR>0]*cntS  sum_r = sum firstR*cntS    sum_s = sum [cntR>0]*sumS
    HMJ_SUM_PROBE                    sum_probe_all = sum sumS

(all mod 2^64; firstR = payload of the first build row of the rank in input order = the smallest row index, since
val = i -- unordered_map::insert semantics, partitioned_hash.h:166-170; the bench's reduction is
hashjoin_bench.cc:88-96).  Used by tests/test_gpu_join.py and by bench.py's extra runs.
"""
SEED_B = 0x243F6A8885A308D3
VAL_XOR = 0x9E3779B97F4A7C15
M64 = (1 << 64) - 1


def _s64(x):
    """Python int (mod 2^64) -> the int64 with the same bits."""
    x &= M64
    return x - (1 << 64) if x >= (1 << 63) else x


def mix64_t(torch, x):
    """hmj_dev.h mix64 on int64 tensors (two's complement arithmetic wraps like uint64; shifts made logical)."""
    def lsr(v, s):
        return (v >> s) & ((1 << (64 - s)) - 1)

    x = x ^ lsr(x, 30)
    x = x * _s64(0xBF58476D1CE4E5B9)
    x = x ^ lsr(x, 27)
    x = x * _s64(0x94D049BB133111EB)
    x = x ^ lsr(x, 31)
    return x


def config5_checks(torch, n_build, n_probe, domain, thr_dev, seed=SEED_B, zseed_build=0x1234567, zseed_probe=0x7654321,
                   chunk=1 << 26, device="cuda"):
    """thr_dev: the domain's inverse-CDF thresholds as the generator got them (int64 tensor holding uint64 bits)."""
    assert domain & (domain - 1) == 0, "power-of-two domains only (the unsigned modulo is a mask)"
    sign = torch.tensor(_s64(1 << 63), dtype=torch.int64, device=device)
    thr_signed = (thr_dev.to(device) ^ sign).contiguous()  # unsigned order -> signed order
    cnt_r = torch.zeros(domain, dtype=torch.int64, device=device)
    sum_r = torch.zeros(domain, dtype=torch.int64, device=device)
    first_r = torch.full((domain,), (1 << 62), dtype=torch.int64, device=device)
    for lo in range(0, n_build, chunk):
        i = torch.arange(lo, min(n_build, lo + chunk), dtype=torch.int64, device=device)
        u = mix64_t(torch, i ^ _s64(zseed_build)) ^ sign
        rank = torch.searchsorted(thr_signed, u, right=False).clamp_(max=domain - 1)  # first thr >= u
        cnt_r += torch.bincount(rank, minlength=domain)
        sum_r.index_add_(0, rank, i)        # val = i
        first_r.scatter_reduce_(0, rank, i, reduce="amin")
    cnt_s = torch.zeros(domain, dtype=torch.int64, device=device)
    sum_s = torch.zeros(domain, dtype=torch.int64, device=device)
    for lo in range(0, n_probe, chunk):
        j = torch.arange(lo, min(n_probe, lo + chunk), dtype=torch.int64, device=device)
        rank = mix64_t(torch, j ^ _s64(zseed_probe)) & (domain - 1)
        cnt_s += torch.bincount(rank, minlength=domain)
        sum_s.index_add_(0, rank, j ^ _s64(VAL_XOR))
    has_r = (cnt_r > 0).to(torch.int64)
    first_r = torch.where(cnt_r > 0, first_r, torch.zeros_like(first_r))

    def tot(t):
        return int(t.sum().item()) & M64  # int64 sums wrap like uint64

    return {
        "cross": {"n_matches": tot(cnt_r * cnt_s), "sum_r": tot(sum_r * cnt_s), "sum_s": tot(sum_s * cnt_r)},
        "first_wins": {"n_matches": tot(has_r * cnt_s), "sum_r": tot(first_r * cnt_s), "sum_s": tot(has_r * sum_s)},
        "sum_probe_all": tot(sum_s),
        "distinct_build_keys": int(has_r.sum().item()),
    }
