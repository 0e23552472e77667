"""tools/closed_forms.py (the rank-domain check of BASELINE configs[4] that the GPU tests and bench.py use at full size)
against the CPU oracle at sizes the oracle finishes in seconds.  No GPU."""
import numpy as np
import pytest

M64 = (1 << 64) - 1


def zipf_thresholds(domain, theta=0.9):
    w = 1.0 / np.arange(1, domain + 1, dtype=np.float64) ** theta
    cdf = np.cumsum(w) / w.sum()
    thr = np.empty(domain, np.uint64)
    big = cdf >= 1.0 - 2.0 ** -53
    thr[~big] = (cdf[~big] * 2.0 ** 64).astype(np.uint64)
    thr[big] = np.uint64(M64)
    thr[-1] = np.uint64(M64)
    return thr


@pytest.mark.parametrize("nb,npb,dom", [(1 << 14, 1 << 16, 1 << 14), (5000, 70001, 1 << 10), (3, 1000, 2)])
def test_rank_domain_closed_forms_match_the_oracle(nb, npb, dom):
    import torch

    from oracle.pyoracle import Oracle
    from tools.closed_forms import config5_checks

    orc = Oracle()
    thr = zipf_thresholds(dom)
    R, S = orc.gen_from_cdf(nb, thr), orc.gen_uniform_domain(npb, dom)
    ck, _ = orc.equijoin(R, S, cap=0)                      # cross product per key
    ckf, _ = orc.equijoin(R, S, first_wins=True, cap=0)    # unordered_map::insert, partitioned_hash.h:166-170
    got = config5_checks(torch, nb, npb, dom, torch.from_numpy(thr.view(np.int64).copy()), device="cpu", chunk=1 << 13)
    assert got["cross"] == {k: ck[k] for k in ("n_matches", "sum_r", "sum_s")}
    assert got["first_wins"] == {k: ckf[k] for k in ("n_matches", "sum_r", "sum_s")}
    assert got["sum_probe_all"] == int(np.sum(S[:, 1], dtype=np.uint64))
