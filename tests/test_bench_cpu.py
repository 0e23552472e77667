"""bench.py's own launcher (`python bench.py --gpus N` without torchrun) must end in bounded time whatever its ranks do
(VERDICT r4 #1): a wall-clock limit, every child stopped by its own handle, one JSON line with "error", non-zero exit.
No GPU: the children are stand-ins."""
import importlib.util
import io
import json
import os
import sys
import time
from contextlib import redirect_stderr, redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_bench():
    spec = importlib.util.spec_from_file_location("hmj_bench", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def run(bench, n, timeout_s, code):
    out, err = io.StringIO(), io.StringIO()
    t0 = time.time()
    # (self_launch writes through sys.stdout / sys.stderr of this process)
    with redirect_stdout(out), redirect_stderr(err):
        rc = bench.self_launch(n, timeout_s, argv=[sys.executable, "-c", code], steps=3, warmup=1)
    return rc, out.getvalue(), err.getvalue(), time.time() - t0


def test_all_ranks_stuck_ends_at_the_wall_clock_limit():
    bench = load_bench()
    rc, out, err, dt = run(bench, 3, 2.0, "import time; time.sleep(600)")
    assert rc != 0 and dt < 15, (rc, dt)
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    d = json.loads(lines[0])
    assert d["value"] is None and "wall-clock limit" in d["error"] and d["n_gpus"] == 3 and d["steps"] == 3, d
    assert d["metric"] == bench.METRIC
    assert "rank 2" in err


def test_one_rank_dies_the_others_are_stopped():
    bench = load_bench()
    code = "import os, sys, time\nif os.environ['RANK'] == '1':\n    sys.stderr.write('boom'); sys.exit(7)\ntime.sleep(600)"
    os.environ["HMJ_BENCH_GRACE_S"] = "1"
    try:
        rc, out, err, dt = run(bench, 2, 120.0, code)
    finally:
        del os.environ["HMJ_BENCH_GRACE_S"]
    assert rc != 0 and dt < 20, (rc, dt)
    d = json.loads([l for l in out.splitlines() if l.startswith("{")][0])
    assert "rank(s) 1 failed" in d["error"] and d["detail"]["rank1"] == 7, d
    assert "boom" in err


def test_a_good_run_relays_rank0s_line_and_returns_zero():
    bench = load_bench()
    code = "import os\nif os.environ['RANK'] == '0':\n    print('{\"value\": 1}', flush=True)"
    rc, out, err, dt = run(bench, 2, 60.0, code)
    assert rc == 0 and out.strip() == '{"value": 1}', (rc, out, err)


def test_rank0s_own_error_line_is_not_doubled():
    bench = load_bench()
    code = ("import os, sys\nif os.environ['RANK'] == '0':\n    print('{\"error\": \"exchange step 2 failed\", \"value\": null}', flush=True)\n"
            "sys.exit(5)")
    rc, out, err, dt = run(bench, 2, 60.0, code)
    assert rc == 5 and len([l for l in out.splitlines() if l.startswith("{")]) == 1, (rc, out)


def test_live_traffic_measurement_gives_way_quietly_without_a_gpu(monkeypatch):
    # roofline.traffic is measured by two rocprofv3 --pmc child runs of bench.py (N = 1).  Whatever goes wrong there -- no
    # profiler, no GPU (this container), a pass past its time -- must come back as (None, reason), never as an exception:
    # the line then carries the committed constant, labelled as such.
    bench = load_bench()
    t0 = time.time()
    got, why = bench.measure_traffic_live(20, timeout_s=90.0)
    assert got is None and isinstance(why, str) and why, (got, why)
    assert time.time() - t0 < 120
    monkeypatch.setenv("PATH", "/nonexistent")
    monkeypatch.setattr(bench.os.path, "exists", lambda p: False)
    assert bench.measure_traffic_live(20, timeout_s=5.0) == (None, "rocprofv3 not found")
