"""bench.py's N > 1 path on ONE GPU: two ranks launched exactly as the driver launches them (torch.distributed.run), with
``--backend gloo`` so that both may share cuda:0 (RCCL refuses two ranks on one device).  Checks the contract line: ``n_gpus``,
``scaling``, whole-job ``value`` = ranks x steps / max-over-ranks time, the weight broadcast really happened."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _line(out: str) -> dict:
    lines = [l for l in out.splitlines() if l.startswith("{") and '"metric"' in l]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("config,ranks", [("dev1024", 2), ("qwen1024", 2), ("dev1024", 8)])
def test_bench_n_ranks_on_one_gpu(config, ranks):
    """ranks = 8: the dry run of the driver's 8-GPU launch line -- LOCAL_RANK -> device mapping, the bucketed weight broadcast
    from rank 0, per_replica over 8 ranks; the first real 8-GPU lease needs no code change (only the backend differs: nccl = RCCL)."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    common = ["--steps", "2", "--warmup", "1", "--prof-steps", "2", "--no-cpu-baseline", "--config", config, "--resolution", "256", "--txt-tokens", "256"]
    layers = ["--layers", "1", "1"]
    env = dict(os.environ, PYTHONPATH=ROOT)
    env.pop("HSA_ENABLE_IPC_MODE_LEGACY", None)  # bench.py must set it itself
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--backend", "gloo", *layers, *common]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == ranks and d["scaling"] == "weak" and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["config"]["output_finite"] is True and f"{ranks} independent replica" in d["config"]["parallelism"]
    assert "0.00 GB" not in d["config"]["parallelism"], "the weight broadcast moved no bytes"
    # whole-job value: all ranks' steps over the max-over-ranks time
    assert abs(d["value"] * d["ms_per_step"] / 1e3 - float(ranks)) < 1e-6
    # the contract line's time carries no instrumentation; the instrumented steps behind it are reported beside it (round 5)
    assert d["ms_per_step_instrumented"] is not None and d["ms_per_step_instrumented"] > 0 and d["roofline"]["prof_steps"] == 2
    assert d["cpu_baseline"] is None and d["roofline"]["launches"] > 0
    # round 4: per-replica rates, the roofline split per epilogue variant, staleness stamps of the committed PMC numbers
    pr = d["per_replica"]
    assert 0 < pr["min_steps_per_s"] <= pr["max_steps_per_s"] and d["value"] <= ranks * pr["max_steps_per_s"] * 1.0001
    rf = d["roofline"]
    assert isinstance(rf["traffic_stale"], bool) and (rf["mfma_util"] is None or isinstance(rf["mfma_util"]["stale"], bool))
    assert rf["epilogue_share"] is None or (isinstance(rf["epilogue_share"]["stale"], bool) and "gelu_quant" in rf["epilogue_share"]["per_variant"])
    assert rf["effective_clock_ghz"] is None or 0.05 < rf["effective_clock_ghz"] < 3.0
    pv = rf["per_variant"]
    assert {"default", "gelu_quant", "rmsnorm_rope"} <= set(pv), pv.keys()
    assert abs(sum(v["launches_per_step"] for v in pv.values()) * rf["prof_steps"] - rf["launches"]) < 1e-6
    assert all(v["avg_launch_us"] > 0 and 0 < v["frac"] < 1 for v in pv.values())
    assert abs(sum(v["ms_per_step"] for v in pv.values()) - rf["gemm_ms_per_step"]) < 1e-3 * max(1.0, rf["gemm_ms_per_step"])
