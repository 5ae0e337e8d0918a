"""The JSON line of bench.py on a real GPU (a short run): the keys and cross-checks the driver and the judge read."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_contract_short_run():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs the GPU")
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "2", "--no-cpu-baseline", "--no-parity"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    line = p.stdout.strip().splitlines()[-1]
    d = json.loads(line)                                          # the LAST stdout line is the JSON line
    assert d["metric"] == "vqa_train_samples_per_sec" and d["unit"] == "samples/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 2 and d["scaling"] == "weak" and d["data"] == "synthetic"
    assert d["dtype"] == "bf16" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"] and d["config"]["global_batch"] == 80
    assert abs(d["value"] - 80 / (d["ms_per_step"] * 1e-3)) < 0.01 * d["value"]                 # whole-job samples / s == B / step time
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0.05 < r["frac"] < 0.6
    assert r["gemm_ms_per_step"] < d["ms_per_step"] * 1.4                                         # (event-timed steps run a little longer)
    assert "traffic" in r and (r["traffic"] is None or r["traffic"] > r["algorithmic_bytes_per_launch"] * 0.5)
    dec = d["decode"]
    assert 0.2 < dec["ms_per_token_step"] < 3.0 and dec["tokens_per_sec"] > 1e4
    assert d["feed"]["bound"] == "hbm" and 0.0 < d["feed"]["frac"] <= 1.0
    assert d["samples_per_sec_pcie_inclusive"] <= d["samples_per_sec_resident_batch"] * 1.05


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("n,algo", [(2, "auto"), (4, "auto"), (8, "auto"), (2, "allreduce"), (4, "rs_ag")])
def test_bench_rehearsal_n_ranks_on_one_gpu(n, algo):
    """Dress rehearsal of the whole multi-rank bench path on the one GPU of the box: `bench.py --gpus N --rehearsal` -> launcher -> N rank
    processes sharing cuda:0 -> gloo rendezvous on 127.0.0.1 -> DataParallelVLT5 (zero1 for N in {2, 4, 8}) -> store-fed train steps with
    the real kernels -> in-situ GEMM timing on every rank -> rank 0's JSON line relayed as the last stdout line.  The line is flagged as a
    rehearsal and carries no value."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs the GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "VQACL_FORCE_DIST")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--rehearsal", "--steps", "2", "--warmup", "1", "--batch", "16",
           "--store-images", "256", "--dp-algo", algo, "--no-cpu-baseline", "--no-parity", "--launch-timeout", "1200"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=1400, env=env, cwd=ROOT)
    assert p.returncode == 0, (p.stderr[-3000:], p.stdout[-500:])
    d = json.loads(p.stdout.strip().splitlines()[-1])
    assert d["rehearsal"] is True and d["value"] is None and d["backend"] == "gloo"
    assert d["n_gpus"] == n and d["ranks_seen"] == n and d["launcher"] == "bench.py" and d["steps"] == 2
    want = algo if algo != "auto" else "zero1"
    assert d["grad_exchange"]["algo"] == want and d["grad_exchange"]["world"] == n
    assert d["grad_exchange"]["sharded_optimizer"] == (want == "zero1")
    assert d["weights_in_sync"] is True
    assert d["config"]["global_batch"] == 16 * n and "REHEARSAL" in d["config"]["parallelism"]
    assert d["rehearsal_samples_per_sec"] > 0 and d["final_loss"] == d["final_loss"]          # (finite)
    assert d["roofline"]["launches_per_step"] > 100                                             # rank 0's in-situ records of the extra steps
