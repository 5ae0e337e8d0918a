"""The JSON line of bench.py on a real GPU (a short run): the keys and cross-checks the driver and the judge read."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


STALL_DIR = os.path.join(ROOT, "gpurun_out", "stall_dumps")


def group_members(pgid):
    """[(pid, cmdline)] of the processes whose process group is `pgid` (the session run_bounded() started: launcher + ranks)."""
    out = []
    for name in os.listdir("/proc"):
        if not name.isdigit():
            continue
        try:
            if os.getpgid(int(name)) != pgid:
                continue
            with open(f"/proc/{name}/cmdline", "rb") as f:
                out.append((int(name), f.read().replace(b"\0", b" ").decode(errors="replace").strip()))
        except (OSError, ProcessLookupError):
            pass
    return out


def run_bounded(cmd, env, timeout):
    """subprocess.run with a bound -- and NO retry.  One full-suite run of round 5 (of about ten) stalled for more than 17 minutes somewhere
    in the multi-process tests and was not reproduced; until round 6 a launch that ran into its bound was started again and only a warning
    was left, which would let a real intermittent deadlock of the launcher or of DataParallelVLT5 pass.  Now a launch that is still running
    after `timeout` seconds (the launcher itself gives up first: --launch-timeout) is asked for the Python stacks of every bench.py process of
    its process group (SIGUSR1 -> bench.arm_stall_dump, files under gpurun_out/stall_dumps/), the group is ended, and the test FAILS with
    the dump attached.  (The retrying loop is an investigation aid: tools/rehearsal_loop.sh.)"""
    import signal
    import time
    tag = time.strftime("%H%M%S") + "_%d" % os.getpid()
    dump_dir = os.path.join(STALL_DIR, tag)
    env = dict(env, VQACL_STALL_DUMP_DIR=dump_dir)
    # own session: on a stall the launcher AND the rank processes it started are ended (exactly that process group)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=ROOT, start_new_session=True)
    try:
        out, err = proc.communicate(timeout=timeout)
        import shutil
        shutil.rmtree(dump_dir, ignore_errors=True)        # (the ranks open their dump files when they arm the handler: empty unless asked)
        return subprocess.CompletedProcess(cmd, proc.returncode, out, err)
    except subprocess.TimeoutExpired:
        members = group_members(proc.pid)
        for pid, cl in members:                 # only the processes that armed the handler: SIGUSR1 would END torch.distributed.run
            if "bench.py" in cl and "torch.distributed.run" not in cl:
                try:
                    os.kill(pid, signal.SIGUSR1)
                except ProcessLookupError:
                    pass
        time.sleep(5.0)
        try:
            os.killpg(proc.pid, signal.SIGTERM)
            out, err = proc.communicate(timeout=30)
        except Exception:
            try:
                os.killpg(proc.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
            out, err = proc.communicate()
        dumps = []
        if os.path.isdir(dump_dir):
            for f in sorted(os.listdir(dump_dir)):
                with open(os.path.join(dump_dir, f)) as fh:
                    dumps.append(f"--- {f} ---\n" + fh.read()[-3000:])
            with open(os.path.join(dump_dir, "launch.txt"), "w") as fh:
                fh.write(" ".join(cmd) + "\n\nprocesses of the group at the bound:\n" + "\n".join(f"{p} {c}" for p, c in members)
                         + "\n\nstderr tail:\n" + (err or "")[-6000:] + "\n\nstdout tail:\n" + (out or "")[-2000:])
        msg = (f"bench launch still running after {timeout} s: {' '.join(cmd[1:8])} ...\nstacks of its processes ({dump_dir}):\n"
               + "\n".join(dumps) + f"\nstderr tail: {(err or '')[-1500:]}")
        try:
            from test_gpu_model import parity_log
            parity_log("STALL " + msg.splitlines()[0] + f" (dumps: {dump_dir})")
        except Exception:
            pass
        pytest.fail(msg, pytrace=False)


def test_bench_line_contract_short_run():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs the GPU")
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    p = run_bounded([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "2", "--no-cpu-baseline", "--no-parity"], env, timeout=500)
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
    # round 5: the cold (in-step-like) feed figure is the one quoted, the cache-resident replay is the side value
    assert d["feed"]["us"] >= d["feed"]["us_replay_cache_resident"] * 0.9 and d["feed"]["frac"] <= d["feed"]["frac_replay_cache_resident"] * 1.1
    assert set(d["configs"]) == {"c1_b4", "c4_v16", "c4_v32", "c5_large_b32", "c5_large_b80"}
    for k, c in d["configs"].items():
        assert c["ms_per_step"] > 0 and abs(c["samples_per_sec"] - c["batch"] * 1e3 / c["ms_per_step"]) < 0.01 * c["samples_per_sec"]
        assert 0.0 < c["step_frac_of_mfma_peak"] < 0.5 and 0.0 < c["clip_adamw_share"] < 0.9
    assert d["configs"]["c5_large_b32"]["adamw_ms_one_eighth_shard"] < d["configs"]["c5_large_b32"]["clip_adamw_ms"]
    sw = dec["rows_sweep"]
    assert set(sw) == {"100", "160", "320", "512"} and all(v["ms_per_token_step"] > 0 for v in sw.values())
    assert sw["512"]["tokens_per_sec"] > dec["tokens_per_sec"]                 # more rows per launch: more tokens per second
    assert dec["weight_bytes_per_token_step"] == 2 * (12 * (6 * 768 * 768 + 2 * 768 * 3072) + 32200 * 768)


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("n,algo", [(2, "auto"), (4, "auto"), (8, "auto"), (2, "allreduce")])
def test_bench_rehearsal_n_ranks_on_one_gpu(n, algo):
    """Dress rehearsal of the whole multi-rank bench path on the one GPU of the box: `bench.py --gpus N --rehearsal` -> launcher -> N rank
    processes sharing cuda:0 -> gloo rendezvous on 127.0.0.1 -> DataParallelVLT5 (zero1 for N in {2, 4, 8}) -> store-fed train steps with
    the real kernels -> in-situ GEMM timing on every rank -> rank 0's JSON line relayed as the last stdout line.  The line is flagged as a
    rehearsal and carries no value."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs the GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "VQACL_FORCE_DIST")}
    # (gloo moves ~2 GB per rank and step through the host: a step of 8 ranks takes about a minute -- the wide runs are kept short)
    steps = 2 if n <= 2 else 1
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--rehearsal", "--steps", str(steps), "--warmup", "1", "--batch", "16",
           "--store-images", "256", "--dp-algo", algo, "--no-cpu-baseline", "--no-parity", "--launch-timeout", "540"] + (["--no-roofline"] if n > 4 else [])
    p = run_bounded(cmd, env, timeout=600)                     # (N = 8 takes about 150 s)
    assert p.returncode == 0, (p.stderr[-3000:], p.stdout[-500:])
    d = json.loads(p.stdout.strip().splitlines()[-1])
    assert d["rehearsal"] is True and d["value"] is None and d["ms_per_step"] is None and d["backend"] == "gloo"
    assert d["n_gpus"] == n and d["ranks_seen"] == n and d["launcher"] == "bench.py" and d["steps"] == steps
    want = algo if algo != "auto" else "zero1"
    assert d["grad_exchange"]["algo"] == want and d["grad_exchange"]["world"] == n
    assert d["grad_exchange"]["sharded_optimizer"] == (want == "zero1")
    assert d["weights_in_sync"] is True
    assert d["config"]["global_batch"] == 16 * n and "REHEARSAL" in d["config"]["parallelism"]
    assert d["rehearsal_samples_per_sec"] > 0 and d["final_loss"] == d["final_loss"]          # (finite)
    if n <= 4:
        assert d["roofline"]["launches_per_step"] > 100                                         # rank 0's in-situ records of the extra steps


@pytest.mark.timeout(900)
def test_bench_rehearsal_under_torch_distributed_run():
    """The command shape the driver uses for N > 1 -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...` -- with `--rehearsal` added so that both ranks can share the one GPU: bench.py must recognise that it
    IS a rank (WORLD_SIZE inherited), not start ranks of its own, and rank 0 must print the one JSON line."""
    import socket
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs the GPU")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "VQACL_FORCE_DIST")}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearsal", "--steps", "2", "--warmup", "1", "--batch", "16", "--store-images", "256",
           "--no-cpu-baseline", "--no-parity", "--no-roofline"]
    p = run_bounded(cmd, env, timeout=400)
    assert p.returncode == 0, (p.stderr[-3000:], p.stdout[-500:])
    lines = [ln for ln in p.stdout.strip().splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, p.stdout[-2000:]                       # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["rehearsal"] is True and d["value"] is None
    assert d["launcher"] == "external" and d["weights_in_sync"] is True and d["grad_exchange"]["algo"] == "zero1"
