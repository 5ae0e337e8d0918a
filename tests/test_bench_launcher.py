"""bench.py's rank launcher on CPU: `python bench.py --gpus N` without WORLD_SIZE has to start N ranks itself (before torch is
imported, without touching a GPU), relay rank 0's JSON line as its last stdout line and fail when a rank fails; under an external
launcher (WORLD_SIZE set) it must be a rank, and a --gpus that contradicts WORLD_SIZE is an error, not a silent 1-rank number."""
import json
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                          "VQACL_FORCE_DIST")}
    env.update(extra)
    return env


@pytest.mark.parametrize("n", [1, 2, 8])
def test_dry_launch_prints_one_rank_environment_per_gpu(n):
    flags = ["--gpus", str(n), "--steps", "3", "--warmup", "1", "--dry-launch"] + (["--force-dist"] if n == 1 else [])
    res = subprocess.run([sys.executable, BENCH] + flags, env=_clean_env(), capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    plans = [json.loads(line) for line in res.stdout.strip().splitlines()]
    assert len(plans) == n
    ports = {p["env"]["MASTER_PORT"] for p in plans}
    assert len(ports) == 1 and 1024 < int(ports.pop()) < 65536
    for r, p in enumerate(plans):
        e = p["env"]
        assert (e["RANK"], e["LOCAL_RANK"], e["WORLD_SIZE"], e["LOCAL_WORLD_SIZE"]) == (str(r), str(r), str(n), str(n))
        assert e["MASTER_ADDR"] == "127.0.0.1" and e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
        assert (e.get("VQACL_FORCE_DIST") == "1") == (n == 1)
        assert p["cmd"][1] == BENCH and "--dry-launch" not in p["cmd"]
        assert p["cmd"][2:4] == ["--gpus", str(n)] and "--steps" in p["cmd"]


def test_plain_single_gpu_run_and_external_launcher_are_not_relaunched():
    sys.path.insert(0, ROOT)
    import bench
    args = bench.parse_args(["--gpus", "1"])
    assert bench.launch_plan(args, ["--gpus", "1"], environ={}) is None                       # the driver's 1-GPU call: one process
    args = bench.parse_args(["--gpus", "4"])
    assert bench.launch_plan(args, ["--gpus", "4"], environ={"WORLD_SIZE": "4", "RANK": "2"}) is None     # a rank of torch.distributed.run
    assert len(bench.launch_plan(args, ["--gpus", "4"], environ={}, port=29999)) == 4
    with pytest.raises(SystemExit) as ei:                                                  # --gpus 8 under a 1-rank WORLD_SIZE: refuse
        bench.launch_plan(bench.parse_args(["--gpus", "8"]), ["--gpus", "8"], environ={"WORLD_SIZE": "1"})
    assert "WORLD_SIZE=1" in str(ei.value)
    # VQACL_FORCE_DIST=1 (the old switch) still goes through the launcher
    one = bench.launch_plan(bench.parse_args(["--gpus", "1"]), ["--gpus", "1"], environ={"VQACL_FORCE_DIST": "1"}, port=29999)
    assert len(one) == 1 and one[0][0]["WORLD_SIZE"] == "1" and one[0][0]["VQACL_FORCE_DIST"] == "1"


def test_rehearsal_goes_through_the_launcher_even_with_one_rank():
    sys.path.insert(0, ROOT)
    import bench
    for n in (1, 2, 8):
        argv = ["--gpus", str(n), "--rehearsal"]
        plan = bench.launch_plan(bench.parse_args(argv), argv, environ={}, port=29998)
        assert len(plan) == n and all("--rehearsal" in cmd for _, cmd in plan)
        assert [e["RANK"] for e, _ in plan] == [str(r) for r in range(n)]


def test_launcher_refuses_to_start_ranks_under_a_preloaded_profiler(monkeypatch, capfd):
    """rocprofv3's preloaded tool library has initialised the GPU before bench.py starts: ranks must not be started from it."""
    sys.path.insert(0, ROOT)
    import bench
    assert bench.profiler_preload({}) is None and bench.profiler_preload({"LD_PRELOAD": "/usr/lib/libfoo.so"}) is None
    assert "rocprofiler" in bench.profiler_preload({"LD_PRELOAD": "/opt/rocm/lib/rocprofiler-sdk/librocprofiler-sdk-tool.so"})
    monkeypatch.setattr(bench, "profiler_preload", lambda environ=None: "LD_PRELOAD=/opt/rocm/lib/rocprofiler-sdk/librocprofiler-sdk-tool.so")
    argv = ["--gpus", "1", "--force-dist"]
    for k in ("WORLD_SIZE", "RANK", "VQACL_FORCE_DIST"):
        monkeypatch.delenv(k, raising=False)
    rc = bench.run_launcher(bench.parse_args(argv), argv)
    out, err = capfd.readouterr()
    assert rc == 2 and "refusing to start ranks under a preloaded profiler" in err and "VQACL_FORCE_DIST=1" in err and '"metric"' not in out
    # a dry launch and a plain single-process run are unaffected
    assert bench.run_launcher(bench.parse_args(["--gpus", "1"]), ["--gpus", "1"]) is None


def test_gpus_contradicting_world_size_fails_loudly():
    res = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--dry-launch"], env=_clean_env(WORLD_SIZE="2"), capture_output=True, text=True,
                         timeout=120)
    assert res.returncode != 0 and "--gpus 8 but WORLD_SIZE=2" in (res.stderr + res.stdout)


def _stub_plan(monkeypatch, bodies):
    import bench
    plan = [({"RANK": str(r), "WORLD_SIZE": str(len(bodies))}, [sys.executable, "-c", body]) for r, body in enumerate(bodies)]
    monkeypatch.setattr(bench, "launch_plan", lambda args, argv, environ=None, port=None: plan)
    return bench


def test_launcher_relays_rank0_line_last(monkeypatch, capfd):
    line = json.dumps({"metric": "vqa_train_samples_per_sec", "value": 1.0, "n_gpus": 2, "rccl_ranks_seen": 2})
    sys.path.insert(0, ROOT)
    bench = _stub_plan(monkeypatch, [
        f"import sys; print('RCCL version banner'); print({line!r}); print('trailing noise', file=sys.stderr)",
        "import time; time.sleep(0.3); print('rank 1 chatter')"])
    rc = bench.run_launcher(bench.parse_args(["--gpus", "2"]), ["--gpus", "2"])
    out, err = capfd.readouterr()
    assert rc == 0
    assert out.strip().splitlines()[-1] == line and out.count('"metric"') == 1       # the line, once, last on stdout
    assert "RCCL version banner" in err and "rank 1 chatter" in err                        # everything else goes to stderr


def test_launcher_fails_when_a_rank_fails_and_stops_the_others(monkeypatch, capfd):
    sys.path.insert(0, ROOT)
    bench = _stub_plan(monkeypatch, ["import time; time.sleep(120)", "import sys; sys.exit(3)"])
    t0 = time.time()
    rc = bench.run_launcher(bench.parse_args(["--gpus", "2"]), ["--gpus", "2"])
    assert rc == 3 and time.time() - t0 < 60, "the surviving rank is terminated, not waited for"
    out, err = capfd.readouterr()
    assert '"metric"' not in out and "rank 1 exited with 3" in err


def test_launcher_needs_a_json_line(monkeypatch, capfd):
    sys.path.insert(0, ROOT)
    bench = _stub_plan(monkeypatch, ["print('no line here')"])
    assert bench.run_launcher(bench.parse_args(["--gpus", "1", "--force-dist"]), ["--gpus", "1", "--force-dist"]) == 1


STALLED_RANK = ("import sys, time; sys.path.insert(0, %r); import bench; bench.arm_stall_dump('rank' + __import__('os').environ.get('RANK', '0'));\n"
                "def stuck_in_a_collective():\n    time.sleep(120)\nstuck_in_a_collective()   # stands in for bench.py") % ROOT


def test_launcher_asks_stalled_ranks_for_their_stacks_before_it_gives_up(monkeypatch, capfd, tmp_path):
    """A launch that runs into --launch-timeout leaves, per rank, the Python stack the rank sits in (SIGUSR1 -> faulthandler), then ends
    the ranks and returns 124 -- a stall is a failure with an artifact, never a silent retry."""
    sys.path.insert(0, ROOT)
    monkeypatch.setenv("VQACL_STALL_DUMP_DIR", str(tmp_path))
    bench = _stub_plan(monkeypatch, [STALLED_RANK, STALLED_RANK])
    t0 = time.time()
    rc = bench.run_launcher(bench.parse_args(["--gpus", "2", "--launch-timeout", "4"]), ["--gpus", "2"])
    assert rc == 124 and time.time() - t0 < 60
    dumps = sorted(f for f in os.listdir(tmp_path) if f.startswith("rank"))
    assert len(dumps) == 2 and dumps[0].startswith("rank0_pid") and dumps[1].startswith("rank1_pid"), dumps
    for f in dumps:
        assert "stuck_in_a_collective" in open(tmp_path / f).read()
    assert "asking them for their stacks" in capfd.readouterr().err


def test_bounded_launch_of_the_gpu_tests_fails_with_the_stacks_attached(monkeypatch, tmp_path):
    """tests/test_gpu_bench_line.py::run_bounded: no retry; a launch over its bound fails the test and the failure carries the stacks."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_gpu_bench_line as T
    monkeypatch.setattr(T, "STALL_DIR", str(tmp_path))
    with pytest.raises(pytest.fail.Exception) as ei:
        T.run_bounded([sys.executable, "-c", STALLED_RANK], _clean_env(), timeout=4)
    assert "still running after 4 s" in str(ei.value) and "stuck_in_a_collective" in str(ei.value)
    sub = [d for d in os.listdir(tmp_path)]
    assert len(sub) == 1 and "launch.txt" in os.listdir(tmp_path / sub[0])
    # a launch inside its bound is returned as subprocess.run would return it
    res = T.run_bounded([sys.executable, "-c", "print('fine')"], _clean_env(), timeout=30)
    assert res.returncode == 0 and res.stdout.strip() == "fine"
