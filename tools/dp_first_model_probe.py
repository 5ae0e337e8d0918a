#!/usr/bin/env python3
"""Does the order of dist.init_process_group() and the first GPU touch change the speed of a DataParallelVLT5 step?  (Round 5: with a HIGH-priority
communication stream a step took 26 ms instead of 9.2 whenever the GPU had been touched before init; profiles/r05_w_comm_stream_priority.txt.)
    python tools/dp_first_model_probe.py {init_first | touch_then_init | import_then_init | init_late | extra_gpu_work}
environment: PROBE_PG_PRIO=0 (process group stream at normal priority), PROBE_EAGER=0 (no device_id), PROBE_COMM_PRIO=0 (wrapper stream forced normal)."""
import os, sys, time, socket
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
mode = sys.argv[1]
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
import torch.distributed as dist
def init():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=os.environ.get("PROBE_PG_PRIO", "1") == "1")
    kw = dict(device_id=dev) if os.environ.get("PROBE_EAGER", "1") == "1" else {}
    dist.init_process_group("nccl", rank=0, world_size=1, pg_options=opts, **kw)
if mode == "init_first":
    init()
if mode == "touch_then_init":
    torch.zeros(1, device=dev); torch.cuda.synchronize(); init()
from bench import synthetic_batch
from vqacl_amd import VLT5VQA, VLT5Config, FusedAdamW, reference_param_groups
from vqacl_amd.parallel import DataParallelVLT5
if mode == "import_then_init":
    init()
batch = {k: v.to(dev) for k, v in synthetic_batch(80, seed=1).items()}
if mode in ("init_late", "extra_gpu_work"):
    init()
model = VLT5VQA(VLT5Config(dropout_rate=0.1), device=dev); model.train()
dp = DataParallelVLT5(model, algo="allreduce")
if os.environ.get("PROBE_COMM_PRIO") == "0":
    dp.comm_stream = torch.cuda.Stream()
opt = FusedAdamW(reference_param_groups(model, 0.01), dp, lr=1e-4, eps=1e-6, max_grad_norm=5.0)
if mode == "extra_gpu_work":
    for _ in range(16):
        x = torch.randn(256, 36, 2048, device=dev); y = torch.relu(x) * 1.5
    torch.cuda.synchronize()
def step():
    res = dp.train_step(batch, 0, 0.5, 0.3); res["loss"].backward(); opt.step()
    for p in model.parameters(): p.grad = None
for rep in range(3):
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize(); ms = round((time.perf_counter() - t0) / 20 * 1e3, 2); print(mode, "block", rep, ms, "ms/step", flush=True)
print("RESULT", mode, ms, getattr(dp.comm_stream, "priority", None), flush=True)
dist.destroy_process_group()
