import os, time, torch, sys
sys.path.insert(0, os.getcwd())
from oracle import ref_cpu as R
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
cfg = R.Cfg(dropout=0.1)
model = R.OracleModel(cfg, seed=0)
batch = R.synthetic_batch(cfg, B=4, L=20, V=36, T=5, seed=1)
for nt in (8, 16, 32, 64):
    torch.set_num_threads(nt)
    ts=[]
    for it in range(2):
        t0=time.time(); model.zero_grad(); o=model.train_step(batch,0,0.5,0.3); o["loss"].backward(); ts.append(time.time()-t0)
    print("threads", nt, "step s", [round(t,2) for t in ts], flush=True)
