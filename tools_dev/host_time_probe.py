"""Dev probe: host (launch-side) time per e2e step vs GPU time per step."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench

sys.argv = ["bench.py"]
args_cfg = os.path.join(os.path.dirname(os.path.abspath(bench.__file__)), "configs", "occ_base_200x200x16.py")
cfg, model, geo = bench.build(args_cfg, torch.device("cuda:0"))
st = bench.Stepper(model, geo, "e2e", "bf16", torch.device("cuda:0"), seed=0, plan="folded")
for _ in range(5):
    st()
torch.cuda.synchronize()
n = 40
t0 = time.perf_counter()
for _ in range(n):
    st()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue time per step {1e3 * (t1 - t0) / n:.2f} ms; wall per step {1e3 * (t2 - t0) / n:.2f} ms")
# unqueued host cost: two steps right after a sync (the launch queue cannot be full yet)
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); st(); st(); t1 = time.perf_counter()
    print(f"host time of 2 steps after a sync: {1e3 * (t1 - t0) / 2:.2f} ms per step")
import cProfile, pstats
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(10):
    st()
pr.disable(); torch.cuda.synchronize()
ps = pstats.Stats(pr); ps.sort_stats("tottime").print_stats(28)
