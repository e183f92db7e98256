"""Dev probe: is the encoder's row pipeline (OCC_ENCODER_ROW_PIPELINE) host-bound?  Hot-path scope; per mode: wall time per
step (queue full), host enqueue time per step, host time of a step right after a sync (unqueued launch cost)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from occnet_amd.plugin import encoder as enc_mod

sys.argv = ["bench.py"]
cfgp = os.path.join(os.path.dirname(os.path.abspath(bench.__file__)), "configs", "occ_base_200x200x16.py")
dev = torch.device("cuda:0")
cfg, model, geo = bench.build(cfgp, dev)
st = bench.Stepper(model, geo, "hotpath", "bf16", dev, seed=0, plan="folded")
for mode, k, serial, native in (("standard", 0, False, False), ("bands=2", 2, False, False), ("bands=2 serial", 2, True, False),
                                ("bands=2, one stream", 2, False, False), ("native launcher, unbanded", 1, False, True),
                                ("native launcher, bands=2", 2, False, True), ("native launcher, bands=3", 3, False, True)):
    enc_mod._ROW_PIPELINE, enc_mod._ROW_PIPELINE_SERIAL, enc_mod._ROW_PIPELINE_NATIVE = k, serial, native
    os.environ["OCC_ROW_PIPELINE_STREAMS"] = "0" if "one stream" in mode else "1"
    model.pts_bbox_head.transformer.encoder._row_plan = None
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
    torch.cuda.synchronize()
    ta = time.perf_counter(); st(); st(); tb = time.perf_counter()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    print(f"{mode:32s} wall {1e3 * (t2 - t0) / n:.3f} ms/step   host enqueue {1e3 * (t1 - t0) / n:.3f} ms/step   "
          f"host, unqueued {1e3 * (tb - ta) / 2:.3f} ms/step", flush=True)
