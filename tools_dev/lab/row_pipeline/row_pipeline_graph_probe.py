"""Dev probe: GPU-side time of the hot-path step with / without the encoder's row pipeline, host launch cost taken out by
capturing the step into a hipGraph (torch.cuda.graph, relaxed capture mode) and timing replays.  The eager pipeline is
host-bound (tools_dev/row_pipeline_host_probe.py), so its wall time says nothing about the overlap itself."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from occnet_amd.plugin import encoder as enc_mod

_argv = sys.argv
sys.argv = ["bench.py"]
cfgp = os.path.join(os.path.dirname(os.path.abspath(bench.__file__)), "configs", "occ_base_200x200x16.py")
dev = torch.device("cuda:0")
cfg, model, geo = bench.build(cfgp, dev)
st = bench.Stepper(model, geo, "hotpath", "bf16", dev, seed=0, plan="folded")
enc = model.pts_bbox_head.transformer.encoder
ref = None
MODES = (("standard", 0, True), ("bands=2 no-serial", 2, False), ("bands=2 serial", 2, True),
         ("bands=3 no-serial", 3, False), ("bands=4 no-serial", 4, False))
if len(_argv) > 1:       # one mode per process (a capture that the runtime refuses can take the process down)
    MODES = tuple(m for i, m in enumerate(MODES) if str(i) in _argv[1:])
for mode, k, serial in MODES:
    enc_mod._ROW_PIPELINE, enc_mod._ROW_PIPELINE_SERIAL = k, serial
    enc._row_plan = None
    try:
        for _ in range(4):
            out = st()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="relaxed"):
            out = st()
        torch.cuda.synchronize()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        n = 40
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(n):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        o = out[0] if isinstance(out, (tuple, list)) else out
        o = o['occ'] if isinstance(o, dict) and 'occ' in o else o
        chk = ""
        if torch.is_tensor(o):
            if ref is None:
                ref = o.float().clone()
            else:
                chk = f"   max|occ - standard| {float((o.float() - ref).abs().max()):.2e}"
        mode = mode + (" [one stream]" if os.environ.get("OCC_ROW_PIPELINE_STREAMS") == "0" else "")
        print(f"{mode:22s} graph replay {e0.elapsed_time(e1) / n:.3f} ms/step (wall {1e3 * (t1 - t0) / n:.3f}){chk}", flush=True)
    except Exception as e:          # capture can refuse an API call: report and go on
        print(f"{mode:22s} FAILED: {type(e).__name__}: {str(e)[:300]}", flush=True)
        try:
            torch.cuda.synchronize()
        except Exception as e2:
            print("   (sync after failure:", str(e2)[:200], ")")
            break
