"""Lab (round 5): the hot-path step with the shipped library or with the mask-free-set-up variant (build_variant.py).
Prints sha256 of the outputs (bit-identity between the two is the claim) and ms per step (best of PASSES x STEPS).
usage: python tools_dev/lab/maskfree_setup/ab.py product|maskfree [steps] [passes]"""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import torch                                                        # noqa: E402
from occnet_amd import _lib                                         # noqa: E402

tag = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 2
if tag == "maskfree":
    _lib.LIB_PATH = os.path.join(ROOT, "tools_dev", "bin", "libocc_amd_maskfree.so")
from occnet_amd import synthetic                                    # noqa: E402
from tests.util import build_pair                                   # noqa: E402

g = dict(synthetic.BASE, num_points=8, num_layers=4)
prod, _ = build_pair(g, seed=12)


def nhwc(f):
    B, N, C, h, w = f.shape
    return f.reshape(B * N, C, h, w).cuda().contiguous(memory_format=torch.channels_last).view(B, N, C, h, w)


x = [nhwc(f.to(torch.bfloat16)) for f in synthetic.make_features(g, seed=12)]
metas = synthetic.make_img_metas(g)
keys = ("bev_embed", "occ", "flow")
with torch.no_grad():
    for _ in range(3):
        out = prod(x, metas)
    torch.cuda.synchronize()
    sha = {k: hashlib.sha256(out[k].float().cpu().numpy().tobytes()).hexdigest()[:16] for k in keys}
    best = []
    for _ in range(passes):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(steps):
            prod(x, metas)
        b.record()
        torch.cuda.synchronize()
        best.append(a.elapsed_time(b) / steps)
print(f"AB {tag:9s} lib={os.path.basename(_lib.LIB_PATH)} sha256 {sha} ms/step {['%.4f' % t for t in best]}", flush=True)
