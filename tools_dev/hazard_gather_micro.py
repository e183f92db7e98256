"""Development (round 5): the gather micro-victim of tools_dev/hazard_micro.hip (the TSA gather's structure on a table with
known contents) next to several neighbours on another stream.  usage: python tools_dev/hazard_gather_micro.py [reps]"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from occnet_amd import ext                                        # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
lib = ctypes.CDLL(os.path.join(ROOT, "tools_dev", "bin", "libhazard_micro.so"))
P = ctypes.c_void_p
st = lambda: P(torch.cuda.current_stream().cuda_stream)
n_rows = 320000                      # 41 MB: the projected BEV of the TSA gather
table = torch.empty(n_rows * 32, dtype=torch.float32, device='cuda')
lib.hz_fill_small(P(table.data_ptr()), ctypes.c_uint32(n_rows), st())
sink = torch.zeros(4, device='cuda')
g = torch.Generator().manual_seed(3)
ca = dict(attn=torch.randn(1, 40000, 256, device='cuda'), q=torch.randn(1, 40000, 256, device='cuda'),
          w1=((torch.rand(256, 256, generator=g) * 2 - 1) * 0.06).cuda(), b1=torch.zeros(256, device='cuda'),
          ln=torch.nn.LayerNorm(256).cuda(), w2=((torch.rand(768, 256, generator=g) * 2 - 1) * 0.06).cuda(),
          b2=torch.zeros(768, device='cuda'))
a = torch.randn(4096, 4096, device='cuda', dtype=torch.bfloat16)
from occnet_amd import synthetic                                   # noqa: E402
feats = synthetic.make_features(dict(synthetic.BASE), seed=12)
maps = [f.reshape(-1, 256, f.shape[3], f.shape[4]).cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last) for f in feats]
map_rows = [m.permute(0, 2, 3, 1).reshape(-1, 256) for m in maps]
hw = [m.shape[2] * m.shape[3] for m in maps]
starts = [sum(hw[:i]) for i in range(len(hw))]
total = sum(hw) + (sum(hw) & 1)
ws = [((torch.rand(256, 256, generator=g) * 2 - 1) * 0.1).cuda() for _ in range(4)]
gbs = [torch.randn(4, 6, 256, generator=g).cuda() for _ in range(4)]
planes = torch.empty(4, 6 * total, 256, dtype=torch.float16, device='cuda')
loads = {
    "occ value projection (x2)": lambda: [ext.value_proj_bf16_planes(map_rows, ws, gbs, planes, rows_per_group=hw, out_group_rows=total,
                                                                     out_row0=starts) for _ in range(2)],
    "none": None,
    "hipBLASLt GEMM": lambda: [a @ a for _ in range(4)],
    "occ chain program A": lambda: [ext.linear_ln_chain(ca['attn'], ca['q'], ca['w1'], ca['b1'], ca['ln'], ca['w2'], ca['b2']) for _ in range(4)],
    "synthetic MFMA + LDS reads": lambda: lib.hz_spin(P(sink.data_ptr()), 1, 512, 4000, st()),
    "synthetic MFMA only": lambda: lib.hz_spin(P(sink.data_ptr()), 0, 512, 4000, st()),
}
load = torch.cuda.Stream()
for fn in loads.values():
    if fn is not None:
        with torch.cuda.stream(load):
            fn()
torch.cuda.synchronize()
err = torch.zeros(1, dtype=torch.int64, device='cuda')
for lname, lfn in loads.items():
    for vname, ldsp in (("parameters through the LDS slab", 1), ("parameters in registers", 0)):
        err.zero_()
        bad_reps = 0
        for rep in range(reps):
            before = int(err.item())
            if lfn is not None:
                with torch.cuda.stream(load):
                    lfn()
            for _ in range(3):
                lib.hz_gather_victim(P(table.data_ptr()), ctypes.c_uint32(n_rows), P(err.data_ptr()), 10000, 4, ldsp, st())
            torch.cuda.synchronize()
            bad_reps += int(err.item()) > before
        print(f"GATHER-MICRO neighbour = {lname:28s} victim = gather, {vname:32s}: {int(err.item()):6d} wrong words, "
              f"{bad_reps} of {reps} repetitions", flush=True)

# ---- the output side: row stores of a gather-shaped kernel, read back by the next kernel ------------------------------------
out = torch.zeros(40000 * 256, dtype=torch.int32, device='cuda')
err2 = torch.zeros(2, dtype=torch.int64, device='cuda')
epoch = 1
for lname, lfn in loads.items():
    err2.zero_()
    bad_reps = 0
    for rep in range(reps):
        before = int(err2[0].item())
        if lfn is not None:
            with torch.cuda.stream(load):
                lfn()
        for _ in range(3):
            epoch += 1
            lib.hz_store_victim(P(table.data_ptr()), ctypes.c_uint32(n_rows), P(out.data_ptr()), ctypes.c_uint32(40000),
                                ctypes.c_uint32(epoch), P(err2.data_ptr()), 0, st())
            lib.hz_store_victim(P(table.data_ptr()), ctypes.c_uint32(n_rows), P(out.data_ptr()), ctypes.c_uint32(40000),
                                ctypes.c_uint32(epoch), P(err2.data_ptr()), 1, st())
        torch.cuda.synchronize()
        bad_reps += int(err2[0].item()) > before
    print(f"STORE-MICRO neighbour = {lname:28s} victim = one 64-lane x 16-byte row store per wave: {int(err2[0].item()):6d} wrong words "
          f"({int(err2[1].item())} = last epoch's), {bad_reps} of {reps} repetitions", flush=True)
