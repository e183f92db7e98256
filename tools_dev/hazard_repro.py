"""Development (round 5): a reproducer of the division hazard that needs no kernel of the library as the VICTIM —
tools_dev/hazard_micro.hip::tsa_div_victim, the TSA gather as it was before occ::fdiv, on a value map of ones (every interior
output must be 1) — next to (a) the library's value projection, (b) the synthetic MFMA + LDS-read spin kernel of the same file
(a fully self-contained pair), (c) MFMAs only, (d) nothing.  usage: python tools_dev/hazard_repro.py [reps]"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from occnet_amd import ext, synthetic                              # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
lib = ctypes.CDLL(os.path.join(ROOT, "tools_dev", "bin", "libhazard_micro.so"))
P = ctypes.c_void_p
st = lambda: P(torch.cuda.current_stream().cuda_stream)
g = torch.Generator().manual_seed(7)
bh = bw = 200
value = torch.ones(bh * bw * 256, device='cuda')
offs = (torch.randn(bh * bw, 128, generator=g) * 1.5).cuda()
logits = torch.randn(bh * bw, 64, generator=g).cuda()
sink = torch.zeros(4, device='cuda')
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
    "none": None,
    "occ value projection": lambda: [ext.value_proj_bf16_planes(map_rows, ws, gbs, planes, rows_per_group=hw, out_group_rows=total,
                                                                out_row0=starts) for _ in range(3)],
    "synthetic MFMA + LDS reads (same file)": lambda: lib.hz_spin(P(sink.data_ptr()), 1, 512, 3000, st()),
    "synthetic MFMA only (same file)": lambda: lib.hz_spin(P(sink.data_ptr()), 0, 512, 3000, st()),
}
load = torch.cuda.Stream()
for fn in loads.values():
    if fn is not None:
        with torch.cuda.stream(load):
            fn()
torch.cuda.synchronize()
err = torch.zeros(5, dtype=torch.int64, device='cuda')
VARIANTS = (("as it was: quotients by `/`", 0), ("quotients by occ::fdiv", 1), ("bare v_exp_f32", 2), ("no softmax (weight 1/4)", 4),
            ("offsets zero", 8), ("no softmax + offsets zero", 12), ("s_nop after the set-up", 16), ("block barrier hand-over", 32),
            ("fdiv + no softmax", 5), ("set-up without scalar lane masks", 64), ("... + fdiv", 65),
            ("... + no softmax + offsets zero", 76))
if os.environ.get("HZ_VARIANTS"):
    keep = set(int(v) for v in os.environ["HZ_VARIANTS"].split(","))
    VARIANTS = tuple(v for v in VARIANTS if v[1] in keep)
if os.environ.get("HZ_ONLY_VPROJ") == "1":
    loads = {k: v for k, v in loads.items() if k == "occ value projection"}
if os.environ.get("HZ_MASK") == "1":      # the suspected instruction pattern in isolation (hazard_micro.hip::mask_victim)
    lfn = loads["occ value projection"]
    for mode, mname in ((0, "v_cmp -> s_and_b64 -> v_cndmask (asm)"), (1, "v_cmp -> s_and_b64 -> s_and_saveexec_b64 (asm)"),
                        (2, "the same test in C")):
        for neighbour in (False, True):
            err.zero_()
            bad_reps = 0
            for rep in range(reps):
                before = int(err[0].item())
                if neighbour:
                    with torch.cuda.stream(load):
                        lfn()
                for _ in range(4):
                    lib.hz_mask_victim(P(err.data_ptr()), 4096, 512, mode, st())
                torch.cuda.synchronize()
                bad_reps += int(err[0].item()) > before
            e = err.tolist()
            print(f"REPRO neighbour = {'occ value projection' if neighbour else 'none':22s} victim = lane-mask micro-kernel, {mname:48s}: "
                  f"{e[0]:8d} wrong bits in {bad_reps} of {reps} repetitions; lanes 0-15 / 16-31 / 32-47 / 48-63: {e[1:]}", flush=True)
    sys.exit(0)
for lname, lfn in loads.items():
    for vname, use_fdiv in VARIANTS:
        err.zero_()
        bad_reps = 0
        for rep in range(reps):
            before = int(err[0].item())
            if lfn is not None:
                with torch.cuda.stream(load):
                    lfn()
            for _ in range(4):
                lib.hz_tsa_div_victim(P(value.data_ptr()), P(offs.data_ptr()), P(logits.data_ptr()), P(err.data_ptr()), bh, bw, use_fdiv, st())
            torch.cuda.synchronize()
            bad_reps += int(err[0].item()) > before
        e = err.tolist()
        print(f"REPRO neighbour = {lname:40s} victim = TSA gather on ones, {vname:34s}: {e[0]:6d} wrong words in {bad_reps} of {reps} "
              f"repetitions; lanes 0-15 / 16-31 / 32-47 / 48-63: {e[1:]}", flush=True)
