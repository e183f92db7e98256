"""Development: the micro-victims of tools_dev/hazard_micro.hip next to the library's value-projection
kernel on another stream — which path (LDS slab hand-off, buffer-load row gather, plain global loads) returns wrong data?
usage: python tools_dev/hazard_micro.py [reps]"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from occnet_amd import ext, synthetic                              # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
lib = ctypes.CDLL(os.path.join(ROOT, "tools_dev", "bin", "libhazard_micro.so"))
g = dict(synthetic.BASE)
feats = synthetic.make_features(g, seed=12)


def nhwc(f):
    B, N, C, h, w = f.shape
    return f.reshape(B * N, C, h, w).cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last).view(B, N, C, h, w)


x = [nhwc(f) for f in feats]
map_rows = [f.permute(0, 1, 3, 4, 2).reshape(-1, f.shape[2]) for f in x]
hw = [f.shape[3] * f.shape[4] for f in x]
starts = [sum(hw[:i]) for i in range(len(hw))]
total = sum(hw) + (sum(hw) & 1)
gl = torch.Generator().manual_seed(3)
ws = [((torch.rand(256, 256, generator=gl) * 2 - 1) * 0.1).cuda() for _ in range(4)]
gbs = [torch.randn(4, 6, 256, generator=gl).cuda() for _ in range(4)]
planes = torch.empty(4, 6 * total, 256, dtype=torch.float16, device='cuda')
n_words = 16 << 20
table = torch.empty(n_words, dtype=torch.int32, device='cuda')
P = ctypes.c_void_p
st = lambda: P(torch.cuda.current_stream().cuda_stream)
lib.hz_fill(P(table.data_ptr()), ctypes.c_long(n_words), st())
torch.cuda.synchronize()
load = torch.cuda.Stream()
a = torch.randn(4096, 4096, device='cuda', dtype=torch.bfloat16)
b = torch.randn(4096, 4096, device='cuda', dtype=torch.bfloat16)


def vproj_load():
    ext.value_proj_bf16_planes(map_rows, ws, gbs, planes, rows_per_group=hw, out_group_rows=total, out_row0=starts)


def gemm_load():
    for _ in range(4):
        a @ b


with torch.cuda.stream(load):
    vproj_load()
    gemm_load()
torch.cuda.synchronize()
err = torch.zeros(1, dtype=torch.int64, device='cuda')
ca = dict(attn=torch.randn(1, 40000, 256, device='cuda'), q=torch.randn(1, 40000, 256, device='cuda'),
          w1=((torch.rand(256, 256, generator=gl) * 2 - 1) * 0.06).cuda(), b1=torch.zeros(256, device='cuda'),
          ln=torch.nn.LayerNorm(256).cuda(), w2=((torch.rand(768, 256, generator=gl) * 2 - 1) * 0.06).cuda(),
          b2=torch.zeros(768, device='cuda'))


def chain_load():
    for _ in range(3):
        ext.linear_ln_chain(ca['attn'], ca['q'], ca['w1'], ca['b1'], ca['ln'], ca['w2'], ca['b2'])


with torch.cuda.stream(load):
    chain_load()
torch.cuda.synchronize()
valu = {
    "VALU packed fp32 FMA (v_pk_fma_f32)": lambda: lib.hz_valu_victim(P(err.data_ptr()), 8192, 64, 0, st()),
    "VALU scalar v_fma_f32": lambda: lib.hz_valu_victim(P(err.data_ptr()), 8192, 64, 1, st()),
    "VALU 32-bit integer multiply-add": lambda: lib.hz_valu_victim(P(err.data_ptr()), 8192, 64, 2, st()),
    "VALU v_exp_f32 / v_rcp_f32": lambda: lib.hz_valu_victim(P(err.data_ptr()), 8192, 256, 3, st()),
    "VALU IEEE fp32 division (v_div_scale/fmas/fixup)": lambda: lib.hz_valu_victim(P(err.data_ptr()), 8192, 256, 5, st()),
    "gather set-up: v_div_fixup -> v_pk_add_f32 -> v_pk_fma_f32": lambda: lib.hz_valu_victim(P(err.data_ptr()), 8192, 256, 6, st()),
    "VALU v_fma_mix_f32": lambda: lib.hz_valu_victim(P(err.data_ptr()), 8192, 64, 4, st()),
}
victims = {
    "lds slab, wave_lds_sync (no waitcnt)": lambda: lib.hz_lds_victim(P(err.data_ptr()), 4096, 64, 0, st()),
    "lds slab, with s_waitcnt lgkmcnt(0)": lambda: lib.hz_lds_victim(P(err.data_ptr()), 4096, 64, 1, st()),
    "buffer-load row gather (TA path)": lambda: lib.hz_ta_victim(P(table.data_ptr()), ctypes.c_uint32(n_words // 32), P(err.data_ptr()), 4096, 64, st()),
    "ds_bpermute exchange (__shfl_xor 1,2,4,32)": lambda: lib.hz_xlane_victim(P(err.data_ptr()), 4096, 256, 0, st()),
    "DPP quad_perm exchange (xor 1,2)": lambda: lib.hz_xlane_victim(P(err.data_ptr()), 4096, 256, 1, st()),
    "plain global dword / dwordx2 loads": lambda: lib.hz_gl_victim(P(table.data_ptr()), ctypes.c_uint32(n_words), P(err.data_ptr()), 4096, 64, st()),
}
epoch = [1]
err2 = torch.zeros(2, dtype=torch.int64, device='cuda')
n16 = (30 << 20) // 16                 # 30 MB: the size of the TSA's zq rows


def pc(bs, bl):
    def f():
        epoch[0] += 1
        lib.hz_produce_consume(P(table.data_ptr()), ctypes.c_uint32(n16), ctypes.c_uint32(epoch[0]), P(err2.data_ptr()), bs, bl, st())
    return f


for lname, lfn in (() if os.environ.get("HZ_SKIP_PC") == "1" else (("no load", None), ("hipBLASLt GEMM on another stream", gemm_load), ("occ value projection on another stream", None))):
    if lname.startswith("occ"):
        lfn = vproj_load
    for vname, vfn in (("producer buffer stores -> consumer dword loads", pc(1, 0)), ("producer buffer stores -> consumer buffer loads", pc(1, 1)),
                       ("producer global stores -> consumer dword loads", pc(0, 0))):
        err2.zero_()
        bad_reps = 0
        for rep in range(reps):
            before = int(err2[0].item())
            if lfn is not None:
                with torch.cuda.stream(load):
                    for _ in range(2):
                        lfn()
            for _ in range(4):
                vfn()
            torch.cuda.synchronize()
            bad_reps += int(err2[0].item()) > before
        print(f"MICRO load = {lname:42s} pair = {vname:48s}: {int(err2[0].item()):8d} wrong words ({int(err2[1].item())} = last epoch's), "
              f"{bad_reps} of {reps} repetitions", flush=True)
victims = dict(valu, **victims)
for lname, lfn in (("no load", None), ("hipBLASLt GEMM on another stream", gemm_load), ("occ value projection on another stream", vproj_load),
                   ("occ chain program A on another stream", chain_load)):
    for vname, vfn in victims.items():
        err.zero_()
        bad_reps = 0
        for rep in range(reps):
            before = int(err.item())
            if lfn is not None:
                with torch.cuda.stream(load):
                    for _ in range(2):
                        lfn()
            for _ in range(4):
                vfn()
            torch.cuda.synchronize()
            bad_reps += int(err.item()) > before
        print(f"MICRO load = {lname:42s} victim = {vname:40s}: {int(err.item()):8d} wrong words, {bad_reps} of {reps} repetitions", flush=True)
