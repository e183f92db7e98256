"""-m gpu, always on: the co-scheduling hazard of round 5 as a STANDING reproducer.

Round 5 ran a wrong-rows symptom down to this: a gather kernel whose sampling set-up keeps its conditions as lane masks in
scalar registers (hipcc's translation of plain `bool` code: v_cmp -> s_and_b64 / s_and_saveexec_b64) returns wrong weights in
lanes 48-63 now and then while a wave of one of this library's MFMA kernels is resident on another hardware queue — 149 of
150 repetitions for a self-contained copy of the TSA gather next to the stacked value projection.  With the conditions kept
as 0 / 1 integers in VGPRs (common.h: lane_flag / bilinear_terms) 0 of 150, outputs bit-identical.  Since round 6 every
gather / scatter kernel of the library (SCA f32 / f16 rows, TSA, generic forward, all backward passes) is built on that
set-up.  This file keeps the evidence alive:
  1. tests/hazard/victim.hip includes common.h AS SHIPPED: the victim with the shipped set-up must be clean 150 / 150 next to
     the value projection; the same victim with the legacy set-up (kept verbatim in that file) is run as the CONTROL and its
     failure count printed (not asserted: it is the hardware's behaviour, not ours);
  2. the library's own TSA kernel on a map of ones, next to the value projection and chain program A on another stream;
  3. a training step's backward (OCC_MSDA_BWD_DETERMINISTIC=1: order-independent accumulation) bit-identical under that load.
The operator contract this protects: the reference's op runs on whatever stream the caller is on
(projects/mmdet3d_plugin/bevformer/modules/multi_scale_deformable_attn_function.py:118-124), next to whatever else the
process has in flight (DDP's communication stream: projects/mmdet3d_plugin/bevformer/apis/mmdet_train.py:71-79)."""
import ctypes
import os
import subprocess

import pytest
import torch

from occnet_amd import synthetic

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "hazard", "victim.hip")
SO = os.path.join(HERE, "hazard", "_build", "libhazard_victim.so")


def build_victim(force=False):
    """hipcc --offload-arch=gfx950 (also called by __graft_entry__.build(): the .so travels to the GPU box)."""
    hdr = os.path.join(os.path.dirname(HERE), "occnet_amd", "csrc", "common.h")
    if not force and os.path.exists(SO) and os.path.getmtime(SO) >= max(os.path.getmtime(SRC), os.path.getmtime(hdr)):
        return SO
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    hipcc = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else "hipcc"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                    "-Wno-unused-command-line-argument", "-o", SO, SRC], check=True)
    return SO


class MfmaNeighbour:
    """The library's own MFMA kernels on buffers of their own, issued on a second stream: the stacked value projection (the
    kernel the hazard was met next to) and chain program A."""

    def __init__(self):
        from occnet_amd import ext
        self.ext = ext
        g = torch.Generator().manual_seed(7)
        feats = synthetic.make_features(dict(synthetic.BASE), seed=12)
        maps = [f.reshape(-1, 256, f.shape[3], f.shape[4]).cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
                for f in feats]
        self.rows = [m.permute(0, 2, 3, 1).reshape(-1, 256) for m in maps]
        self.hw = [m.shape[2] * m.shape[3] for m in maps]
        self.starts = [sum(self.hw[:i]) for i in range(len(self.hw))]
        self.total = sum(self.hw) + (sum(self.hw) & 1)
        self.ws = [((torch.rand(256, 256, generator=g) * 2 - 1) * 0.1).cuda() for _ in range(4)]
        self.gbs = [torch.randn(4, 6, 256, generator=g).cuda() for _ in range(4)]
        self.planes = torch.empty(4, 6 * self.total, 256, dtype=torch.float16, device='cuda')
        self.ca = dict(attn=torch.randn(1, 40000, 256, device='cuda'), q=torch.randn(1, 40000, 256, device='cuda'),
                       w1=((torch.rand(256, 256, generator=g) * 2 - 1) * 0.06).cuda(), b1=torch.zeros(256, device='cuda'),
                       ln=torch.nn.LayerNorm(256).cuda(), w2=((torch.rand(768, 256, generator=g) * 2 - 1) * 0.06).cuda(),
                       b2=torch.zeros(768, device='cuda'))
        self.stream = torch.cuda.Stream()
        self.issue(1)
        torch.cuda.synchronize()

    def issue(self, n=3, chain=False):
        with torch.cuda.stream(self.stream):
            for _ in range(n):
                self.ext.value_proj_bf16_planes(self.rows, self.ws, self.gbs, self.planes, rows_per_group=self.hw,
                                                out_group_rows=self.total, out_row0=self.starts)
                if chain:
                    c = self.ca
                    self.ext.linear_ln_chain(c['attn'], c['q'], c['w1'], c['b1'], c['ln'], c['w2'], c['b2'])


@pytest.fixture(scope="module")
def neighbour():
    return MfmaNeighbour()


def _victim_inputs(bh, bw):
    g = torch.Generator().manual_seed(7)
    value = torch.ones(bh * bw * 256, device='cuda')
    offs = (torch.randn(bh * bw, 128, generator=g) * 1.5).cuda()
    logits = torch.randn(bh * bw, 64, generator=g).cuda()
    return value, offs, logits


def test_shipped_setup_is_clean_next_to_the_value_projection(neighbour):
    lib = ctypes.CDLL(build_victim())
    P = ctypes.c_void_p
    bh = bw = 200
    value, offs, logits = _victim_inputs(bh, bw)
    err = torch.zeros(5, dtype=torch.int64, device='cuda')

    def run(shipped, reps):
        err.zero_()
        bad_reps = 0
        for _ in range(reps):
            before = int(err[0].item())
            neighbour.issue(3)
            st = P(torch.cuda.current_stream().cuda_stream)
            for _ in range(4):
                rc = lib.hz_tsa_victim(P(value.data_ptr()), P(offs.data_ptr()), P(logits.data_ptr()), P(err.data_ptr()), bh, bw,
                                       int(shipped), st)
                assert rc == 0
            torch.cuda.synchronize()
            bad_reps += int(err[0].item()) > before
        return bad_reps, err.tolist()

    # quiet chip first: both set-ups must be exact (a failure here would be a bug of the victim, not the hazard)
    torch.cuda.synchronize()
    for shipped in (1, 0):
        err.zero_()
        lib.hz_tsa_victim(P(value.data_ptr()), P(offs.data_ptr()), P(logits.data_ptr()), P(err.data_ptr()), bh, bw, shipped,
                          P(torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        assert int(err[0].item()) == 0, f"victim (shipped={shipped}) wrong on a quiet chip"
    ctl_reps, ctl = run(False, 50)
    print(f"HAZARD control (legacy scalar-lane-mask set-up, `/`): wrong in {ctl_reps} of 50 repetitions, {ctl[0]} words; "
          f"lanes 0-15 / 16-31 / 32-47 / 48-63: {ctl[1:]}")
    bad_reps, e = run(True, 150)
    print(f"HAZARD shipped set-up (common.h lane_flag / bilinear_terms, occ::fdiv): wrong in {bad_reps} of 150 repetitions, "
          f"{e[0]} words; lanes by quarter: {e[1:]}")
    assert bad_reps == 0 and e[0] == 0


def test_library_tsa_kernel_on_ones_next_to_mfma_kernels(neighbour):
    """The shipped TSA gather itself: on value maps of ones every interior query's output is 1 (weights sum to one)."""
    from occnet_amd import ext
    g = torch.Generator().manual_seed(11)
    bh = bw = 200
    M, D, P = 8, 32, 4
    Nq = bh * bw
    value = torch.ones(1, Nq, M, D, device='cuda')
    offs = (torch.randn(1, Nq, M * 2 * P * 2, generator=g) * 1.5).cuda()
    logits = torch.randn(1, Nq, M * 2 * P, generator=g).cuda()
    ys, xs = torch.meshgrid(torch.arange(bh), torch.arange(bw), indexing='ij')
    ref = torch.stack([(xs.flatten() + 0.5) / bw, (ys.flatten() + 0.5) / bh], -1)
    ref = ref[None, :, None, :].expand(2, Nq, 1, 2).contiguous().cuda()
    interior = ((ys >= 16) & (ys < bh - 16) & (xs >= 16) & (xs < bw - 16)).flatten().cuda()
    bad = 0
    for rep in range(100):
        neighbour.issue(2, chain=True)
        out = ext.tsa_fused_forward(value, offs, logits, ref, bh, bw, M, P, shared_queue=True)
        torch.cuda.synchronize()
        d = float((out[0][interior] - 1.0).abs().max())
        bad += d > 1e-5
    print(f"library TSA kernel on ones next to value projection + chain A: {bad} of 100 repetitions off by > 1e-5")
    assert bad == 0


def test_training_backward_is_bit_identical_under_mfma_load(neighbour, monkeypatch):
    """Forward + backward through the autograd path (MultiScaleDeformableAttnFunction: HIP forward and the binned backward with
    order-independent accumulation) with the library's MFMA kernels running on another stream: every parameter gradient
    bit-identical to the solo run."""
    from occnet_amd.train import synthetic_targets
    from tests.util import build_pair, small_cfg
    monkeypatch.setenv("OCC_MSDA_BWD_DETERMINISTIC", "1")
    g = small_cfg(bev=(40, 40), num_layers=2)
    prod, _ = build_pair(g, seed=3)
    feats = [f.cuda() for f in synthetic.make_features(g, seed=3)]
    metas = synthetic.make_img_metas(g)
    sem, flow, mask = synthetic_targets(g['bev_h'], g['bev_w'], g['pillar_h'], num_classes=17, batch=1, seed=0, device='cuda')

    def grads():
        prod.zero_grad(set_to_none=True)
        out = prod(feats, metas)
        lp = prod.loss(sem, flow, mask, out)
        (lp['loss_occ'] + lp['loss_flow']).backward()
        torch.cuda.synchronize()
        return {n: p.grad.clone() for n, p in prod.named_parameters() if p.grad is not None}

    solo = grads()
    again = grads()
    unstable = [n for n in solo if not torch.equal(solo[n], again[n])]
    if unstable:
        pytest.skip(f"solo backward not bit-reproducible on this stack ({len(unstable)} tensors, e.g. {unstable[0]}): "
                    f"cannot tell a hazard from reordering")
    bad = []
    for rep in range(20):
        neighbour.issue(4, chain=True)
        got = grads()
        for n in solo:
            if not torch.equal(got[n], solo[n]):
                bad.append((rep, n, float((got[n] - solo[n]).abs().max())))
    print(f"training backward under MFMA load: {len(solo)} gradients x 20 repetitions, {len(bad)} differ {bad[:4]}")
    assert not bad
