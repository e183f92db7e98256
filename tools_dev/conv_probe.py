"""Dev probe: per-shape timing of the backbone convolution kernels on the ResNet-50 / FPN shapes of the base
config (6 cameras, 928 x 1600 padded input)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from occnet_amd import ext


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


def main(which):
    g = torch.Generator().manual_seed(0)
    N = 6
    mk = lambda *s: (torch.randn(*s, generator=g) * 0.05).cuda()
    act = lambda c, h, w: torch.randn(N, c, h, w, generator=g).cuda().to(torch.bfloat16).contiguous(
        memory_format=torch.channels_last)
    if 'c3' in which:
        for (cin, cout, h, w, s) in [(128, 128, 232, 400, 2), (128, 128, 116, 200, 1), (256, 256, 116, 200, 2),
                                     (256, 256, 58, 100, 1), (512, 512, 58, 100, 2), (512, 512, 29, 50, 1),
                                     (256, 256, 116, 200, 1), (256, 256, 29, 50, 2)]:
            x, wp, b = act(cin, h, w), ext.conv3x3_pack_weight(mk(cout, cin, 3, 3)), mk(cout)
            us = timeit(lambda: ext.conv3x3_nhwc(x, wp, b, cout, relu=True, stride=s))
            ho, wo = (h - 1) // s + 1, (w - 1) // s + 1
            fl = 2.0 * N * ho * wo * cout * cin * 9
            print(f"conv3x3 {cin:4d}->{cout:4d} {h:3d}x{w:3d} s{s}: {us:7.1f} us  {fl / us * 1e-6:6.1f} TF/s")
    if 'c1' in which:
        for (cin, cout, h, w, s, res) in [(256, 128, 232, 400, 1, 0), (128, 512, 116, 200, 1, 1),
                                          (512, 128, 116, 200, 1, 0), (256, 512, 232, 400, 2, 0),
                                          (512, 256, 116, 200, 1, 0), (256, 1024, 58, 100, 1, 1),
                                          (1024, 256, 58, 100, 1, 0), (512, 1024, 116, 200, 2, 0),
                                          (1024, 512, 58, 100, 1, 0), (512, 2048, 29, 50, 1, 1),
                                          (2048, 512, 29, 50, 1, 0), (1024, 2048, 58, 100, 2, 0),
                                          (512, 256, 116, 200, 1, 0), (1024, 256, 58, 100, 1, 0),
                                          (2048, 256, 29, 50, 1, 0)]:
            x, wp, b = act(cin, h, w), ext.conv1x1_pack_weight(mk(cout, cin)), mk(cout)
            ho, wo = (h - 1) // s + 1, (w - 1) // s + 1
            r = act(cout, ho, wo) if res else None
            us = timeit(lambda: ext.conv1x1_nhwc(x, wp, b, residual=r, relu=True, stride=s))
            fl = 2.0 * N * ho * wo * cout * cin
            mb = N * (ho * wo * (cin + cout * (2 if res else 1))) * 2 / 1e6
            print(f"conv1x1 {cin:4d}->{cout:4d} {h:3d}x{w:3d} s{s} res{res}: {us:7.1f} us  {fl / us * 1e-6:6.1f} TF/s  "
                  f"{mb / us:5.2f} TB/s")


def vp():
    g = torch.Generator().manual_seed(0)
    w = (torch.randn(256, 256, generator=g) / 16).cuda()
    gb = torch.randn(6, 256, generator=g).cuda()
    total = 23200 + 5800 + 1450 + 375
    out = torch.empty(6 * total, 256, device='cuda')
    hws = (23200, 5800, 1450, 375)
    a_list = [torch.randn(6 * hw, 256, generator=g).cuda().to(torch.bfloat16) for hw in hws]
    gb4 = gb.unsqueeze(0).repeat(4, 1, 1).contiguous()
    starts = [0, 23200, 29000, 30450]
    us = timeit(lambda: ext.value_proj_bf16(a_list, w, gb4, out, rows_per_group=list(hws), out_group_rows=total,
                                            out_row0=starts))
    mb = 6 * total * 256 * (2 + 4) / 1e6
    print(f"value_proj_bf16 all levels M={6 * total}: {us:7.1f} us  {mb / us:5.2f} TB/s")
    x = torch.randn(6 * total, 256, generator=g).cuda()
    b = torch.randn(256, generator=g).cuda()
    us = timeit(lambda: ext.linear(x, w, b))
    print(f"linear bf16x3 f32 in M={6 * total}: {us:7.1f} us")


if __name__ == '__main__':
    if 'vp' in sys.argv[1:]:
        vp()
    main(sys.argv[1:] or ['c3', 'c1'])
