"""Group a rocpd_summary.py kernel table (profiles/*train_kernel_trace*.txt) into the categories DESIGN.md §9 quotes.

    python tools_dev/train_trace_categories.py <summary.txt> <steps in the window>
"""
import collections
import re
import sys


def cat(k):
    if 'msda_bwd' in k:
        return 'occ msda backward'
    if 'linear_wgrad' in k:
        return 'occ linear wgrad'
    if 'linear_bf16x3' in k or 'linear_pack' in k:
        return 'occ linear fwd/dx'
    if 'occ::' in k:
        return 'occ other'
    if 'batch_norm' in k or 'BatchNorm' in k:
        return 'batch norm (torch/MIOpen)'
    if k.startswith('Cijk'):
        return 'GEMM (hipBLASLt/Tensile)'
    if any(t in k for t in ('igemm', 'ck::', 'Im3d2Col', 'Col2Im', 'naive_conv', 'Conv', 'conv')):
        return 'convolution (MIOpen/CK)'
    if 'copyBuffer' in k or 'direct_copy' in k:
        return 'copies'
    if 'at::native' in k or 'SubTensorOp' in k:
        return 'torch elementwise/reduce/index'
    return 'other'


def main(path, steps):
    rows = []
    head = []
    for line in open(path):
        m = re.match(r'\s*([\d.]+)\s+([\d.]+)\s+(\d+)\s+[\d.]+\s+[\d.]+\s+[\d.]+\s+(.*)', line)
        if m:
            rows.append((float(m.group(1)), int(m.group(3)), m.group(4)))
        elif line.startswith('#'):
            head.append(line.rstrip())
    print('\n'.join(head))
    tot = sum(r[0] for r in rows)
    print(f'# listed kernels: {len(rows)} rows, {tot:.1f} ms, {sum(r[1] for r in rows)} dispatches; {steps} steps in the window')
    c = collections.defaultdict(lambda: [0.0, 0])
    for ms, n, k in rows:
        c[cat(k)][0] += ms
        c[cat(k)][1] += n
    for k, (ms, n) in sorted(c.items(), key=lambda x: -x[1][0]):
        print(f'{k:34s} {ms:8.1f} ms  {ms / steps:6.1f} ms/step  {n / steps:7.0f} launches/step')
    for ms, n, k in rows:
        if 'occ::' in k:
            print(f'  {ms / steps:7.2f} ms/step {n / steps:5.0f}x  {k[:90]}')


if __name__ == '__main__':
    main(sys.argv[1], float(sys.argv[2]))
