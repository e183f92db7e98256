#!/usr/bin/env python
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE'S OWN module files
(from /root/reference, under the stubs of oracle/refshim.py) on seeded synthetic inputs — TEST
INFRASTRUCTURE.  Run in the build container only (the GPU box has no /root/reference):

    python -m oracle.gen_golden

For each case: weights = tests.util.randomize(seed) of the shared state_dict layout, inputs =
occnet_amd.synthetic (seeded), outputs of the reference head / of its individual attention modules
are stored as float32 .npz together with checksums of the seeded weights and inputs (so a change of
the RNG stream is detected instead of silently comparing different problems).
"""
import copy
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, 'tests', 'golden')

from tests.golden_cases import CASES, case_inputs, checksum  # noqa: E402


def main():
    from oracle import refshim
    ref = refshim.install()
    print('reference files executed:')
    for f in ref.files:
        print('  ', f)
    from tests.util import head_cfg, randomize
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(os.cpu_count() or 1)
    for name, case in CASES.items():
        g = case['geometry']
        cfg = head_cfg(g)
        head = ref.build_head(copy.deepcopy(cfg))
        assert type(head).__module__.startswith('projects.mmdet3d_plugin'), type(head).__module__
        randomize(head, case['seed'])
        head.eval()
        feats, metas, prev_bev = case_inputs(case)
        taps = {}

        def tap(label):
            def hook(mod, args, out):
                taps[label] = out.detach().clone()
            return hook
        layer0 = head.transformer.encoder.layers[0]
        hs = [layer0.attentions[0].register_forward_hook(tap('layer0_tsa_out')),
              layer0.attentions[1].register_forward_hook(tap('layer0_sca_out'))]
        with torch.no_grad():
            out = head(feats, metas, prev_bev=None if prev_bev is None else prev_bev.clone())
        for h in hs:
            h.remove()
        arrays = dict(
            bev_embed=out['bev_embed'].numpy(), occ=out['occ'].numpy(), flow=out['flow'].numpy(),
            weights_checksum=np.float64(checksum(head.state_dict().values())),
            inputs_checksum=np.float64(checksum(feats)),
            **{k: v.numpy() for k, v in taps.items()})
        path = os.path.join(OUT, f'{name}.npz')
        np.savez_compressed(path, **arrays)
        print(f'{name}: wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB); '
              f'occ {arrays["occ"].shape} mean|occ| {np.abs(arrays["occ"]).mean():.4f}')


if __name__ == '__main__':
    main()
