"""not-gpu: the CPU oracle (oracle/model.py) against the golden vectors produced by running the
reference's own module files (oracle/gen_golden.py).  This is what pins the oracle."""
import copy
import os

import numpy as np
import pytest
import torch

from tests.golden_cases import CASES, case_inputs, checksum
from tests.util import head_cfg, randomize

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _oracle_head(case):
    import oracle.model as om
    cfg = head_cfg(case['geometry'])
    cfg.pop('type')
    head = om.BEVFormerOccHead(**copy.deepcopy(cfg))
    randomize(head, case['seed'])
    return head.eval()


@pytest.mark.parametrize('name', sorted(CASES))
def test_oracle_matches_reference_golden(name):
    case = CASES[name]
    gold = np.load(os.path.join(GOLD, f'{name}.npz'))
    head = _oracle_head(case)
    feats, metas, prev_bev = case_inputs(case)
    # same seeded problem as the generator saw (guards against RNG-stream drift)
    assert abs(checksum(head.state_dict().values()) - float(gold['weights_checksum'])) < 1e-3
    assert abs(checksum(feats) - float(gold['inputs_checksum'])) < 1e-3
    taps = {}
    layer0 = head.transformer.encoder.layers[0]
    layer0.attentions[0].register_forward_hook(lambda m, a, o: taps.__setitem__('layer0_tsa_out', o))
    layer0.attentions[1].register_forward_hook(lambda m, a, o: taps.__setitem__('layer0_sca_out', o))
    with torch.no_grad():
        out = head(feats, metas, prev_bev=prev_bev)
    for k in ('bev_embed', 'occ', 'flow'):
        d = float(np.abs(out[k].numpy() - gold[k]).max())
        assert out[k].shape == gold[k].shape
        assert d < 1e-5, f'{name}/{k}: oracle differs from the reference golden by {d}'
    for k, v in taps.items():
        d = float(np.abs(v.numpy() - gold[k]).max())
        assert d < 1e-5, f'{name}/{k}: {d}'


# ---- the benchmarked geometry (BASELINE.json configs[1] / configs[2]) --------------------------------------------
from tests.golden_cases import FULL_CASES, FULL_KEYS, compare_digest, full_case_inputs  # noqa: E402


@pytest.mark.parametrize('name', sorted(FULL_CASES))
def test_oracle_matches_reference_golden_at_base_geometry(name):
    """oracle/model.py == the reference's own files at 40 000 queries / 6 x 30 825 keys / max_len ~ 9 900, one
    layer, without and with a rotated history BEV (fixtures: oracle/gen_golden.py::fullsize_golden)."""
    case = FULL_CASES[name]
    gold = np.load(os.path.join(GOLD, f'{name}.npz'))
    head = _oracle_head(case)
    feats, metas, prev_bev = full_case_inputs(case)
    assert abs(checksum(head.state_dict().values()) / float(gold['weights_checksum']) - 1) < 1e-9
    assert abs(checksum(feats) / float(gold['inputs_checksum']) - 1) < 1e-9
    taps = {}
    layer0 = head.transformer.encoder.layers[0]
    layer0.attentions[0].register_forward_hook(lambda m, a, o: taps.__setitem__('layer0_tsa_out', o))
    layer0.attentions[1].register_forward_hook(lambda m, a, o: taps.__setitem__('layer0_sca_out', o))
    with torch.no_grad():
        out = dict(head(feats, metas, prev_bev=prev_bev))
    out.update(taps)
    for k in FULL_KEYS:
        assert tuple(out[k].shape) == tuple(int(v) for v in gold[f'{k}_shape'])
        sub, slab = compare_digest(k, out[k], gold, 2e-5)
        print(f'{name}/{k}: subsample max diff {sub:.2e}, slab mean diff {slab:.2e}')
