"""not-gpu: the q16 value-row format's restatement (tests/q16_ref.py) — error bounds and the exponent-in-the-low-bits trick.
(The HIP encoders against this restatement and the gather against the oracle: tests/test_gpu_q16.py.)"""
import numpy as np
import pytest

from tests import q16_ref


@pytest.mark.parametrize("group", [8, 32])
def test_q16_round_trip_error_bounds(group):
    rng = np.random.default_rng(3)
    for amp in (1e-6, 1e-3, 1.0, 37.0, 3e4):
        v = (rng.standard_normal((4000, 32)) * amp).astype(np.float32)
        v[::7, 3] *= 50                                       # outliers: one element sets its group's exponent
        s = np.float32(2.0 ** (14 - int(np.floor(np.log2(np.abs(v).max())))))     # max|v| * s in [2^14, 2^15)
        q = q16_ref.encode(v, s, group=group)
        assert np.abs(q.astype(np.int64)).max() <= 32767
        back = q16_ref.decode(q, s)
        m = np.repeat(np.abs(v.reshape(-1, group)).max(1).astype(np.float64), group // 8)[:, None]   # the group's largest magnitude
        err = np.abs(back.reshape(-1, 8) - v.reshape(-1, 8).astype(np.float64))
        floor = 2.0 ** -15 / (float(s) * 0.5)                  # groups below 2^-15 of the stored range: exponent 0
        step = np.maximum(m * 1.0003 * 2.0 ** -14, floor)      # one step of the group's 15-bit grid (m < 2^E <= 2 m (1 + 2^-12))
        # elements 2 .. 7: half a step; elements 0, 1 (they carry the exponent in their low bits): two steps
        assert (err[:, 2:] <= step * 0.5 * 1.0000001).all()
        assert (err[:, :2] <= step * 2.0 * 1.0000001).all()
        # rounding is unbiased on both kinds of element
        signed = (back.reshape(-1, 8) - v.reshape(-1, 8).astype(np.float64)) / step
        assert abs(signed[:, 2:].mean()) < 0.01 and abs(signed[:, :2].mean()) < 0.05
    # against fp16 rows of the same values: the group's largest elements are much closer (it is what the gather's sums see)
    v = (rng.standard_normal((4000, 32)) * 37).astype(np.float32)
    s = np.float32(2.0 ** (14 - int(np.floor(np.log2(np.abs(v).max())))))
    e_q = np.abs(q16_ref.decode(q16_ref.encode(v, s, group=group), s) - v.astype(np.float64))
    e_h = np.abs((v * s).astype(np.float16).astype(np.float64) / float(s) - v.astype(np.float64))
    assert np.sqrt((e_q ** 2).mean()) * 3 < np.sqrt((e_h ** 2).mean())


def test_q16_exponent_survives_in_the_low_bits_and_extremes():
    v = np.zeros((6, 8), np.float32)
    v[1] = 32768.0                                            # the top of the range (scale 1): exponent 15, no overflow
    v[2] = [-32768.0, 3.0, 1.0, 0.5, 0.25, -0.125, 7.0, -9.0]
    v[3] = [1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0, 8.0]
    v[4] = 1e-9                                               # far below the range: exponent 0, decodes to ~0
    v[5] = [1.0, -1.0, 0.75, 0.5, 0.3, 0.2, 0.1, 0.0]
    q = q16_ref.encode(v, 1.0)
    E = (q[:, 0].astype(np.int64) & 3) | ((q[:, 1].astype(np.int64) & 3) << 2)
    assert list(E) == [0, 15, 15, 3, 0, 0]                    # stored under s / 2: |v| = 8 -> 4 < 2^3; |v| = 1 -> 0.5 < 2^0
    assert (np.abs(q.astype(np.int64)) <= 32767).all()
    back = q16_ref.decode(q, 1.0)
    assert (back[0] == 0).all() and np.abs(back[1] - 32768.0).max() <= 8.0 and abs(back[2][0] + 32768.0) <= 8.0
    assert np.abs(back[3] - v[3]).max() <= 2.0 ** -9 and np.abs(back[4]).max() <= 2.0 ** -13
    assert np.abs(back[5] - v[5]).max() <= 2.0 ** -12
