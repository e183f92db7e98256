"""not-gpu: the q16 value-row format's restatement (tests/q16_ref.py) — error bounds and the exponent-in-the-low-bits trick.
(The HIP encoders against this restatement and the gather against the oracle: tests/test_gpu_q16.py.)"""
import numpy as np

from tests import q16_ref


def test_q16_round_trip_error_bounds():
    rng = np.random.default_rng(3)
    for amp in (1e-6, 1e-3, 1.0, 37.0, 3e4):
        v = (rng.standard_normal((4000, 32)) * amp).astype(np.float32)
        v[::7, 3] *= 50                                       # outliers: one element sets its piece's exponent
        s = np.float32(2.0 ** (14 - int(np.floor(np.log2(np.abs(v).max())))))     # max|v| * s in [2^14, 2^15)
        q = q16_ref.encode(v, s)
        back = q16_ref.decode(q, s)
        m = np.abs(v.reshape(-1, 8)).max(1, keepdims=True).astype(np.float64)     # the piece's largest magnitude
        err = np.abs(back.reshape(-1, 8) - v.reshape(-1, 8).astype(np.float64))
        floor = 2.0 ** -15 / float(s)                          # pieces below 2^-15 of the plane's range: exponent 0
        # elements 2 .. 7: half a step of the piece's 15-bit grid; elements 0, 1 (they carry the exponent): two steps
        assert (err[:, 2:] <= np.maximum(m * 2.0 ** -15, floor * 0.5) * 1.0000001).all()
        assert (err[:, :2] <= np.maximum(m * 2.0 ** -13, floor * 2.0) * 1.0000001).all()
        # against fp16 rows of the same values: the piece's largest element is >= 8x closer
        big = np.abs(v.reshape(-1, 8)).argmax(1)
        keep = big >= 2
        e16 = np.abs((v * s).astype(np.float16).astype(np.float64) / float(s) - v.astype(np.float64)).reshape(-1, 8)
        rows = np.arange(len(big))[keep]
        assert err[rows, big[keep]].mean() * 8 < e16[rows, big[keep]].mean()


def test_q16_exponent_survives_in_the_low_bits_and_extremes():
    v = np.zeros((6, 8), np.float32)
    v[1] = 32768.0                                            # the top of the range: clamped to 32767, exponent 15
    v[2] = [-32768.0, 3.0, 1.0, 0.5, 0.25, -0.125, 7.0, -9.0]
    v[3] = [np.nan, np.inf, -np.inf, 1.0, 2.0, 3.0, 4.0, 5.0]
    v[4] = 1e-9                                               # far below the range: exponent 0, all zeros
    v[5] = [1.0, -1.0, 0.75, 0.5, 0.3, 0.2, 0.1, 0.0]
    q = q16_ref.encode(v, 1.0)
    E = (q[:, 0].astype(np.int64) & 3) | ((q[:, 1].astype(np.int64) & 3) << 2)
    assert list(E) == [0, 15, 15, 15, 0, 1]
    assert (np.abs(q.astype(np.int64)) <= 32768).all()
    back = q16_ref.decode(q, 1.0)
    assert (back[0] == 0).all() and np.abs(back[1] - 32767.0).max() <= 3.0
    assert back[3][0] == 0 or abs(back[3][0]) <= 3.0            # NaN -> 0 (the exponent bits may sit in it)
    assert abs(back[3][1] - 32767.0) <= 3.0 and abs(back[3][2] + 32767.0) <= 1.0
    assert np.abs(back[5] - v[5]).max() <= 2.0 ** -13
