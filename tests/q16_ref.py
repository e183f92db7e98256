"""Test infrastructure: numpy restatement of the q16 value-row format (csrc/common.h: q16_clamp / q16_exponent / q16_quant /
q16_quant_tagged / fma8q) — block floating point, 8 int16 mantissas per 16-byte piece under one 4-bit exponent kept in the two
low bits of elements 0 and 1.  The HIP encoders (the value projection's epilogue, occ_sca_rows_encode_q16) must agree with
`encode` bit for bit; `decode` is what the gather's fma8q computes per element."""
import numpy as np


def encode(v, scale=1.0):
    """v (..., 8k) float32 -> int16 of the same shape (row order, NOT the pixel-pair layout), pieces of 8 along the last axis."""
    v = np.asarray(v, dtype=np.float32)
    shp = v.shape
    u = (v * np.float32(scale)).astype(np.float32).reshape(-1, 8)
    u = np.where(np.isnan(u), np.float32(0), u)
    u = np.clip(u, np.float32(-32768), np.float32(32768))
    m = np.abs(u).max(axis=1)
    _, x = np.frexp(m)                                   # m = f 2^x, f in [0.5, 1); frexp(0) = (0, 0)
    E = np.clip(np.where(m > 0, x, 0), 0, 15).astype(np.int32)
    y = np.ldexp(u, (15 - E)[:, None]).astype(np.float32)          # exact: a power-of-two multiple
    qf = np.rint(y).astype(np.int32)                     # round half to even, like v_rndne_f32
    q = np.clip(qf, -32767, 32767)
    for j, r in ((0, E & 3), (1, E >> 2)):               # elements 0, 1: the nearest integer = r (mod 4)
        d = (qf[:, j] - r) & 3
        up = y[:, j] >= qf[:, j].astype(np.float32)
        t = np.where(d == 0, qf[:, j], np.where(d == 1, qf[:, j] - 1, np.where(d == 3, qf[:, j] + 1,
                                                                                 np.where(up, qf[:, j] + 2, qf[:, j] - 2))))
        t = np.where(t > 32767, t - 4, np.where(t < -32768, t + 4, t))
        q[:, j] = t
    return q.astype(np.int16).reshape(shp)


def decode(q, scale=1.0):
    """int16 (..., 8k) -> float64 values: q_j 2^(E - 15) / scale with E read from the low bits of elements 0 and 1."""
    q = np.asarray(q, dtype=np.int16)
    shp = q.shape
    p = q.reshape(-1, 8).astype(np.int64)
    E = (p[:, 0] & 3) | ((p[:, 1] & 3) << 2)
    return (p.astype(np.float64) * np.ldexp(1.0, E - 15)[:, None] / float(scale)).reshape(shp)


def pair_layout(rows):
    """(BN, S, M, 32) row-ordered elements -> the gather's pixel-pair order (BN, S + (S & 1), M, 32), zero pad row."""
    BN, S, M, D = rows.shape
    if S & 1:
        rows = np.concatenate([rows, np.zeros((BN, 1, M, D), rows.dtype)], 1)
    return np.ascontiguousarray(rows.reshape(BN, -1, 2, M, D).transpose(0, 1, 3, 2, 4)).reshape(BN, -1, M, D)
