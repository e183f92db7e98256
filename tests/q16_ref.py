"""Test infrastructure: numpy restatement of the q16 value-row format (csrc/common.h: q16_group_exponent / q16_pair /
q16_pair_tagged / fma8q) — block floating point, 8 int16 mantissas per 16-byte piece under one 4-bit exponent kept in the two
low bits of elements 0 and 1.  The HIP encoders (occ_sca_rows_encode_q16: one exponent per piece, group = 8; the value
projection's epilogue: one exponent per 64-byte head row, group = 32) must agree with `encode` bit for bit on finite inputs;
`decode` is what the gather's fma8q computes per element."""
import numpy as np

MAGIC = np.float32(12582912.0)           # 1.5 * 2^23


def _rne_low16(y32):
    """low 16 bits of rne(y) as the device takes them: the mantissa bits of y + 1.5 * 2^23 (float32 add)."""
    t = (y32.astype(np.float32) + MAGIC).astype(np.float32)
    return t.view(np.uint32) & np.uint32(0xffff)


def encode(v, scale=1.0, group=8):
    """v (..., C) float32, C % group == 0, group in (8, 32) -> int16 of the same shape (row order, NOT the pixel-pair layout).
    scale: the plane's range scale s (a power of two, max|v| * s <= 2^15); the rows are stored under s / 2."""
    v = np.asarray(v, dtype=np.float32)
    shp = v.shape
    eosc = int(np.frexp(np.float32(scale))[1]) - 2                       # log2(s / 2)
    g = v.reshape(-1, group)
    m = np.abs(g).max(axis=1).astype(np.float32)
    mp = (m * np.float32(1.000244140625)).astype(np.float32)
    x = np.frexp(mp)[1].astype(np.int64) + eosc
    E = np.where(m > 0, np.clip(x, 0, 15), 0).astype(np.int64)
    f = np.ldexp(np.float32(1.0), (eosc + 15 - E)).astype(np.float32)    # per group
    E8 = np.repeat(E, group // 8)
    f8 = np.repeat(f, group // 8)
    p = v.reshape(-1, 8)
    out = np.empty(p.shape, np.uint32)
    out[:, 2:] = _rne_low16((p[:, 2:] * f8[:, None]).astype(np.float32))          # v * f is exact (a power of two)
    for j, r in ((0, E8 & 3), (1, E8 >> 2)):
        z = (p[:, j].astype(np.float64) * (f8.astype(np.float64) * 0.25) - 0.25 * r).astype(np.float32)   # one rounding = fmaf
        out[:, j] = ((_rne_low16(z).astype(np.uint32) << np.uint32(2)) + r.astype(np.uint32)) & np.uint32(0xffff)
    return out.astype(np.uint16).view(np.int16).reshape(shp)


def decode(q, scale=1.0):
    """int16 (..., 8k) -> float64 values: q_j 2^(E - 15) / (scale / 2) with E read from the low bits of elements 0 and 1."""
    q = np.asarray(q, dtype=np.int16)
    shp = q.shape
    p = q.reshape(-1, 8).astype(np.int64)
    E = (p[:, 0] & 3) | ((p[:, 1] & 3) << 2)
    return (p.astype(np.float64) * np.ldexp(1.0, E - 15)[:, None] / (float(scale) * 0.5)).reshape(shp)


def pair_layout(rows):
    """(BN, S, M, 32) row-ordered elements -> the gather's pixel-pair order (BN, S + (S & 1), M, 32), zero pad row."""
    BN, S, M, D = rows.shape
    if S & 1:
        rows = np.concatenate([rows, np.zeros((BN, 1, M, D), rows.dtype)], 1)
    return np.ascontiguousarray(rows.reshape(BN, -1, 2, M, D).transpose(0, 1, 3, 2, 4)).reshape(BN, -1, M, D)
