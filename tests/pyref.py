"""Independent pure-Python big-integer reference (test infrastructure).

Nothing here shares code with oracle/*.c or the HIP kernels: points are handled in affine
coordinates over Python ints, so agreement with the oracle is a real cross-check.
Used to pin the oracle (tests/test_oracle.py) next to the reference's own k13 golden vector.

Wire formats follow the reference (src/utils.rs:1-14; affine.cuh:11-19; projective.cuh:9-20):
Montgomery-form little-endian u32 limbs, affine = x||y, identity <=> x == 0.
"""
from __future__ import annotations

import numpy as np

BN254 = 0
BLS12_377 = 1
BLS12_381 = 2


class Curve:
    def __init__(self, name, p, r, b, gx, gy, lc_q, lc_r, r_bits):
        self.name, self.p, self.r, self.b, self.g = name, p, r, b, (gx, gy)
        self.lc_q, self.lc_r, self.r_bits = lc_q, lc_r, r_bits
        self.Rq = (1 << (32 * lc_q)) % p
        self.Rr = (1 << (32 * lc_r)) % r
        self.Rq_inv = pow(self.Rq, -1, p)
        self.Rr_inv = pow(self.Rr, -1, r)


CURVES = {
    BN254: Curve(
        "bn254",
        0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47,
        0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001,
        3, 1, 2, 8, 8, 254),
    BLS12_377: Curve(
        "bls12_377",
        0x01AE3A4617C510EAC63B05C06CA1493B1A22D9F300F5138F1EF3622FBA094800170B5D44300000008508C00000000001,
        0x12AB655E9A2CA55660B44D1E5C37B00159AA76FED00000010A11800000000001,
        1,
        0x008848DEFE740A67C8FC6225BF87FF5485951E2CAA9D41BB188282C8BD37CB5CD5481512FFCD394EEAB9B16EB21BE9EF,
        0x01914A69C5102EFF1F674F5D30AFEEC4BD7FB348CA3E52D96D182AD44FB82305C2FE3D3634A9591AFD82DE55559C8EA6,
        12, 8, 253),
    BLS12_381: Curve(  # the reference names the curve (curve.cuh:12) without parameters: the standard ones
        "bls12_381",
        0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB,
        0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001,
        4,
        0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB,
        0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1,
        12, 8, 255),
}


def limbs_to_int(a) -> int:
    return int.from_bytes(np.ascontiguousarray(a, dtype=np.uint32).tobytes(), "little")


def int_to_limbs(v: int, lc: int) -> np.ndarray:
    return np.frombuffer(int(v).to_bytes(4 * lc, "little"), dtype=np.uint32).copy()


def ec_add(c: Curve, P, Q):
    """Affine addition over Python ints; None is the identity."""
    if P is None:
        return Q
    if Q is None:
        return P
    p = c.p
    x1, y1 = P
    x2, y2 = Q
    if x1 == x2:
        if (y1 + y2) % p == 0:
            return None
        lam = 3 * x1 * x1 * pow(2 * y1, -1, p) % p
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, p) % p
    x3 = (lam * lam - x1 - x2) % p
    return x3, (lam * (x1 - x3) - y1) % p


def ec_mul(c: Curve, k: int, P):
    R = None
    while k:
        if k & 1:
            R = ec_add(c, R, P)
        P = ec_add(c, P, P)
        k >>= 1
    return R


def is_on_curve(c: Curve, P) -> bool:
    x, y = P
    return (y * y - x * x * x - c.b) % c.p == 0


def decode_affine(c: Curve, raw):
    """x||y Montgomery limbs -> (x, y) ints or None for the wire identity (x == 0)."""
    raw = np.ascontiguousarray(raw, dtype=np.uint32)
    x = limbs_to_int(raw[: c.lc_q]) * c.Rq_inv % c.p
    y = limbs_to_int(raw[c.lc_q:]) * c.Rq_inv % c.p
    if limbs_to_int(raw[: c.lc_q]) == 0:
        return None
    return x, y


def encode_affine(c: Curve, P) -> np.ndarray:
    """(x, y) -> Montgomery limbs; identity -> (0, R) as Projective::to_affine does (projective.cuh:81-87)."""
    if P is None:
        return np.concatenate([int_to_limbs(0, c.lc_q), int_to_limbs(c.Rq, c.lc_q)])
    return np.concatenate([int_to_limbs(P[0] * c.Rq % c.p, c.lc_q), int_to_limbs(P[1] * c.Rq % c.p, c.lc_q)])


def decode_jacobian(c: Curve, raw):
    raw = np.ascontiguousarray(raw, dtype=np.uint32)
    lc = c.lc_q
    X, Y, Z = (limbs_to_int(raw[i * lc:(i + 1) * lc]) * c.Rq_inv % c.p for i in range(3))
    if Z == 0:
        return None
    zi = pow(Z, -1, c.p)
    return X * zi * zi % c.p, Y * zi * zi * zi % c.p


def decode_homogeneous(c: Curve, raw):
    raw = np.ascontiguousarray(raw, dtype=np.uint32)
    lc = c.lc_q
    X, Y, Z = (limbs_to_int(raw[i * lc:(i + 1) * lc]) * c.Rq_inv % c.p for i in range(3))
    if Z == 0:
        return None
    zi = pow(Z, -1, c.p)
    return X * zi % c.p, Y * zi % c.p


def decode_scalar(c: Curve, raw) -> int:
    """Montgomery-form Fr limbs -> canonical integer."""
    return limbs_to_int(raw) * c.Rr_inv % c.r


def msm(c: Curve, bases: np.ndarray, scalars: np.ndarray):
    """Sum s_i * P_i with plain double-and-add; bases (n, 2*lc_q) u32, scalars (n, lc_r) u32."""
    acc = None
    for b, s in zip(bases, scalars):
        P = decode_affine(c, b)
        if P is None:
            continue
        acc = ec_add(c, acc, ec_mul(c, decode_scalar(c, s), P))
    return acc


def dft(c: Curve, x, omega: int):
    """y[k] = sum_j x[j] omega^(jk) over Fr on canonical ints."""
    n = len(x)
    return [sum(x[j] * pow(omega, j * k, c.r) for j in range(n)) % c.r for k in range(n)]
