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


# ---------------------------------------------------------------------------------------------------- BN254 G2
# The twist y^2 = x^3 + 3/(9+u) over Fq2 = Fq[u]/(u^2 + 1).  The reference holds nothing for G2 (it is not mentioned anywhere in it):
# this implementation -- affine arithmetic over pairs of Python ints -- is the only oracle the G2 tests have, pinned by the curve
# equation and the group order of the standard generator (EIP-197 / arkworks).  Wire format: an Fq2 element is c0 || c1, each 8
# Montgomery-form u32 limbs; affine x || y = 32 words (identity <=> x == 0), Jacobian X || Y || Z = 48 words.
BN254_G2 = 3
G2_GEN = ((10857046999023057135944570762232829481370756359578518086990519993285655852781,
           11559732032986387107991004021392285783925812861821192530917403151452391805634),
          (8495653923123431417604973247489272438418190587263600148770280649306958101930,
           4082367875863433681332203403145435568316851327593401208105741076214120093531))


def f2_mul(a, b, p):
    return (a[0] * b[0] - a[1] * b[1]) % p, (a[0] * b[1] + a[1] * b[0]) % p


def f2_add(a, b, p):
    return (a[0] + b[0]) % p, (a[1] + b[1]) % p


def f2_sub(a, b, p):
    return (a[0] - b[0]) % p, (a[1] - b[1]) % p


def f2_inv(a, p):
    n = pow((a[0] * a[0] + a[1] * a[1]) % p, -1, p)
    return a[0] * n % p, -a[1] * n % p


def g2_b():
    p = CURVES[BN254].p
    return f2_mul((3, 0), f2_inv((9, 1), p), p)


def g2_is_on_curve(P) -> bool:
    p = CURVES[BN254].p
    x, y = P
    return f2_mul(y, y, p) == f2_add(f2_mul(f2_mul(x, x, p), x, p), g2_b(), p)


def g2_add(P, Q):
    """Affine addition on the twist; None is the identity."""
    p = CURVES[BN254].p
    if P is None:
        return Q
    if Q is None:
        return P
    if P[0] == Q[0]:
        if f2_add(P[1], Q[1], p) == (0, 0):
            return None
        lam = f2_mul(f2_mul((3, 0), f2_mul(P[0], P[0], p), p), f2_inv(f2_add(P[1], P[1], p), p), p)
    else:
        lam = f2_mul(f2_sub(Q[1], P[1], p), f2_inv(f2_sub(Q[0], P[0], p), p), p)
    x3 = f2_sub(f2_sub(f2_mul(lam, lam, p), P[0], p), Q[0], p)
    return x3, f2_sub(f2_mul(lam, f2_sub(P[0], x3, p), p), P[1], p)


def g2_mul(k: int, P):
    R = None
    while k:
        if k & 1:
            R = g2_add(R, P)
        P = g2_add(P, P)
        k >>= 1
    return R


def _f2_from_wire(raw):
    c = CURVES[BN254]
    return limbs_to_int(raw[:8]) * c.Rq_inv % c.p, limbs_to_int(raw[8:16]) * c.Rq_inv % c.p


def _f2_to_wire(v):
    c = CURVES[BN254]
    return np.concatenate([int_to_limbs(v[0] * c.Rq % c.p, 8), int_to_limbs(v[1] * c.Rq % c.p, 8)])


def g2_decode_affine(raw):
    raw = np.ascontiguousarray(raw, dtype=np.uint32).reshape(-1)
    if not raw[:16].any():
        return None
    return _f2_from_wire(raw[:16]), _f2_from_wire(raw[16:32])


def g2_encode_affine(P) -> np.ndarray:
    if P is None:
        return np.zeros(32, np.uint32)
    return np.concatenate([_f2_to_wire(P[0]), _f2_to_wire(P[1])])


def g2_decode_jacobian(raw):
    p = CURVES[BN254].p
    raw = np.ascontiguousarray(raw, dtype=np.uint32).reshape(-1)
    X, Y, Z = (_f2_from_wire(raw[16 * i:16 * (i + 1)]) for i in range(3))
    if Z == (0, 0):
        return None
    zi = f2_inv(Z, p)
    zi2 = f2_mul(zi, zi, p)
    return f2_mul(X, zi2, p), f2_mul(Y, f2_mul(zi2, zi, p), p)


def g2_decode_homogeneous(raw):
    p = CURVES[BN254].p
    raw = np.ascontiguousarray(raw, dtype=np.uint32).reshape(-1)
    X, Y, Z = (_f2_from_wire(raw[16 * i:16 * (i + 1)]) for i in range(3))
    if Z == (0, 0):
        return None
    zi = f2_inv(Z, p)
    return f2_mul(X, zi, p), f2_mul(Y, zi, p)


def g2_msm(bases: np.ndarray, scalars: np.ndarray):
    c = CURVES[BN254]
    acc = None
    for b, s in zip(bases, scalars):
        P = g2_decode_affine(b)
        if P is not None:
            acc = g2_add(acc, g2_mul(decode_scalar(c, s), P))
    return acc
