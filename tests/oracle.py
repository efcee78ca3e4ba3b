"""ctypes binding to oracle/libpanda_oracle.so -- the CPU oracle (test infrastructure).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
The product package (panda_amd/) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_DIR = os.path.join(_ROOT, "oracle")
_SO = os.path.join(_DIR, "libpanda_oracle.so")

BN254, BLS12_377, BLS12_381 = 0, 1, 2
F_BN254_FQ, F_BN254_FR, F_BLS377_FQ, F_BLS377_FR, F_BLS381_FQ, F_BLS381_FR = 0, 1, 2, 3, 4, 5
OP_ADD, OP_SUB, OP_MUL, OP_SQR, OP_TO_MONT, OP_FROM_MONT, OP_INV = range(7)
COP_MADD, COP_ADD, COP_DBL = range(3)

LC_Q = {BN254: 8, BLS12_377: 12, BLS12_381: 12}
LC_R = {BN254: 8, BLS12_377: 8, BLS12_381: 8}
FIELD_LC = {F_BN254_FQ: 8, F_BN254_FR: 8, F_BLS377_FQ: 12, F_BLS377_FR: 8, F_BLS381_FQ: 12, F_BLS381_FR: 8}
FQ_OF = {BN254: F_BN254_FQ, BLS12_377: F_BLS377_FQ, BLS12_381: F_BLS381_FQ}
FR_OF = {BN254: F_BN254_FR, BLS12_377: F_BLS377_FR, BLS12_381: F_BLS381_FR}


def build(force: bool = False) -> str:
    srcs = [os.path.join(_DIR, f) for f in ("field.c", "curve.c", "msm.c", "ntt.c", "gen.c", "panda_oracle.h")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
        subprocess.run(["make", "-C", _DIR, "-B"], check=True, capture_output=True)
    return _SO


class _PoField(C.Structure):
    _fields_ = [("lc", C.c_uint32), ("bits", C.c_uint32), ("inv", C.c_uint32),
                ("p", C.c_uint32 * 12), ("one", C.c_uint32 * 12), ("r2", C.c_uint32 * 12)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.po_field_get.restype = C.POINTER(_PoField)
        _lib.po_gen_multiplier.restype = C.c_uint64
        _lib.po_gen_multiplier.argtypes = [C.c_uint64, C.c_uint64]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def field_info(field_id):
    f = lib().po_field_get(field_id).contents
    lc = f.lc
    return dict(lc=lc, bits=f.bits, inv=f.inv, p=np.array(f.p[:lc], dtype=np.uint32),
                one=np.array(f.one[:lc], dtype=np.uint32), r2=np.array(f.r2[:lc], dtype=np.uint32))


def f_vec(field_id, op, a, b=None):
    a = _u32(a)
    b = _u32(b) if b is not None else None
    r = np.empty_like(a)
    rc = lib().po_f_vec(field_id, op, _p(r), _p(a), _p(b), C.c_size_t(a.shape[0]))
    assert rc == 0
    return r


def curve_vec(curve, op, a, b=None):
    a = _u32(a)
    b = _u32(b) if b is not None else None
    r = np.empty((a.shape[0], 3 * LC_Q[curve]), dtype=np.uint32)
    rc = lib().po_curve_vec(curve, op, _p(r), _p(a), _p(b), C.c_size_t(a.shape[0]))
    assert rc == 0
    return r


def generator(curve):
    g = np.empty(2 * LC_Q[curve], dtype=np.uint32)
    lib().po_generator(curve, _p(g))
    return g


def is_on_curve(curve, aff):
    return bool(lib().po_is_on_curve(curve, _p(_u32(aff))))


def to_affine(curve, jac):
    jac = _u32(jac)
    out = np.empty(2 * LC_Q[curve], dtype=np.uint32)
    lib().po_to_affine(curve, _p(out), _p(jac))
    return out


def to_projective(curve, jac):
    jac = _u32(jac)
    out = np.empty(3 * LC_Q[curve], dtype=np.uint32)
    lib().po_to_projective(curve, _p(out), _p(jac))
    return out


def hom_to_affine(curve, hom):
    hom = _u32(hom)
    out = np.empty(2 * LC_Q[curve], dtype=np.uint32)
    lib().po_hom_to_affine(curve, _p(out), _p(hom))
    return out


def scalar_mul(curve, aff, k_limbs):
    aff, k = _u32(aff), _u32(k_limbs)
    out = np.empty(3 * LC_Q[curve], dtype=np.uint32)
    lib().po_scalar_mul(curve, _p(out), _p(aff), _p(k), C.c_uint(k.size))
    return out


def msm(curve, bases, scalars, window_bits=16, threads=1):
    """Jacobian X||Y||Z (u32 limbs) of sum_i scalars[i] * bases[i]; inputs are wire format."""
    bases, scalars = _u32(bases), _u32(scalars)
    n = scalars.size // LC_R[curve]
    assert bases.size == n * 2 * LC_Q[curve]
    out = np.zeros(3 * LC_Q[curve], dtype=np.uint32)
    rc = lib().po_msm_mt(curve, _p(bases), _p(scalars), C.c_uint64(n), C.c_uint(window_bits), C.c_uint(threads), _p(out))
    assert rc == 0, rc
    return out


def msm_naive(curve, bases, scalars):
    bases, scalars = _u32(bases), _u32(scalars)
    n = scalars.size // LC_R[curve]
    out = np.zeros(3 * LC_Q[curve], dtype=np.uint32)
    rc = lib().po_msm_naive(curve, _p(bases), _p(scalars), C.c_uint64(n), _p(out))
    assert rc == 0
    return out


def msm_affine(curve, bases, scalars, window_bits=16, threads=1):
    return to_affine(curve, msm(curve, bases, scalars, window_bits, threads))


def ntt(field_id, x, omega, log_n):
    x, omega = _u32(x), _u32(omega)
    out = np.empty_like(x)
    rc = lib().po_ntt(field_id, _p(out), _p(x), _p(omega), C.c_uint(log_n))
    assert rc == 0
    return out


def dft_naive(field_id, x, omega, log_n):
    x, omega = _u32(x), _u32(omega)
    out = np.empty_like(x)
    rc = lib().po_dft_naive(field_id, _p(out), _p(x), _p(omega), C.c_uint(log_n))
    assert rc == 0
    return out


def ntt_eval_at(field_id, x, omega, log_n, k):
    """y[k] = sum_j x[j] omega^(jk) for one k, O(n)."""
    x, omega = _u32(x), _u32(omega)
    out = np.empty(FIELD_LC[field_id], dtype=np.uint32)
    rc = lib().po_ntt_eval_at(field_id, _p(out), _p(x), _p(omega), C.c_uint(log_n), C.c_uint64(k))
    assert rc == 0
    return out


def ntt_passes(field_id, x, omega, log_n):
    x, omega = _u32(x), _u32(omega)
    out = np.empty_like(x)
    flag = C.c_uint(7)
    rc = lib().po_ntt_passes(field_id, _p(out), _p(x), _p(omega), C.c_uint(log_n), C.byref(flag))
    assert rc == 0
    return out, flag.value


def root_of_unity(field_id, log_n):
    out = np.empty(FIELD_LC[field_id], dtype=np.uint32)
    rc = lib().po_root_of_unity(field_id, C.c_uint(log_n), _p(out))
    assert rc == 0
    return out


def f_scale(field_id, x, s):
    x, s = _u32(x), _u32(s)
    out = np.empty_like(x)
    lc = FIELD_LC[field_id]
    rc = lib().po_f_scale(field_id, _p(out), _p(x), _p(s), C.c_size_t(x.size // lc))
    assert rc == 0
    return out


def gen_scalars(field_id, seed, n, first=0):
    out = np.empty((n, FIELD_LC[field_id]), dtype=np.uint32)
    lib().po_gen_scalars(field_id, C.c_uint64(seed), C.c_uint64(first), C.c_uint64(n), _p(out))
    return out


def gen_multiplier(seed, i):
    return int(lib().po_gen_multiplier(seed, i))


def gen_bases(curve, seed, n, first=0):
    out = np.empty((n, 2 * LC_Q[curve]), dtype=np.uint32)
    rc = lib().po_gen_bases(curve, C.c_uint64(seed), C.c_uint64(first), C.c_uint64(n), _p(out))
    assert rc == 0
    return out


def linear_combination(curve, seed_bases, scalars, first=0):
    scalars = _u32(scalars)
    n = scalars.size // LC_R[curve]
    out = np.empty(LC_R[curve], dtype=np.uint32)
    rc = lib().po_linear_combination(curve, C.c_uint64(seed_bases), C.c_uint64(first), _p(scalars), C.c_uint64(n), _p(out))
    assert rc == 0
    return out


def expected_from_linearity(curve, seed_bases, scalars, first=0):
    """Affine x||y of (sum s_i m_i) * G for bases generated by gen_bases(seed_bases)."""
    k = linear_combination(curve, seed_bases, scalars, first)
    return to_affine(curve, scalar_mul(curve, generator(curve), k))
