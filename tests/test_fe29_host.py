"""The product's field/curve headers (panda_amd/csrc/fe29.h, curve29.h) compiled for the HOST with
FE29_CHECK instrumentation (128-bit shadow column accumulators, limb-range asserts) and compared
with the oracle.  This is the CPU-side sanitizer run for the arithmetic the HIP kernels execute."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle as po
import pyref

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "host_check", "fe29_host.cpp")
SO = os.path.join(HERE, "host_check", "libfe29_host.so")
CSRC = os.path.join(os.path.dirname(HERE), "panda_amd", "csrc")


@pytest.fixture(scope="module")
def h29():
    deps = [SRC] + [os.path.join(CSRC, f) for f in ("fe29.h", "curve29.h", "fe29_params.h")]
    if not os.path.exists(SO) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in deps):
        subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-o", SO, SRC], check=True)
    return C.CDLL(SO)


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def test_params_header_is_current():
    out = subprocess.run(["python3", os.path.join(CSRC, "gen_params.py")], check=True, capture_output=True, text=True).stdout
    assert out == open(os.path.join(CSRC, "fe29_params.h")).read()


@pytest.mark.parametrize("fid", [0, 1, 2, 3])
def test_field_ops(h29, fid):
    lc = po.FIELD_LC[fid]
    info = po.field_info(fid)
    mod = pyref.limbs_to_int(info["p"])
    rng = np.random.default_rng(100 + fid)
    vals = [0, 1, 2, mod - 1, mod - 2, (mod - 1) // 2, (1 << (mod.bit_length() - 1)), (1 << (mod.bit_length() - 1)) - 1]
    vals += [int.from_bytes(rng.bytes(4 * lc), "little") % mod for _ in range(3000)]
    # adversarial limb patterns: all 29-bit limbs at their maximum below p
    vals += [min(mod - 1, sum(((1 << 29) - 1) << (29 * i) for i in range(k))) for k in range(1, 14)]
    a = np.stack([pyref.int_to_limbs(v, lc) for v in vals])
    b = np.stack([pyref.int_to_limbs(v, lc) for v in reversed(vals)])
    for op in (po.OP_ADD, po.OP_SUB, po.OP_MUL, po.OP_SQR, po.OP_TO_MONT, po.OP_FROM_MONT):
        r = np.empty_like(a)
        assert h29.h29_field_op(fid, op, _p(r), _p(a), _p(b), C.c_size_t(len(vals))) == 0
        want = po.f_vec(fid, op, a, b)
        assert (r == want).all(), op
    r = np.empty_like(a[:40])
    h29.h29_field_op(fid, po.OP_INV, _p(r), _p(a[:40]), None, C.c_size_t(40))
    assert (r == po.f_vec(fid, po.OP_INV, a[:40])).all()


def _affine_all(cid, jac):
    return np.stack([po.to_affine(cid, j) for j in jac])


@pytest.mark.parametrize("cid", [0, 1])
def test_curve_ops(h29, cid):
    lc = po.LC_Q[cid]
    n = 200
    g = po.generator(cid)
    bases = po.gen_bases(cid, 42 + cid, n)
    scal = po.gen_scalars(po.FR_OF[cid], 43 + cid, n)
    # random Jacobian points with non-trivial Z: k_i * G
    jac = np.stack([po.scalar_mul(cid, g, pyref.int_to_limbs(pyref.limbs_to_int(s) % pyref.CURVES[cid].r, 8)) for s in scal[:n]])
    ident = np.zeros(3 * lc, dtype=np.uint32)
    aff_of_jac = _affine_all(cid, jac)
    negy = aff_of_jac.copy()
    negy[:, lc:] = po.f_vec(po.FQ_OF[cid], po.OP_SUB, np.zeros_like(negy[:, lc:]), negy[:, lc:])
    bz = bases.copy()
    bz[::3, :lc] = 0  # identity bases
    cases_madd = [(jac, bases), (jac, aff_of_jac), (jac, negy), (np.tile(ident, (n, 1)), bases), (jac, bz)]
    for A, B in cases_madd:
        r = np.empty_like(A)
        h29.h29_curve_op(cid, 0, _p(r), _p(np.ascontiguousarray(A)), _p(np.ascontiguousarray(B)), C.c_size_t(n))
        want = po.curve_vec(cid, po.COP_MADD, A, B)
        assert (_affine_all(cid, r) == _affine_all(cid, want)).all()
        assert ((r[:, 2 * lc:] == 0).all(axis=1) == (want[:, 2 * lc:] == 0).all(axis=1)).all()
    neg_jac = jac.copy()
    neg_jac[:, lc:2 * lc] = po.f_vec(po.FQ_OF[cid], po.OP_SUB, np.zeros_like(jac[:, lc:2 * lc]), jac[:, lc:2 * lc])
    for A, B in [(jac, jac[::-1].copy()), (jac, jac), (jac, neg_jac), (np.tile(ident, (n, 1)), jac), (jac, np.tile(ident, (n, 1)))]:
        r = np.empty_like(A)
        h29.h29_curve_op(cid, 1, _p(r), _p(np.ascontiguousarray(A)), _p(np.ascontiguousarray(B)), C.c_size_t(n))
        want = po.curve_vec(cid, po.COP_ADD, A, B)
        assert (_affine_all(cid, r) == _affine_all(cid, want)).all()
    r = np.empty_like(jac)
    h29.h29_curve_op(cid, 2, _p(r), _p(jac), None, C.c_size_t(n))
    assert (_affine_all(cid, r) == _affine_all(cid, po.curve_vec(cid, po.COP_DBL, jac))).all()


@pytest.mark.parametrize("cid", [0, 1])
def test_long_madd_chain_with_signs(h29, cid):
    """2000 dependent mixed additions with random signs, repeated and cancelling bases: the lazy
    bounds (X < 9p, Y < 5p ...) must hold along the whole chain (asserts inside the library)."""
    lc = po.LC_Q[cid]
    c = pyref.CURVES[cid]
    n = 2000
    seed = 7 + cid
    bases = po.gen_bases(cid, seed, n)
    bases[100:110] = bases[99]          # repeated base -> doubling branch
    bases[500] = bases[499]
    rng = np.random.default_rng(5)
    neg = rng.integers(0, 2, n).astype(np.uint8)
    neg[100:110] = neg[99]
    neg[500] = 1 - neg[499]             # P + (-P)
    bases[700, :lc] = 0                 # identity base
    out = np.empty(3 * lc, dtype=np.uint32)
    for hom in (0, 1):
        h29.h29_chain(cid, _p(out), _p(bases), _p(neg), C.c_size_t(n), hom)
        # expected via multipliers: sum (+/-) m_i
        k = 0
        for i in range(n):
            if i == 700:
                continue
            src = i
            if 100 <= i < 110:
                src = 99
            if i == 500:
                src = 499
            m = po.gen_multiplier(seed, src)
            k += -m if neg[i] else m
        want = pyref.encode_affine(c, pyref.ec_mul(c, k % c.r, c.g))
        got = po.hom_to_affine(cid, out) if hom else po.to_affine(cid, out)
        assert (got == want).all()


def test_k13_all_equal_bases_chain(h29, golden_dir):
    # every step after the first is the P + P / general madd mix the reference's fixture exercises
    g = po.generator(0)
    n = 300
    bases = np.tile(g, (n, 1))
    out = np.empty(24, dtype=np.uint32)
    h29.h29_chain(0, _p(out), _p(bases), None, C.c_size_t(n), 0)
    c = pyref.CURVES[0]
    assert (po.to_affine(0, out) == pyref.encode_affine(c, pyref.ec_mul(c, n, c.g))).all()


@pytest.mark.parametrize("fid", [1, 3, 5])
def test_precomputed_quotient_product(h29, fid):
    """fe_mul_shoup / fe_shoup_prepare (the NTT's twiddle products): x * w - q * p is congruent to x * w, tight, below
    (x / R + 2) p, for operands at the edge of the limb contract (limbs up to 2^31 + 2^24, values up to 8 R); the stored
    quotient is floor(w R / p).  The column accumulators are shadowed in 128 bits inside the library (FE29_CHECK)."""
    info = po.field_info(fid)
    mod = pyref.limbs_to_int(info["p"])
    N, W = 9, 29
    R = 1 << (W * N)
    rng = np.random.default_rng(900 + fid)
    n = 4000
    limb_max = (1 << 31) + (1 << 24) - 1
    x = rng.integers(0, 1 << 29, size=(n, N), dtype=np.uint64)
    x[n // 4: n // 2] = rng.integers(0, limb_max + 1, size=(n // 4, N), dtype=np.uint64)  # raw differences
    x[n // 2: n // 2 + 50] = limb_max                                                        # every column at its maximum
    x[n // 2 + 50: n // 2 + 100] = 0
    x[:, N - 1] = np.minimum(x[:, N - 1], (1 << 32) - 1)
    x = x.astype(np.uint32)
    ws = [0, 1, 2, mod - 1, mod - 2, (mod - 1) // 2] + [int.from_bytes(rng.bytes(40), "little") % mod for _ in range(n - 6)]
    w_int = np.array([[((w * R % mod) >> (W * j)) & ((1 << W) - 1) if j < N - 1 else (w * R % mod) >> (W * (N - 1)) for j in range(N)] for w in ws], dtype=np.uint32)
    r = np.empty((n, N), dtype=np.uint32)
    tw = np.empty((n, 2 * N), dtype=np.uint32)
    assert h29.h29_shoup(fid, _p(r), _p(tw), _p(x), _p(w_int), C.c_size_t(n)) == 0
    for i in range(n):
        X = sum(int(x[i, j]) << (W * j) for j in range(N))
        got = sum(int(r[i, j]) << (W * j) for j in range(N))
        assert all(int(r[i, j]) < (1 << W) for j in range(N))
        assert got % mod == X * ws[i] % mod
        assert got < (X // R + 3) * mod
        assert sum(int(tw[i, j]) << (W * j) for j in range(N)) == ws[i]
        assert sum(int(tw[i, N + j]) << (W * j) for j in range(N)) == ws[i] * R // mod


@pytest.mark.parametrize("fid", [1, 3, 5])
def test_multiply_add_reduction(h29, fid):
    """fe_reduce_mad_2p: raw limbs (< 2^32), value below 2^9 p -> tight, same residue, below 2p."""
    info = po.field_info(fid)
    mod = pyref.limbs_to_int(info["p"])
    N, W = 9, 29
    rng = np.random.default_rng(950 + fid)
    n = 4000
    x = np.zeros((n, N), dtype=np.uint32)
    vals = []
    for i in range(n):
        k = int(rng.integers(0, 512))
        v = k * mod + int.from_bytes(rng.bytes(40), "little") % mod if i % 7 else k * mod + (mod - 1 if i % 2 else 0)
        v = min(v, 512 * mod - 1)
        # a random carry-free limb decomposition with limbs up to 2^32 - 1: move multiples of 2^29 downwards
        l = [(v >> (W * j)) & ((1 << W) - 1) for j in range(N - 1)] + [v >> (W * (N - 1))]
        for j in range(N - 1, 0, -1):
            t = min(int(rng.integers(0, 8)), l[j])
            l[j] -= t
            l[j - 1] += t << W
        assert all(0 <= a < (1 << 32) for a in l) and sum(a << (W * j) for j, a in enumerate(l)) == v
        x[i] = l
        vals.append(v)
    r = np.empty_like(x)
    assert h29.h29_reduce_mad(fid, _p(r), _p(x), C.c_size_t(n)) == 0
    for i in range(n):
        got = sum(int(r[i, j]) << (W * j) for j in range(N))
        assert all(int(r[i, j]) < (1 << W) for j in range(N))
        assert got % mod == vals[i] % mod and got < 2 * mod


@pytest.mark.parametrize("fid,units", [(1, 2), (1, 3), (3, 3), (5, 3)])
def test_butterfly_difference_into_product(h29, fid, units):
    """The NTT butterfly's difference: a - b + K p with the per-limb bias at `units` * 2^29 and NO carry pass, fed straight into the
    precomputed-quotient product.  Operands at the edge of their limb class -- loose (2^29 + 8) for 2 units, un-normalised sums
    (2^30 + 16) for 3 -- and values up to 8 p; the library's own asserts (limb ranges, 128-bit shadow columns) run inside."""
    info = po.field_info(fid)
    mod = pyref.limbs_to_int(info["p"])
    N, W = 9, 29
    R = 1 << (W * N)
    rng = np.random.default_rng(970 + fid + units)
    n = 3000
    lim = (1 << 29) + 8 if units == 2 else (1 << 30) + 16

    def operands():
        v = np.empty((n, N), dtype=np.uint32)
        vals = []
        for i in range(n):
            k = int(rng.integers(0, 8))
            val = k * mod + int.from_bytes(rng.bytes(40), "little") % mod
            l = [(val >> (W * j)) & ((1 << W) - 1) for j in range(N - 1)] + [val >> (W * (N - 1))]
            for j in range(N - 1, 0, -1):  # push weight downwards until the limbs sit at the top of their class
                t = min((lim - l[j - 1]) >> W, l[j]) if i % 3 else 0
                l[j] -= t
                l[j - 1] += t << W
            assert all(0 <= a <= lim for a in l[:-1]) and sum(a << (W * j) for j, a in enumerate(l)) == val
            v[i] = l
            vals.append(val)
        return v, vals

    a, av = operands()
    b, bv = operands()
    a[:20, :N - 1] = lim  # every column at its maximum (the value no longer matters for the limb asserts; keep it below the bound)
    b[:20, :N - 1] = 0
    av[:20] = [sum(int(x) << (W * j) for j, x in enumerate(row)) for row in a[:20]]
    bv[:20] = [sum(int(x) << (W * j) for j, x in enumerate(row)) for row in b[:20]]
    ws = [int.from_bytes(rng.bytes(40), "little") % mod for _ in range(n)]
    w_int = np.array([[((w * R % mod) >> (W * j)) & ((1 << W) - 1) if j < N - 1 else (w * R % mod) >> (W * (N - 1)) for j in range(N)] for w in ws], dtype=np.uint32)
    r = np.empty((n, N), dtype=np.uint32)
    assert h29.h29_bfly_diff(fid, units, _p(r), _p(a), _p(b), _p(w_int), C.c_size_t(n)) == 0
    for i in range(n):
        got = sum(int(r[i, j]) << (W * j) for j in range(N))
        assert all(int(r[i, j]) < (1 << W) for j in range(N))
        assert got % mod == (av[i] - bv[i]) * ws[i] % mod
        assert got < 4 * mod
