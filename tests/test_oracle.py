"""Pins the CPU oracle: reference golden vector (k13), reference parameter tables, and an
independent pure-Python big-int implementation.  No GPU needed."""
import hashlib
import os

import numpy as np
import pytest

import oracle as po
import pyref

REF = "/root/reference/src/cuda"


def k13_bases():
    g = po.generator(po.BN254)
    b = np.tile(g, (8192, 1))
    assert hashlib.sha256(b.tobytes()).hexdigest() == "bd5242cf1c5eda2de10338d28d0b0cb48176326ef1b01b0382c55dccfa8db8ce"
    return b


def test_field_constants_match_reference_tables():
    # bn254/paramter.cuh:99-119 (Fq R2, inv, ONE), :216-237 (Fr); bls12_377/paramter.cuh:39 (inv)
    fq = po.field_info(po.F_BN254_FQ)
    assert fq["inv"] == 0xE4866389
    assert [int(x) for x in fq["one"]] == [0xC58F0D9D, 0xD35D438D, 0xF5C70B3D, 0x0A78EB28, 0x7879462C, 0x666EA36F, 0x9A07DF2F, 0x0E0A77C1]
    assert [int(x) for x in fq["r2"]] == [0x538AFA89, 0xF32CFC5B, 0xD44501FB, 0xB5E71911, 0x0A417FF6, 0x47AB1EFF, 0xCAB8351F, 0x06D89F71]
    fr = po.field_info(po.F_BN254_FR)
    assert fr["inv"] == 0xEFFFFFFF
    assert [int(x) for x in fr["one"]] == [0x4FFFFFFB, 0xAC96341C, 0x9F60CD29, 0x36FC7695, 0x7879462E, 0x666EA36F, 0x9A07DF2F, 0x0E0A77C1]
    assert [int(x) for x in fr["r2"]] == [0xAE216DA7, 0x1BB8E645, 0xE35C59E3, 0x53FE3AB1, 0x53BB8085, 0x8C49833D, 0x7F4E44A5, 0x0216D0B1]
    assert po.field_info(po.F_BLS377_FQ)["inv"] == 0xFFFFFFFF
    assert po.field_info(po.F_BLS377_FR)["inv"] == 0xFFFFFFFF
    for fid, (cid, which) in {0: (0, "p"), 1: (0, "r"), 2: (1, "p"), 3: (1, "r")}.items():
        c = pyref.CURVES[cid]
        info = po.field_info(fid)
        mod = getattr(c, which)
        assert pyref.limbs_to_int(info["p"]) == mod
        assert pyref.limbs_to_int(info["one"]) == (1 << (32 * info["lc"])) % mod
        assert pyref.limbs_to_int(info["r2"]) == (1 << (64 * info["lc"])) % mod
        assert info["bits"] == mod.bit_length()


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")
def test_bls12_377_tables_match_reference_text():
    # study of the reference as text: the 32-bit-limb tables of bls12_377/paramter.cuh:23-75
    import re
    txt = open(f"{REF}/core/curve/bls12_377/paramter.cuh").read()
    nums = [int(x, 16) for x in re.findall(r"0x([0-9a-fA-F]{8})\b", txt)]
    fq = po.field_info(po.F_BLS377_FQ)
    assert nums[0:12] == [int(x) for x in fq["p"]]
    assert nums[13:25] == [int(x) for x in fq["one"]]
    assert nums[25:37] == [int(x) for x in fq["r2"]]


@pytest.mark.parametrize("fid", [0, 1, 2, 3, 4, 5])
def test_field_ops_vs_python(fid):
    cid, which = {0: (0, "p"), 1: (0, "r"), 2: (1, "p"), 3: (1, "r"), 4: (2, "p"), 5: (2, "r")}[fid]
    mod = getattr(pyref.CURVES[cid], which)
    lc = po.FIELD_LC[fid]
    R = (1 << (32 * lc)) % mod
    Ri = pow(R, -1, mod)
    rng = np.random.default_rng(fid + 1)
    vals = [0, 1, mod - 1, mod - 2, R, 2, (mod + 1) // 2] + [int.from_bytes(rng.bytes(4 * lc), "little") % mod for _ in range(64)]
    a = np.stack([pyref.int_to_limbs(v, lc) for v in vals])
    b = np.stack([pyref.int_to_limbs(v, lc) for v in reversed(vals)])
    ai, bi = vals, list(reversed(vals))
    got = lambda arr: [pyref.limbs_to_int(r) for r in arr]
    assert got(po.f_vec(fid, po.OP_ADD, a, b)) == [(x + y) % mod for x, y in zip(ai, bi)]
    assert got(po.f_vec(fid, po.OP_SUB, a, b)) == [(x - y) % mod for x, y in zip(ai, bi)]
    assert got(po.f_vec(fid, po.OP_MUL, a, b)) == [x * y * Ri % mod for x, y in zip(ai, bi)]
    assert got(po.f_vec(fid, po.OP_SQR, a)) == [x * x * Ri % mod for x in ai]
    assert got(po.f_vec(fid, po.OP_TO_MONT, a)) == [x * R % mod for x in ai]
    assert got(po.f_vec(fid, po.OP_FROM_MONT, a)) == [x * Ri % mod for x in ai]
    inv = got(po.f_vec(fid, po.OP_INV, a))
    for x, y in zip(ai, inv):
        if x == 0:
            assert y == 0
        else:
            assert x * y % mod == R * R % mod  # (xR)^-1 in Montgomery form: x^-1 R


@pytest.mark.parametrize("cid", [0, 1, 2])
def test_generators_on_curve(cid):
    c = pyref.CURVES[cid]
    assert pyref.is_on_curve(c, c.g)
    g = po.generator(cid)
    assert po.is_on_curve(cid, g)
    assert pyref.decode_affine(c, g) == c.g
    assert pyref.ec_mul(c, c.r, c.g) is None


@pytest.mark.parametrize("cid", [0, 1, 2])
def test_group_law_vs_python(cid):
    c = pyref.CURVES[cid]
    lc = c.lc_q
    g = po.generator(cid)
    rng = np.random.default_rng(11 + cid)
    ks = [1, 2, 3, c.r - 1, c.r - 2] + [int.from_bytes(rng.bytes(32), "little") % c.r for _ in range(12)]
    jac = [po.scalar_mul(cid, g, pyref.int_to_limbs(k, 8)) for k in ks]
    pts = [pyref.ec_mul(c, k, c.g) for k in ks]
    for j, p in zip(jac, pts):
        assert pyref.decode_jacobian(c, j) == p
        assert (po.to_affine(cid, j) == pyref.encode_affine(c, p)).all()
        assert pyref.decode_homogeneous(c, po.to_projective(cid, j)) == p
        assert (po.hom_to_affine(cid, po.to_projective(cid, j)) == po.to_affine(cid, j)).all()
    ident = np.zeros(3 * lc, dtype=np.uint32)
    A = np.stack(jac + [ident, jac[0], jac[3], ident])
    B = np.stack(list(reversed(jac)) + [jac[1], ident, jac[0], ident])  # includes P+identity, P+(-P)=(r-1)G+G, 0+0
    exp = [pyref.ec_add(c, pyref.decode_jacobian(c, a), pyref.decode_jacobian(c, b)) for a, b in zip(A, B)]
    got = po.curve_vec(cid, po.COP_ADD, A, B)
    assert [pyref.decode_jacobian(c, x) for x in got] == exp
    # equal operands hit the doubling branch of add_2007_bl (projective.cuh:228-231)
    got = po.curve_vec(cid, po.COP_ADD, np.stack(jac), np.stack(jac))
    assert [pyref.decode_jacobian(c, x) for x in got] == [pyref.ec_add(c, p, p) for p in pts]
    got = po.curve_vec(cid, po.COP_DBL, A)
    assert [pyref.decode_jacobian(c, x) for x in got] == [pyref.ec_add(c, pyref.decode_jacobian(c, a), pyref.decode_jacobian(c, a)) for a in A]
    # mixed add incl. identity accumulator, identity base (x == 0), equal points, opposite points
    aff = np.stack([po.to_affine(cid, j) for j in jac])
    affz = aff.copy()
    affz[0, :lc] = 0
    negy = aff.copy()
    negy[:, lc:] = po.f_vec(po.FQ_OF[cid], po.OP_SUB, np.zeros_like(aff[:, lc:]), aff[:, lc:])
    for P1, Q in ((np.stack(jac), aff[::-1]), (np.stack(jac), aff), (np.stack(jac), negy), (np.stack([ident] * len(jac)), aff), (np.stack(jac), affz)):
        got = po.curve_vec(cid, po.COP_MADD, P1, Q)
        exp = [pyref.ec_add(c, pyref.decode_jacobian(c, a), pyref.decode_affine(c, q)) for a, q in zip(P1, Q)]
        assert [pyref.decode_jacobian(c, x) for x in got] == exp


def test_k13_golden_vector(golden_dir):
    """The reference's own fixture: 8192 x generator, result_affine.bin (tests/test.rs:149-162)."""
    scalars = np.fromfile(os.path.join(golden_dir, "ref_k13_scalars.bin"), dtype=np.uint32).reshape(-1, 8)
    want = np.fromfile(os.path.join(golden_dir, "ref_k13_result_affine.bin"), dtype=np.uint32)
    bases = k13_bases()
    got = po.msm_affine(po.BN254, bases, scalars, window_bits=16)
    assert (got == want).all()
    # independent: (sum s_i) * G with Python ints
    c = pyref.CURVES[0]
    k = sum(pyref.decode_scalar(c, s) for s in scalars) % c.r
    assert (pyref.encode_affine(c, pyref.ec_mul(c, k, c.g)) == want).all()
    # other window sizes give the same point
    for w in (7, 13):
        assert (po.msm_affine(po.BN254, bases, scalars, window_bits=w) == want).all()


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")
def test_k13_fixture_files_equal_reference(golden_dir):
    for ours, theirs in (("ref_k13_scalars.bin", "scalars.bin"), ("ref_k13_result_affine.bin", "result_affine.bin")):
        assert open(os.path.join(golden_dir, ours), "rb").read() == open(f"{REF}/test/data/msm/k13/{theirs}", "rb").read()


@pytest.mark.parametrize("cid", [0, 1, 2])
def test_msm_vs_python_and_naive(cid):
    c = pyref.CURVES[cid]
    n = 48
    bases = po.gen_bases(cid, 0x1234 + cid, n)
    scalars = po.gen_scalars(po.FR_OF[cid], 0x99 + cid, n)
    # edge scalars: 0, 1, r-1, < 2^16 ; edge bases: identity (x==0), duplicate, opposite
    R = c.Rr
    for i, v in enumerate([0, 1, c.r - 1, 0xBEEF]):
        scalars[i] = pyref.int_to_limbs(v * R % c.r, 8)
    bases[5, : c.lc_q] = 0
    bases[7] = bases[6]
    bases[9] = bases[8]
    bases[9, c.lc_q:] = po.f_vec(po.FQ_OF[cid], po.OP_SUB, np.zeros((1, c.lc_q), np.uint32), bases[8:9, c.lc_q:])[0]
    scalars[9] = scalars[8]
    want = pyref.encode_affine(c, pyref.msm(c, bases, scalars))
    assert (po.to_affine(cid, po.msm_naive(cid, bases, scalars)) == want).all()
    for w in (4, 9, 16):
        assert (po.msm_affine(cid, bases, scalars, window_bits=w) == want).all()
    assert (po.msm_affine(cid, bases, scalars, window_bits=11, threads=4) == want).all()


@pytest.mark.parametrize("cid", [0, 1, 2])
def test_generated_bases_and_linearity(cid):
    c = pyref.CURVES[cid]
    n = 300
    seed = 0x70616E6461 ^ cid
    bases = po.gen_bases(cid, seed, n)
    for i in (0, 1, 17, n - 1):
        m = po.gen_multiplier(seed, i)
        assert m & 1 and m < 2**64
        assert pyref.decode_affine(c, bases[i]) == pyref.ec_mul(c, m, c.g)
    # window [first, first+n) of the stream equals the tail of a longer one
    assert (po.gen_bases(cid, seed, 10, first=290) == bases[290:]).all()
    scalars = po.gen_scalars(po.FR_OF[cid], seed + 1, n)
    assert all(pyref.limbs_to_int(s) < c.r for s in scalars)
    assert (po.gen_scalars(po.FR_OF[cid], seed + 1, 5, first=100) == scalars[100:105]).all()
    want = po.expected_from_linearity(cid, seed, scalars)
    k = sum(pyref.decode_scalar(c, s) * po.gen_multiplier(seed, i) for i, s in enumerate(scalars)) % c.r
    assert (want == pyref.encode_affine(c, pyref.ec_mul(c, k, c.g))).all()
    assert (po.msm_affine(cid, bases, scalars, window_bits=8) == want).all()


def test_msm_all_zero_scalars_gives_identity():
    n = 64
    bases = po.gen_bases(0, 5, n)
    out = po.msm(0, bases, np.zeros((n, 8), np.uint32), window_bits=8)
    assert not out[16:].any()  # Z == 0
    aff = po.to_affine(0, out)
    assert not aff[:8].any() and (aff[8:] == po.field_info(0)["one"]).all()


@pytest.mark.parametrize("fid", [po.F_BN254_FR, po.F_BLS377_FR, po.F_BLS381_FR])
def test_ntt_definition(fid):
    cid = {po.F_BN254_FR: 0, po.F_BLS377_FR: 1, po.F_BLS381_FR: 2}[fid]
    c = pyref.CURVES[cid]
    for log_n in (0, 1, 3, 6):
        n = 1 << log_n
        om = po.root_of_unity(fid, log_n)
        w = pyref.decode_scalar(c, om)
        assert pow(w, n, c.r) == 1 and (n == 1 or pow(w, n // 2, c.r) == c.r - 1)
        x = po.gen_scalars(fid, 77 + log_n, n)
        xi = [pyref.decode_scalar(c, v) for v in x]
        want = pyref.dft(c, xi, w)
        for impl in (po.dft_naive, po.ntt):
            y = impl(fid, x, om, log_n)
            assert [pyref.decode_scalar(c, v) for v in y] == want
        y, flag = po.ntt_passes(fid, x, om, log_n)
        assert [pyref.decode_scalar(c, v) for v in y] == want
        assert flag == ((log_n + 7) // 8) % 2


@pytest.mark.parametrize("fid,log_n", [(po.F_BN254_FR, 0), (po.F_BN254_FR, 5), (po.F_BN254_FR, 11), (po.F_BLS377_FR, 9), (po.F_BLS381_FR, 9)])
def test_ntt_single_output_evaluation(fid, log_n):
    """po_ntt_eval_at (Horner, O(n) per output: the full-size spot check of the GPU tests) against the transform and the O(n^2) definition."""
    n = 1 << log_n
    om = po.root_of_unity(fid, log_n)
    x = po.gen_scalars(fid, 77 + log_n, n)
    y = po.ntt(fid, x, om, log_n)
    if log_n <= 9:
        assert (y == po.dft_naive(fid, x, om, log_n)).all()
    for k in sorted({0, 1 % n, n // 2, n - 1, 37 % n}):
        assert (po.ntt_eval_at(fid, x, om, log_n, k) == y[k]).all()
    assert (po.ntt_eval_at(fid, x, om, log_n, n + 3) == y[3 % n]).all()  # k is taken mod n


def test_bn254_omega_table_value():
    # bn254/paramter.cuh:251-258: omega of order 2^28 in Montgomery form
    om = po.root_of_unity(po.F_BN254_FR, 28)
    assert [int(x) for x in om] == [0xB639FEB8, 0x9632C7C5, 0x0D0FF299, 0x985CE340, 0x01B0ECD8, 0xB2DD8800, 0x6D98CE29, 0x1D69070D]


@pytest.mark.parametrize("log_n", [9, 10, 13, 17])
def test_ntt_fast_vs_passes_roundtrip_linearity(log_n):
    fid = po.F_BN254_FR
    c = pyref.CURVES[0]
    n = 1 << log_n
    om = po.root_of_unity(fid, log_n)
    x = po.gen_scalars(fid, 1000 + log_n, n)
    y = po.ntt(fid, x, om, log_n)
    if log_n <= 10:
        assert (po.dft_naive(fid, x, om, log_n) == y).all()
    y2, flag = po.ntt_passes(fid, x, om, log_n)
    assert (y2 == y).all() and flag == ((log_n + 7) // 8) % 2
    # inverse: omega^-1 then scale by n^-1
    om_inv = po.f_vec(fid, po.OP_INV, om[None])[0]
    n_inv = pyref.int_to_limbs(pow(n, -1, c.r) * c.Rr % c.r, 8)
    back = po.f_scale(fid, po.ntt(fid, y, om_inv, log_n), n_inv)
    assert (back == x).all()
    # delta -> all ones (Montgomery one); all ones -> n * delta
    one = po.field_info(fid)["one"]
    d = np.zeros_like(x)
    d[0] = one
    assert (po.ntt(fid, d, om, log_n) == one[None, :]).all()
    # linearity
    z = po.gen_scalars(fid, 2000 + log_n, n)
    lhs = po.ntt(fid, po.f_vec(fid, po.OP_ADD, x, z), om, log_n)
    rhs = po.f_vec(fid, po.OP_ADD, y, po.ntt(fid, z, om, log_n))
    assert (lhs == rhs).all()


def test_committed_golden_cases_still_match_oracle(golden_dir):
    """tests/golden/*.json were produced by tests/golden/make_golden.py; the oracle must keep reproducing them."""
    import json
    import sys
    sys.path.insert(0, golden_dir)
    import make_golden
    cases = json.load(open(os.path.join(golden_dir, "msm_cases.json")))
    for c in cases[:4] + cases[5:8]:
        n = 1 << c["log_n"]
        bases = np.tile(po.generator(0), (n, 1)) if c.get("bases") == "all_generator" else po.gen_bases(c["curve"], c["bases_seed"], n)
        scalars = make_golden.scalar_set(c["scalars"], c["curve"], n, c["scalars_seed"])
        assert po.msm_affine(c["curve"], bases, scalars, window_bits=9).tobytes().hex() == c["affine_hex"]
    for c in json.load(open(os.path.join(golden_dir, "ntt_cases.json")))[:5]:
        x = po.gen_scalars(po.F_BN254_FR, c["seed"], 1 << c["log_n"])
        om = np.frombuffer(bytes.fromhex(c["omega_hex"]), dtype=np.uint32)
        y = po.ntt(po.F_BN254_FR, x, om, c["log_n"])
        assert hashlib.sha256(y.tobytes()).hexdigest() == c["sha256"]
