"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle and the golden fixtures.
Bit-exact throughout (integer arithmetic): MSM on the affine-normalised result, as the reference's own tests
compare (tests/test.rs:101-108); NTT on the fully reduced Montgomery-form outputs."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle as po
import pyref
from gpu_util import NULL_STREAM, DeviceBuffer
from panda_amd import gpu_ffi as ffi
from panda_amd import gpu_manager as pgm
from panda_amd import multi_gpu

pytestmark = pytest.mark.gpu


def _soak(*values):
    """a case of an off-by-default experiment path that only runs under `-m gpu_soak` (tests/conftest.py): `-m gpu` keeps one representative
    case per path -- the fuzzers (tools/fuzz_msm.py, fuzz_ntt.py, fuzz_ranges.py) cover these paths at random"""
    return pytest.param(*values, marks=pytest.mark.gpu_soak)



@pytest.fixture(scope="module")
def gm():
    m = pgm.PandaGpuManager(0)
    yield m
    m.deinit()


@pytest.fixture(autouse=True)
def _built_in_policies():
    """Every test starts and ends on the library's built-in policies: the global knobs a test turns (window width, chunk size, overlap,
    accumulate variant, merge mode, phase timers, streamed NTT tables, clock stamps) never leak into the tests that run after it, so the
    shipped defaults are what the rest of the suite covers whatever the order."""
    def reset():
        lib = ffi.load()
        lib.panda_msm_set_window_bits(0)
        lib.panda_msm_set_chunk_entries(0)
        lib.panda_msm_set_overlap(0xFFFFFFFF, 0)
        lib.panda_msm_set_accumulate_variant(0)
        lib.panda_msm_set_wide_merge(0)
        lib.panda_msm_set_chunk_first(1)
        lib.panda_msm_set_phase_timing(0)
        lib.panda_msm_set_paranoid(0)
        lib.panda_ntt_set_streamed_tables(0xFFFFFFFF)
        lib.panda_set_clock_stamps(0)
    reset()
    yield
    reset()


def affine_of(cid, result_bytes, coord=pgm.JACOBIAN):
    w = np.asarray(result_bytes).view(np.uint32)
    return po.hom_to_affine(cid, w) if coord == pgm.PROJECTIVE else po.to_affine(cid, w)


# ------------------------------------------------------------------ building blocks

@pytest.mark.parametrize("fid", [0, 1, 2, 3, 4, 5])
def test_field_ops_elementwise(fid):
    lc = po.FIELD_LC[fid]
    n = 1 << 14
    mod = pyref.limbs_to_int(po.field_info(fid)["p"])
    a = po.gen_scalars(fid, 11, n)
    b = po.gen_scalars(fid, 12, n)
    for i, v in enumerate([0, 1, mod - 1, mod - 2, (mod + 1) // 2]):
        a[i] = pyref.int_to_limbs(v, lc)
        b[n - 1 - i] = pyref.int_to_limbs(v, lc)
    da, db, dr = DeviceBuffer.from_host(a), DeviceBuffer.from_host(b), DeviceBuffer(a.nbytes)
    lib = ffi.load()
    for op in range(7):  # add, sub, mul, sqr, to / from Montgomery, inverse (field.cuh:925-972 / field_host.cuh:404-472; 0 -> 0)
        ffi.check(lib.panda_debug_field_op(fid, op, dr.ptr, da.ptr, db.ptr, n, NULL_STREAM), "op")
        got = dr.to_host().reshape(n, lc)
        assert (got == po.f_vec(fid, op, a, b)).all(), (fid, op)
    for d in (da, db, dr):
        d.free()


@pytest.mark.parametrize("cid", [0, 1, 2])
def test_curve_ops_elementwise(cid):
    lc = po.LC_Q[cid]
    n = 512
    c = pyref.CURVES[cid]
    g = po.generator(cid)
    scal = po.gen_scalars(po.FR_OF[cid], 21, n)
    jac = np.stack([po.scalar_mul(cid, g, pyref.int_to_limbs(pyref.limbs_to_int(s) % c.r, 8)) for s in scal])
    bases = po.gen_bases(cid, 22, n)
    aff = np.stack([po.to_affine(cid, j) for j in jac])
    neg = aff.copy()
    neg[:, lc:] = po.f_vec(po.FQ_OF[cid], po.OP_SUB, np.zeros_like(neg[:, lc:]), neg[:, lc:])
    ident = np.zeros_like(jac)
    bz = bases.copy()
    bz[::5, :lc] = 0
    lib = ffi.load()

    def run(op, A, B):
        dA, dB, dR = DeviceBuffer.from_host(A), DeviceBuffer.from_host(B), DeviceBuffer(A.nbytes)
        ffi.check(lib.panda_debug_curve_op(cid, op, dR.ptr, dA.ptr, dB.ptr, n, NULL_STREAM), "op")
        r = dR.to_host().reshape(n, 3 * lc)
        for d in (dA, dB, dR):
            d.free()
        return r

    def same(got, want):
        for x, y in zip(got, want):
            assert (po.to_affine(cid, x) == po.to_affine(cid, y)).all()
            assert (not x[2 * lc:].any()) == (not y[2 * lc:].any())

    for A, B in ((jac, bases), (jac, aff), (jac, neg), (ident, bases), (jac, bz)):  # incl. P+P, P+(-P), identities
        same(run(0, A, B), po.curve_vec(cid, po.COP_MADD, A, B))
    negj = jac.copy()
    negj[:, lc:2 * lc] = neg[:, lc:] if False else po.f_vec(po.FQ_OF[cid], po.OP_SUB, np.zeros_like(jac[:, lc:2 * lc]), jac[:, lc:2 * lc])
    for A, B in ((jac, jac[::-1].copy()), (jac, jac), (jac, negj), (ident, jac), (jac, ident)):
        same(run(1, A, B), po.curve_vec(cid, po.COP_ADD, A, B))
        same(run(3, A, B), po.curve_vec(cid, po.COP_ADD, A, B))  # four lanes per addition (curve29_quad.h: the MSM's reduction trees)
    same(run(2, jac, jac), po.curve_vec(cid, po.COP_DBL, jac))
    same(run(4, jac, jac), po.curve_vec(cid, po.COP_DBL, jac))
    same(run(4, ident, ident), po.curve_vec(cid, po.COP_DBL, ident))


@pytest.mark.parametrize("cid", [0, 1, 2])
def test_device_generators_match_oracle(cid):
    n = 3000
    lib = ffi.load()
    ds = DeviceBuffer(n * 32)
    ffi.check(lib.panda_gen_scalars(cid, 0xABC, 7, n, ds.ptr, NULL_STREAM), "gen")
    assert (ds.to_host().reshape(n, 8) == po.gen_scalars(po.FR_OF[cid], 0xABC, n, first=7)).all()
    db = DeviceBuffer(n * 2 * po.LC_Q[cid] * 4)
    ffi.check(lib.panda_gen_bases(cid, 0xDEF, 5, n, db.ptr, NULL_STREAM), "gen")
    assert (db.to_host().reshape(n, -1) == po.gen_bases(cid, 0xDEF, n, first=5)).all()
    ds.free()
    db.free()


# ------------------------------------------------------------------ MSM

def test_msm_k13_reference_golden(gm, golden_dir):
    """The reference's own fixture (src/cuda/test/data/msm/k13): 8192 x generator, all P + P."""
    scalars = np.fromfile(os.path.join(golden_dir, "ref_k13_scalars.bin"), dtype=np.uint32).reshape(-1, 8)
    want = np.fromfile(os.path.join(golden_dir, "ref_k13_result_affine.bin"), dtype=np.uint32)
    bases = np.tile(po.generator(0), (8192, 1))
    keep = scalars.copy()
    out = pgm.panda_msm_bn254_gpu(gm, scalars, bases)
    assert (affine_of(0, out) == want).all()
    assert (scalars == keep).all()  # the caller's scalars are not de-Montgomeryed in place


@pytest.mark.parametrize("k", [10, 11, 12, 13, 14, 16])
def test_msm_bn254_correctness_device(gm, k):
    """tests/test.rs:50-112 with the oracle in place of ark: random points and scalars, affine equality."""
    n = 1 << k
    bases = po.gen_bases(0, 100 + k, n)
    scalars = po.gen_scalars(po.F_BN254_FR, 200 + k, n)
    out = pgm.panda_msm_bn254_gpu(gm, scalars, bases)
    if k <= 14:
        want = po.msm_affine(0, bases, scalars, window_bits=10)
    else:
        # BASELINE config 1: 2^16 random scalars/bases against the reference's host-debug algorithm as the oracle restates
        # it (16-bit windows, msm_host.cuh:267-370), and against the linearity identity
        want = po.msm_affine(0, bases, scalars, window_bits=16, threads=8)
        assert (want == po.expected_from_linearity(0, 100 + k, scalars)).all()
    assert (affine_of(0, out) == want).all()
    if k == 16:  # the product's own CPU entry point at the same size (test_msm_bn254_correctness_host, tests/test.rs:115-194 runs k = 10..16)
        host = pgm.panda_msm_bn254_gpu_host(gm, scalars, bases)
        assert (affine_of(0, host) == want).all()


def test_msm_bn254_correctness_host_entry(gm):
    """tests/test.rs:115-194: the CPU host-debug entry point of the library (not the oracle) against the oracle."""
    for k in (10, 12):
        n = 1 << k
        bases = po.gen_bases(0, 300 + k, n)
        scalars = po.gen_scalars(po.F_BN254_FR, 400 + k, n)
        keep = scalars.copy()
        out = pgm.panda_msm_bn254_gpu_host(gm, scalars, bases)
        assert (affine_of(0, out) == po.msm_affine(0, bases, scalars, window_bits=9)).all()
        assert (scalars == keep).all()


EDGE_SETS = ["zeros", "ones", "minus_one", "small", "all_equal", "half_zero", "top_bit"]


@pytest.mark.parametrize("kind", EDGE_SETS)
def test_msm_edge_scalar_sets(gm, kind):
    """SURVEY 8d robustness sets: heavy bucket skew and empty windows."""
    c = pyref.CURVES[0]
    n = 1 << 12
    bases = po.gen_bases(0, 55, n)
    rng = np.random.default_rng(9)
    mont = lambda v: pyref.int_to_limbs(v * c.Rr % c.r, 8)
    if kind == "zeros":
        scalars = np.zeros((n, 8), np.uint32)
    elif kind == "ones":
        scalars = np.tile(mont(1), (n, 1))
    elif kind == "minus_one":
        scalars = np.tile(mont(c.r - 1), (n, 1))
    elif kind == "small":
        scalars = np.stack([mont(int(v)) for v in rng.integers(0, 1 << 16, n)])
    elif kind == "all_equal":
        scalars = np.tile(po.gen_scalars(po.F_BN254_FR, 77, 1), (n, 1))
    elif kind == "half_zero":
        scalars = po.gen_scalars(po.F_BN254_FR, 78, n)
        scalars[::2] = 0
    else:
        scalars = np.stack([mont((1 << 253) + int(v)) for v in rng.integers(0, 1 << 30, n)])
    out = pgm.panda_msm_bn254_gpu(gm, scalars, bases)
    want = po.expected_from_linearity(0, 55, scalars)
    got = affine_of(0, out)
    assert (got == want).all()
    if kind == "zeros":
        assert not out.view(np.uint32)[16:].any()  # Z == 0: the identity


def test_msm_degenerate_bases(gm):
    """identity bases (x == 0), P and -P with the same scalar, repeated points."""
    n = 1 << 10
    lc = 8
    bases = po.gen_bases(0, 66, n)
    scalars = po.gen_scalars(po.F_BN254_FR, 67, n)
    bases[3::7, :lc] = 0
    bases[100:200] = bases[100]
    bases[301] = bases[300]
    bases[301, lc:] = po.f_vec(0, po.OP_SUB, np.zeros((1, lc), np.uint32), bases[300:301, lc:])[0]
    scalars[301] = scalars[300]
    out = pgm.panda_msm_bn254_gpu(gm, scalars, bases)
    assert (affine_of(0, out) == po.msm_affine(0, bases, scalars, window_bits=9)).all()


def test_msm_cached_variants_and_projective(gm):
    """unit.rs:103-361 (cached bases / scalars / both) and set_config(Projective) (wrapper.rs:212-214)."""
    n = 1 << 12
    bases = po.gen_bases(0, 88, n)
    scalars = po.gen_scalars(po.F_BN254_FR, 89, n)
    want = po.expected_from_linearity(0, 88, scalars)
    gm.d_bases.append(pgm.PandaGpuManager.init_msm_cached_bases(bases))
    gm.d_scalars.append(pgm.PandaGpuManager.init_msm_cached_scalars(scalars))
    gm.scalars_len.append(scalars.nbytes)
    bi, si = len(gm.d_bases) - 1, len(gm.d_scalars) - 1
    assert (affine_of(0, pgm.panda_msm_bn254_gpu_with_cached_bases(gm, scalars, bi)) == want).all()
    # cached scalars are reusable: twice, then together with cached bases
    for _ in range(2):
        assert (affine_of(0, pgm.panda_msm_bn254_gpu_with_cached_scalars(gm, si, bases)) == want).all()
    assert (affine_of(0, pgm.panda_msm_bn254_gpu_with_cached_input(gm, si, bi)) == want).all()
    gm.set_config(pgm.PROJECTIVE)
    try:
        out = pgm.panda_msm_bn254_gpu_with_cached_input(gm, si, bi)
        assert (affine_of(0, out, pgm.PROJECTIVE) == want).all()
    finally:
        gm.set_config(pgm.JACOBIAN)


@pytest.mark.parametrize("wbits", [5, 8, 13, 16])
def test_msm_window_override(gm, wbits):
    n = 1 << 11
    bases = po.gen_bases(0, 91, n)
    scalars = po.gen_scalars(po.F_BN254_FR, 92, n)
    lib = ffi.load()
    ffi.check(lib.panda_msm_set_window_bits(wbits), "cfg")
    try:
        out = pgm.panda_msm_bn254_gpu(gm, scalars, bases)
    finally:
        lib.panda_msm_set_window_bits(0)
    assert (affine_of(0, out) == po.expected_from_linearity(0, 91, scalars)).all()


@pytest.mark.parametrize("cid,k,wbits,kind", [(0, 14, 17, "uniform"), (0, 16, 19, "uniform"), (0, 15, 20, "small"), (0, 16, 20, "top_bit"), (0, 13, 18, "equal"),
                                              (1, 14, 19, "uniform"), (2, 13, 20, "uniform")])
def test_msm_plain_path_wide_windows(gm, cid, k, wbits, kind):
    """The plain drop-in call (nothing registered) with windows wider than 16 bits: u32 digit codes and the three-level sort with a
    bucket space per window (what the policy picks from 2^22 points on; forced here at sizes the suite affords).  Result checked by
    MSM(s, m*G) = (sum s_i m_i)*G, scalar sets incl. the top-window carry and skew into a single bucket."""
    lib = ffi.load()
    n = 1 << k
    db, ds, dr = DeviceBuffer(n * 2 * po.LC_Q[cid] * 4), DeviceBuffer(n * 32), DeviceBuffer(3 * po.LC_Q[cid] * 4)
    ffi.check(lib.panda_gen_bases(cid, 7700 + k, 0, n, db.ptr, NULL_STREAM), "gen")
    c = pyref.CURVES[cid]
    rng = np.random.default_rng(k)
    mont = lambda v: pyref.int_to_limbs(v * c.Rr % c.r, 8)
    scalars = po.gen_scalars(po.FR_OF[cid], 7800 + k, n)
    if kind == "small":
        scalars = np.stack([mont(int(v)) for v in rng.integers(0, 1 << 16, n)])
    elif kind == "top_bit":  # r - small: every window's digit carries into the next, the top window included
        scalars = np.stack([mont(c.r - 1 - int(v)) for v in rng.integers(0, 1 << 20, n)])
    elif kind == "equal":
        scalars[:] = scalars[0]
    scalars = np.ascontiguousarray(scalars)
    ffi.check(lib.panda_memcpy(ds.ptr, C.c_void_p(scalars.ctypes.data), n * 32), "memcpy")
    cfg = ffi.MSMConfiguration(gm.mem_pool, gm.exec_stream.raw, db.ptr, ds.ptr, dr.ptr, k, pgm.JACOBIAN)
    fn = (lib.panda_msm_execute_bn254, lib.panda_msm_execute_bls12_377, lib.panda_msm_execute_bls12_381)[cid]
    ffi.check(lib.panda_msm_set_window_bits(wbits), "cfg")
    try:
        ffi.check(fn(cfg), "msm")
        assert (ds.to_host().reshape(n, 8) == scalars).all()  # scalars untouched
        got = po.to_affine(cid, dr.to_host())
        # and through registered (converted, untabled) bases + the in-call upload pipeline over point ranges
        ffi.check(lib.panda_msm_register_bases(cid, db.ptr, k, gm.exec_stream.raw), "register")
        ffi.check(fn(cfg), "msm")
        got_reg = po.to_affine(cid, dr.to_host())
        ffi.check(lib.panda_msm_unregister_bases(db.ptr), "unregister")
    finally:
        lib.panda_msm_set_window_bits(0)
    want = po.expected_from_linearity(cid, 7700 + k, scalars)
    assert (got == want).all() and (got_reg == want).all()
    for d in (db, ds, dr):
        d.free()


@pytest.mark.parametrize("k", [10, 13])
def test_msm_bls12_377(gm, k):
    n = 1 << k
    bases = po.gen_bases(1, 500 + k, n)
    scalars = po.gen_scalars(po.F_BLS377_FR, 600 + k, n)
    for coord in (pgm.JACOBIAN, pgm.PROJECTIVE):
        gm.set_config(coord)
        try:
            out = pgm.panda_msm_bn254_gpu(gm, scalars, bases, curve=pgm.BLS12_377)
        finally:
            gm.set_config(pgm.JACOBIAN)
        assert out.size == 144
        assert (affine_of(1, out, coord) == po.expected_from_linearity(1, 500 + k, scalars)).all()
    if k == 10:
        assert (affine_of(1, out, pgm.PROJECTIVE) == po.msm_affine(1, bases, scalars, window_bits=8)).all()


def _msm_on_device_inputs(gm, cid, k, seed_b, seed_s):
    n = 1 << k
    lib = ffi.load()
    db = DeviceBuffer(n * 2 * po.LC_Q[cid] * 4)
    ds = DeviceBuffer(n * 32)
    dr = DeviceBuffer(3 * po.LC_Q[cid] * 4)
    ffi.check(lib.panda_gen_bases(cid, seed_b, 0, n, db.ptr, NULL_STREAM), "gen")
    ffi.check(lib.panda_gen_scalars(cid, seed_s, 0, n, ds.ptr, NULL_STREAM), "gen")
    cfg = ffi.MSMConfiguration(gm.mem_pool, gm.exec_stream.raw, db.ptr, ds.ptr, dr.ptr, k, pgm.JACOBIAN)
    fn = (lib.panda_msm_execute_bn254, lib.panda_msm_execute_bls12_377, lib.panda_msm_execute_bls12_381)[cid]
    ffi.check(fn(cfg), "msm")
    out = dr.to_host()
    scalars = ds.to_host().reshape(n, 8)
    for d in (db, ds, dr):
        d.free()
    return out, scalars


def test_msm_bn254_2_20_linearity(gm):
    """BASELINE config 2 (BN254 2^20, Jacobian, bases resident): inputs generated in HBM, checked by the
    size-independent identity MSM(s, m*G) = (sum s_i m_i)*G."""
    out, scalars = _msm_on_device_inputs(gm, 0, 20, 0x70616E6461 ^ 2, 0x5CA1A5)
    assert (po.to_affine(0, out) == po.expected_from_linearity(0, 0x70616E6461 ^ 2, scalars)).all()


def test_msm_bls12_377_2_16_linearity(gm):
    out, scalars = _msm_on_device_inputs(gm, 1, 16, 0x70616E6461 ^ 5, 0x5CA1A6)
    assert (po.to_affine(1, out) == po.expected_from_linearity(1, 0x70616E6461 ^ 5, scalars)).all()


def test_msm_bases_in_a_virtual_memory_mapping(gm):
    """Caller buffers are checked against log_n at the boundary, but only against allocations made through the library itself: for memory
    mapped with hipMemAddressReserve / hipMemMap (torch's expandable segments) the runtime's own description of a pointer may be a single
    mapped chunk, and a base buffer that legitimately spans two chunks must not be refused (panda_internal.h)."""
    try:
        hip = C.CDLL("libamdhip64.so")
    except OSError:
        pytest.skip("no HIP runtime library to map memory with")

    class Loc(C.Structure):
        _fields_ = [("type", C.c_int), ("id", C.c_int)]

    class Prop(C.Structure):
        _fields_ = [("type", C.c_int), ("requestedHandleType", C.c_int), ("location", Loc), ("win32HandleMetaData", C.c_void_p),
                    ("compressionType", C.c_ubyte), ("gpuDirectRDMACapable", C.c_ubyte), ("usage", C.c_ushort)]

    class Access(C.Structure):
        _fields_ = [("location", Loc), ("flags", C.c_int)]

    prop = Prop(1, 0, Loc(1, 0), None, 0, 0, 0)  # hipMemAllocationTypePinned, hipMemLocationTypeDevice, device 0
    gran = C.c_size_t(0)
    if hip.hipMemGetAllocationGranularity(C.byref(gran), C.byref(prop), 0) != 0 or gran.value == 0:
        pytest.skip("virtual memory management is not available on this device")
    k = 15
    n = 1 << k
    while n * 64 <= gran.value:  # the bases must span two chunks
        k, n = k + 1, n << 1
    chunk = ((n * 64 // 2 + gran.value - 1) // gran.value) * gran.value
    va = C.c_void_p()
    assert hip.hipMemAddressReserve(C.byref(va), C.c_size_t(2 * chunk), C.c_size_t(0), None, C.c_ulonglong(0)) == 0
    handles = []
    try:
        for j in range(2):
            h = C.c_void_p()
            assert hip.hipMemCreate(C.byref(h), C.c_size_t(chunk), C.byref(prop), C.c_ulonglong(0)) == 0
            handles.append(h)
            assert hip.hipMemMap(C.c_void_p(va.value + j * chunk), C.c_size_t(chunk), C.c_size_t(0), h, C.c_ulonglong(0)) == 0
        acc = Access(Loc(1, 0), 3)  # hipMemAccessFlagsProtReadWrite
        assert hip.hipMemSetAccess(va, C.c_size_t(2 * chunk), C.byref(acc), C.c_size_t(1)) == 0
        lib = ffi.load()
        ds, dr = DeviceBuffer(n * 32), DeviceBuffer(96)
        ffi.check(lib.panda_gen_bases(0, 0x7A7A, 0, n, va, NULL_STREAM), "gen")
        ffi.check(lib.panda_gen_scalars(0, 0x7A7B, 0, n, ds.ptr, NULL_STREAM), "gen")
        cfg = ffi.MSMConfiguration(gm.mem_pool, gm.exec_stream.raw, va, ds.ptr, dr.ptr, k, pgm.JACOBIAN)
        ffi.check(lib.panda_msm_execute_bn254(cfg), "msm")
        assert (po.to_affine(0, dr.to_host()) == po.expected_from_linearity(0, 0x7A7A, ds.to_host().reshape(n, 8))).all()
        ds.free()
        dr.free()
    finally:
        hip.hipDeviceSynchronize()
        for j, h in enumerate(handles):
            hip.hipMemUnmap(C.c_void_p(va.value + j * chunk), C.c_size_t(chunk))
            hip.hipMemRelease(h)
        hip.hipMemAddressFree(va, C.c_size_t(2 * chunk))


def test_msm_phase_timers(gm):
    """device timers are off by default (an event between two kernels costs GPU idle time); level 1 times the total and k_accumulate,
    level 2 every phase"""
    n = 1 << 12
    lib = ffi.load()
    ms = (C.c_float * 8)()
    seen = {}
    try:
        for level in (0, 1, 2):
            ffi.check(lib.panda_msm_set_phase_timing(level), "cfg")
            pgm.panda_msm_bn254_gpu(gm, po.gen_scalars(po.F_BN254_FR, 1, n), po.gen_bases(0, 2, n))
            ffi.check(lib.panda_msm_last_phase_ms(ms), "phase")
            seen[level] = list(ms)
    finally:
        lib.panda_msm_set_phase_timing(0)
    assert all(v == 0 for v in seen[0])
    assert seen[1][7] > 0 and seen[1][3] > 0 and seen[1][1] == 0
    assert all(v >= 0 for v in seen[2]) and seen[2][7] > 0 and seen[2][1] > 0
    assert lib.panda_msm_set_phase_timing(3) != 0
    assert lib.panda_msm_phase_name(3) == b"accumulate"


# ------------------------------------------------------------------ NTT

def ntt_passes(log_n):
    """passes of a natural-order transform (the library's plan: eight bits per pass as fft.cu:171-216, or radix-512 passes in front)"""
    passes, bits = C.c_uint(0), (C.c_uint * 4)()
    ffi.check(ffi.load().panda_ntt_pass_plan(log_n, C.byref(passes), bits), "plan")
    assert sum(bits) == log_n and all(b <= 9 for b in bits)
    return passes.value


def test_ntt_pass_plan():
    want = {17: 2, 18: 2, 25: 3, 26: 3, 27: 3}
    for log_n in range(0, 29):
        assert ntt_passes(log_n) == want.get(log_n, (log_n + 7) // 8), log_n
    assert ffi.load().panda_ntt_pass_plan(29, C.byref(C.c_uint(0)), None) != 0


@pytest.mark.parametrize("log_n", [0, 1, 2, 3, 5, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20])
def test_ntt_v1_vs_oracle(gm, log_n):
    fid = po.F_BN254_FR
    n = 1 << log_n
    om = po.root_of_unity(fid, log_n)
    x = po.gen_scalars(fid, 3000 + log_n, n)
    want = po.ntt(fid, x, om, log_n)
    buf = x.copy()
    flag = pgm.panda_ntt_bn254_gpu_v1(gm, buf, om, log_n)
    assert flag == ntt_passes(log_n) % 2  # *flag = passes & 1 (fft.cu:211); the reference's loop runs ceil(log_n / 8) passes, 2^17 / 2^18 here one less
    assert (buf == want).all()
    # inverse with fused n^-1 returns the input
    back = buf.copy()
    pgm.panda_intt_bn254_gpu(gm, back, om, log_n)
    assert (back == x).all()


@pytest.mark.parametrize("log_n", [0, 1, 3, 7, 8, 9, 11, 12, 16, 17, 18, 19, 20])
def test_ntt_bit_reversed_orderings(gm, log_n):
    """forward with bit-reversed output (y[k] at bitrev(k)) equals the oracle's natural-order transform permuted;
    the inverse from bit-reversed input returns the coefficients; the flag protocol is unchanged."""
    fid = po.F_BN254_FR
    n = 1 << log_n
    om = po.root_of_unity(fid, log_n)
    x = po.gen_scalars(fid, 3100 + log_n, n)
    perm = np.array([int(format(k, f"0{log_n}b")[::-1], 2) if log_n else 0 for k in range(n)])
    want = po.ntt(fid, x, om, log_n)
    buf = x.copy()
    flag = pgm.panda_ntt_bn254_gpu_bitrev(gm, buf, om, log_n)
    assert flag == (ntt_passes(log_n) if log_n not in (18, 27) else (log_n + 7) // 8) % 2  # 2^18 / 2^27 keep the eight-bit plan in these orderings
    assert (buf[perm] == want).all()  # buf[bitrev(k)] = y[k]
    pgm.panda_ntt_bn254_gpu_bitrev(gm, buf, om, log_n, inverse=True)
    assert (buf == x).all()
    # the inverse alone, from the oracle's output permuted on the host
    buf = np.ascontiguousarray(want[perm])  # position bitrev(k) holds y[k] (the permutation is an involution)
    pgm.panda_ntt_bn254_gpu_bitrev(gm, buf, om, log_n, inverse=True)
    assert (buf == x).all()


@pytest.mark.parametrize("cid,log_n", [(0, 19), (0, 20), (0, 21), (1, 20), (2, 20), (0, 23), (0, 25)])
def test_ntt_streamed_inter_pass_table(gm, cid, log_n):
    """panda_ntt_set_streamed_tables: the second boundary of a three-pass transform multiplies by ONE entry of a table over the whole
    index range (a Montgomery product) instead of two 2^16-entry tables' entries -- same outputs, forward and inverse, for the three
    scalar fields (BLS12-381's R / p = 70 is the tight one); the twiddle cache keeps the two kinds of table apart."""
    lib = ffi.load()
    fid = po.FR_OF[cid]
    n = 1 << log_n
    om = po.root_of_unity(fid, log_n)
    fwd = (lib.panda_ntt_execute_bn254_v1, lib.panda_ntt_execute_bls12_377_v1, lib.panda_ntt_execute_bls12_381_v1)[cid]
    inv = (lib.panda_ntt_execute_bn254_inverse, lib.panda_ntt_execute_bls12_377_inverse, lib.panda_ntt_execute_bls12_381_inverse)[cid]
    d_a, d_b = DeviceBuffer(n * 32), DeviceBuffer(n * 32)
    ffi.check(lib.panda_gen_scalars(cid, 0x57AB + log_n, 0, n, d_a.ptr, NULL_STREAM), "gen")
    x = d_a.to_host().reshape(n, 8)
    flag = C.c_uint(9)
    outs = []
    try:
        for on in ((0, 2) if log_n > 24 else (0, 1, 0, 1)):  # 2^25 / 2^26: the 1 / 2 GiB tables are opt-in (mode 2); one round each (1 GiB per comparison)
            ffi.check(lib.panda_ntt_set_streamed_tables(on), "option")
            ffi.check(lib.panda_memcpy(d_a.ptr, C.c_void_p(x.ctypes.data), n * 32), "memcpy")
            cfg = ffi.NttconfigurationV1(gm.mem_pool, gm.exec_stream.raw, d_a.ptr, d_b.ptr, C.c_void_p(om.ctypes.data), log_n, C.pointer(flag))
            ffi.check(fwd(cfg), "ntt")
            assert flag.value == ntt_passes(log_n) % 2
            f, o = (d_b, d_a) if flag.value else (d_a, d_b)
            outs.append(f.to_host().reshape(n, 8))
            cfg2 = ffi.NttconfigurationV1(gm.mem_pool, gm.exec_stream.raw, f.ptr, o.ptr, C.c_void_p(om.ctypes.data), log_n, C.pointer(flag))
            ffi.check(inv(cfg2), "intt")
            back = (o if flag.value else f).to_host().reshape(n, 8)
            assert np.array_equal(back, x), on
    finally:
        lib.panda_ntt_set_streamed_tables(0xFFFFFFFF)  # the built-in policy (mode 1), not "off"
    assert all(np.array_equal(outs[0], o) for o in outs[1:])
    if log_n <= 20:
        assert np.array_equal(outs[1], po.ntt(fid, x, om, log_n))
    else:
        rng = np.random.default_rng(log_n)
        for k in [0, 1, n - 1] + [int(v) for v in rng.integers(0, n, 5)]:
            assert (outs[1][k] == po.ntt_eval_at(fid, x, om, log_n, k)).all(), k
    d_a.free()
    d_b.free()


def test_ntt_streamed_table_policy_and_out_of_memory_fallback(gm):
    """The default policy builds no streamed table beyond 2^24 entries (a 2^25-point transform under mode 1 and mode 0 builds the same
    tables); and a streamed table that cannot be allocated (mode 3: the allocation is treated as failed) is given up ONCE: the call
    succeeds with the two small tables, and the next calls hit the cache instead of rebuilding (the fallback used to store its
    tables under a key the next call never looked up)."""
    lib = ffi.load()
    fid, log_n = po.F_BN254_FR, 19  # 8 + 8 + 3 bits: the middle pass takes a streamed table of 2^19 entries
    n = 1 << log_n
    om = po.root_of_unity(fid, log_n)
    x = po.gen_scalars(fid, 0xFA11, n)
    want = po.ntt(fid, x, om, log_n)
    builds = C.c_uint64(0)

    def count():
        ffi.check(lib.panda_ntt_table_builds(C.byref(builds)), "builds")
        return builds.value

    def run():
        buf = x.copy()
        pgm.panda_ntt_bn254_gpu_v1(gm, buf, om, log_n)
        assert (buf == want).all()

    assert lib.panda_ntt_set_streamed_tables(4) == 1
    ffi.check(lib.panda_ntt_set_streamed_tables(3), "option")
    b0 = count()
    run()
    assert count() == b0 + 1  # built once, without the streamed table
    run()
    run()
    assert count() == b0 + 1  # hits
    pgm.panda_intt_bn254_gpu(gm, want.copy(), om, log_n)  # the other direction: the other cache slot, its own fallback, once
    run()
    assert count() == b0 + 2
    ffi.check(lib.panda_ntt_set_streamed_tables(0xFFFFFFFF), "option")  # a new setting gets a new try: now the streamed table is built
    run()
    assert count() == b0 + 3
    run()
    assert count() == b0 + 3
    # policy: 2^25 points under mode 1 use the same (non-streamed) tables as mode 0 -- no 1 GiB table by default
    log_big = 25
    d_a, d_b = DeviceBuffer(32 << log_big), DeviceBuffer(32 << log_big)
    ffi.check(lib.panda_gen_scalars(0, 0xB16, 0, 1 << log_big, d_a.ptr, NULL_STREAM), "gen")
    omb = po.root_of_unity(fid, log_big)
    flag = C.c_uint(9)
    free0, free1, total = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
    cfg = ffi.NttconfigurationV1(gm.mem_pool, gm.exec_stream.raw, d_a.ptr, d_b.ptr, C.c_void_p(omb.ctypes.data), log_big, C.pointer(flag))
    ffi.check(lib.panda_ntt_tear_down(), "tear_down")
    ffi.check(lib.panda_mem_get_info(C.byref(free0), C.byref(total)), "mem_info")
    ffi.check(lib.panda_ntt_execute_bn254_v1(cfg), "ntt")
    ffi.check(lib.panda_mem_get_info(C.byref(free1), C.byref(total)), "mem_info")
    assert free0.value - free1.value < (256 << 20), "the default policy must not hold a table as large as a 2^25-point transform"
    ffi.check(lib.panda_ntt_tear_down(), "tear_down")
    d_a.free()
    d_b.free()


def test_clock_stamps_around_accumulate_and_ntt(gm):
    """panda_set_clock_stamps: the marker kernels around k_accumulate (MSM) and around the passes (NTT) report shader cycles and 10 ns
    ticks from all eight XCDs; cycles / ticks x 100 MHz is a plausible shader clock, the ticks agree with the HIP-event time of the same
    launches, results are unchanged, and with the knob off nothing is stamped."""
    lib = ffi.load()
    k = 18
    n = 1 << k
    db, ds, dr = DeviceBuffer(n * 64), DeviceBuffer(n * 32), DeviceBuffer(96)
    ffi.check(lib.panda_gen_bases(0, 0xC10C, 0, n, db.ptr, NULL_STREAM), "gen")
    ffi.check(lib.panda_gen_scalars(0, 0xC10D, 0, n, ds.ptr, NULL_STREAM), "gen")
    scratch = DeviceBuffer(32 << 22)
    ffi.check(lib.panda_msm_precompute_bases(0, db.ptr, k, 0, NULL_STREAM), "precompute")
    cfg = ffi.MSMConfiguration(gm.mem_pool, gm.exec_stream.raw, db.ptr, ds.ptr, dr.ptr, k, pgm.JACOBIAN)
    out, ph = (C.c_uint64 * ffi.CLOCK_WORDS)(), (C.c_float * 8)()
    try:
        ffi.check(lib.panda_msm_execute_bn254(cfg), "msm")
        plain = dr.to_host()
        ffi.check(lib.panda_msm_last_clock(out), "clock")
        assert list(out) == [0] * ffi.CLOCK_WORDS
        ffi.check(lib.panda_set_clock_stamps(1), "stamps")
        ffi.check(lib.panda_msm_set_phase_timing(1), "timing")
        for _ in range(3):
            ffi.check(lib.panda_msm_execute_bn254(cfg), "msm")
        assert (affine_of(0, dr.to_host()) == affine_of(0, plain)).all()  # (the Jacobian triple itself differs from run to run)
        ffi.check(lib.panda_msm_last_clock(out), "clock")
        ffi.check(lib.panda_msm_last_phase_ms(ph), "phases")
        cycles, ticks, xcds, mean = out[0], out[1], out[2], out[3]
        assert xcds == 8 and cycles > 0 and ticks > 0
        per = list(out[4:12])
        assert min(per) == cycles and cycles <= mean <= max(per) and max(per) < 1.2 * cycles, per  # the XCDs' clocks differ by a few per cent, not more
        mhz = mean / ticks * 100.0
        assert 900.0 < mhz < 2700.0, mhz
        assert abs(ticks * 1e-5 - ph[3]) < 0.05 + 0.1 * ph[3], (ticks * 1e-5, ph[3])  # 10 ns ticks vs the events around the same launch
        # NTT
        log_n = 20
        om = po.root_of_unity(po.F_BN254_FR, log_n)
        x = po.gen_scalars(po.F_BN254_FR, 0xC10E, 1 << log_n)
        buf = x.copy()
        pgm.panda_ntt_bn254_gpu_v1(gm, buf, om, log_n)
        buf = x.copy()
        pgm.panda_ntt_bn254_gpu_v1(gm, buf, om, log_n)
        assert (buf == po.ntt(po.F_BN254_FR, x, om, log_n)).all()
        ms = C.c_float(0)
        ffi.check(lib.panda_ntt_last_clock(out), "clock")
        ffi.check(lib.panda_ntt_last_device_ms(C.byref(ms)), "ms")
        assert out[2] == 8 and 900.0 < out[3] / out[1] * 100.0 < 2700.0 and max(out[4:12]) < 1.2 * out[0]
        assert abs(out[1] * 1e-5 - ms.value) < 0.05 + 0.1 * ms.value
        # the generic marker: two stamps on a stream with a known amount of work between them
        sb = ffi.CLOCK_STAMP_BYTES
        blocks = DeviceBuffer(2 * sb)
        ffi.check(lib.panda_memset(blocks.ptr, 0, 2 * sb), "memset")
        ffi.check(lib.panda_clock_stamp(gm.exec_stream.raw, blocks.ptr), "stamp")
        ffi.check(lib.panda_gen_scalars(0, 1, 0, 1 << 22, scratch.ptr, gm.exec_stream.raw), "work")
        ffi.check(lib.panda_clock_stamp(gm.exec_stream.raw, C.c_void_p(blocks.ptr.value + sb)), "stamp")
        ffi.check(lib.panda_stream_sync(gm.exec_stream.raw), "sync")
        host = blocks.to_host(np.uint64)
        ffi.check(lib.panda_clock_delta(C.c_void_p(host.ctypes.data), C.c_void_p(host.ctypes.data + sb), out), "delta")
        assert out[2] == 8 and out[0] > 0 and 900.0 < out[3] / out[1] * 100.0 < 2700.0
        ffi.check(lib.panda_set_clock_stamps(0), "stamps")
        ffi.check(lib.panda_msm_execute_bn254(cfg), "msm")
        ffi.check(lib.panda_msm_last_clock(out), "clock")
        assert list(out) == [0] * ffi.CLOCK_WORDS
        blocks.free()
    finally:
        lib.panda_msm_unregister_bases(db.ptr)
        for d in (db, ds, dr, scratch):
            d.free()


def test_ntt_setup_then_execute(gm):
    """init_ntt + panda_ntt_bn254_gpu (wrapper.rs:199-210, unit.rs:418-479): omega from the global setup."""
    fid, log_n = po.F_BN254_FR, 11
    om = po.root_of_unity(fid, log_n)
    pgm.PandaGpuManager.init_ntt(om)
    x = po.gen_scalars(fid, 42, 1 << log_n)
    buf = x.copy()
    pgm.panda_ntt_bn254_gpu(gm, buf, log_n)
    assert (buf == po.ntt(fid, x, om, log_n)).all()


@pytest.mark.parametrize("cid", [0, 1])
def test_ntt_2_24_values_and_full_roundtrip(gm, cid):
    """BASELINE config 3 (2^24 forward + inverse) at full size over BN254 Fr and over BLS12-377 Fr (north_star: "NTT butterfly over
    BN254/BLS12-377"; SURVEY 8d "correctness at sizes the CPU cannot reach"):
      * forward VALUES: y[k] for k = 0, 1, n/2, n-1 and 8 seeded random k evaluated directly from the definition
        y[k] = sum_j x[j] w^(jk) in O(n) each by the oracle (po_ntt_eval_at) and compared with the device output;
      * the forward transform of a delta x = e_j must be w^(jk): 64 random k against Python integers;
      * the WHOLE inverse(forward(x)) buffer equals x, compared byte for byte (2^24 x 32 B), not a sample."""
    fid, log_n = po.FR_OF[cid], 24
    n = 1 << log_n
    lib = ffi.load()
    fwd_fn, inv_fn = ((lib.panda_ntt_execute_bn254_v1, lib.panda_ntt_execute_bn254_inverse), (lib.panda_ntt_execute_bls12_377_v1, lib.panda_ntt_execute_bls12_377_inverse))[cid]
    om = po.root_of_unity(fid, log_n)
    c = pyref.CURVES[cid]
    rng = np.random.default_rng(0x24 + cid)
    d_a, d_b = DeviceBuffer(n * 32), DeviceBuffer(n * 32)
    ffi.check(lib.panda_gen_scalars(cid, 0x1234, 0, n, d_a.ptr, NULL_STREAM), "gen")
    x = d_a.to_host().reshape(n, 8)
    assert (x[:64] == po.gen_scalars(fid, 0x1234, 64)).all() and (x[-64:] == po.gen_scalars(fid, 0x1234, 64, first=n - 64)).all()
    flag = C.c_uint(9)
    cfg = ffi.NttconfigurationV1(gm.mem_pool, gm.exec_stream.raw, d_a.ptr, d_b.ptr, C.c_void_p(om.ctypes.data), log_n, C.pointer(flag))
    ffi.check(fwd_fn(cfg), "ntt")
    assert flag.value == 1  # three passes (fft.cu:193-211)
    fwd, other = (d_b, d_a) if flag.value else (d_a, d_b)
    ks = [0, 1, n - 1] + [int(v) for v in rng.integers(0, n, 3)]  # O(n) on the host each (~1.3 s): six outputs here, 64 more positions through the identities below
    for k in ks:
        got = fwd.to_host(nbytes=32, offset=k * 32)
        assert (got == po.ntt_eval_at(fid, x, om, log_n, k)).all(), k
    cfg2 = ffi.NttconfigurationV1(gm.mem_pool, gm.exec_stream.raw, fwd.ptr, other.ptr, C.c_void_p(om.ctypes.data), log_n, C.pointer(flag))
    ffi.check(inv_fn(cfg2), "intt")
    res = other if flag.value else fwd
    back = res.to_host().reshape(n, 8)
    assert np.array_equal(back, x)
    del back, x
    # delta input at a random position j: y[k] = w^(jk), in Montgomery form
    w = pyref.decode_scalar(c, om)
    j = int(rng.integers(1, n))
    ffi.check(lib.panda_memset(d_a.ptr, 0, n * 32), "memset")
    one = pyref.int_to_limbs(c.Rr % c.r, 8)
    ffi.check(lib.panda_memcpy(C.c_void_p(d_a.ptr.value + j * 32), C.c_void_p(one.ctypes.data), 32), "memcpy")
    ffi.check(fwd_fn(cfg), "ntt")
    fwd = d_b if flag.value else d_a
    for k in [0, 1, n - 1] + [int(v) for v in rng.integers(0, n, 61)]:
        want = pyref.int_to_limbs(pow(w, (j * k) % n, c.r) * c.Rr % c.r, 8)
        assert (fwd.to_host(nbytes=32, offset=k * 32) == want).all(), (j, k)
    d_a.free()
    d_b.free()


@pytest.mark.parametrize("log_ranks,log_n", [(1, 6), (2, 12), (3, 15)])
def test_ntt_slab_steps_single_gpu(gm, log_ranks, log_n):
    """The two device halves of the sharded NTT (panda_ntt_slab_step{1,2}_bn254), with the all-to-all done on the host:
    every rank's slab goes through the same GPU one after the other; the assembled output must be the plain NTT."""
    from panda_amd import multi_gpu
    fid = po.F_BN254_FR
    G, n = 1 << log_ranks, 1 << log_n
    m = n // G
    om = po.root_of_unity(fid, log_n)
    x = po.gen_scalars(fid, 0x51AB, n)
    lib = ffi.load()
    after1 = []
    for r in range(G):
        d_slab, d_scr = DeviceBuffer.from_host(multi_gpu.slab_of(x, G, r)), DeviceBuffer(m * 32)
        flag = C.c_uint(7)
        cfg = ffi.NttSlabConfiguration(gm.exec_stream.raw, d_slab.ptr, d_scr.ptr, C.c_void_p(om.ctypes.data), log_n, log_ranks, r, C.pointer(flag))
        ffi.check(lib.panda_ntt_slab_step1_bn254(cfg), "step1")
        after1.append((d_scr if flag.value else d_slab).to_host().reshape(m, 8))
        d_slab.free()
        d_scr.free()
    outs = []
    for q in range(G):
        recv = np.concatenate([after1[j1][q * (m // G):(q + 1) * (m // G)] for j1 in range(G)])  # what all_to_all_single delivers
        d_slab, d_scr = DeviceBuffer.from_host(recv), DeviceBuffer(m * 32)
        flag = C.c_uint(7)
        cfg = ffi.NttSlabConfiguration(gm.exec_stream.raw, d_slab.ptr, d_scr.ptr, C.c_void_p(om.ctypes.data), log_n, log_ranks, q, C.pointer(flag))
        ffi.check(lib.panda_ntt_slab_step2_bn254(cfg), "step2")
        outs.append((d_scr if flag.value else d_slab).to_host().reshape(m, 8))
        d_slab.free()
        d_scr.free()
    assert (multi_gpu.natural_from_slab_outputs(outs) == po.ntt(fid, x, om, log_n)).all()


@pytest.mark.parametrize("log_ranks,log_n", [(1, 9), (2, 14), (3, 18), (3, 21)])
def test_ntt_sharded_composed_on_device(gm, log_ranks, log_n):
    """multi_gpu.ntt_sharded's composition (step 1 -> all-to-all -> step 2) with every buffer on the device: the G ranks are
    played one after the other on this GPU and the exchange is device-to-device copies (ntt_sharded_one_process), so the
    flag / buffer protocol either side of the exchange meets the real kernels.  Output vs the oracle's plain transform."""
    import torch
    fid = po.F_BN254_FR
    G, n = 1 << log_ranks, 1 << log_n
    m = n // G
    om = po.root_of_unity(fid, log_n)
    x = po.gen_scalars(fid, 0x51AC + log_n, n)
    dev = torch.device("cuda", 0)
    slabs = [torch.from_numpy(multi_gpu.slab_of(x, G, r).view(np.uint8).reshape(-1).copy()).to(dev) for r in range(G)]
    scratches = [torch.empty_like(t) for t in slabs]
    outs = multi_gpu.ntt_sharded_one_process(slabs, scratches, om, log_n)
    y = multi_gpu.natural_from_slab_outputs([o.cpu().numpy().view(np.uint32).reshape(m, 8) for o in outs])
    assert (y == po.ntt(fid, x, om, log_n)).all()
    # and back: the mirrored steps take the forward output layout to the decimated input layout, n^-1 included
    spare = [torch.empty_like(t) for t in outs]
    back = multi_gpu.intt_sharded_one_process([o.clone() for o in outs], spare, om, log_n)
    for r in range(G):
        assert (back[r].cpu().numpy().view(np.uint32).reshape(m, 8) == multi_gpu.slab_of(x, G, r)).all(), r


def test_msm_config4_partition_8_ranges_of_2_23(gm):
    """BASELINE config 4's partition on the HIP path: 2^26 points cut into 8 contiguous base ranges of 2^23, each through
    panda_msm_execute_bn254 (as each rank of bench.py --gpus 8 does), the 8 Jacobian partials through
    panda_msm_combine_bn254; the total is checked by linearity over all 2^26 scalars."""
    k, parts = 26, 8
    n, per = 1 << k, (1 << k) // parts
    lib = ffi.load()
    db, ds, dr = DeviceBuffer(n * 64), DeviceBuffer(n * 32), DeviceBuffer(96)
    seed_b, seed_s = 0x70616E6461 ^ 0xC4, 0xFEED + 0xC4
    ffi.check(lib.panda_gen_bases(0, seed_b, 0, n, db.ptr, NULL_STREAM), "gen")
    ffi.check(lib.panda_gen_scalars(0, seed_s, 0, n, ds.ptr, NULL_STREAM), "gen")
    partials = np.empty((parts, 24), np.uint32)
    for g in range(parts):
        cfg = ffi.MSMConfiguration(gm.mem_pool, gm.exec_stream.raw, C.c_void_p(db.ptr.value + g * per * 64), C.c_void_p(ds.ptr.value + g * per * 32),
                                   dr.ptr, k - 3, pgm.JACOBIAN)
        ffi.check(lib.panda_msm_execute_bn254(cfg), "msm")
        partials[g] = dr.to_host()
    total = multi_gpu.combine_partials(partials)
    scalars = ds.to_host().reshape(n, 8)
    for d in (db, ds, dr):
        d.free()
    assert (po.to_affine(0, total.view(np.uint32)) == po.expected_from_linearity(0, seed_b, scalars)).all()
    # and each partial alone is the MSM of its own range (first = g * 2^23 of the same seeded stream)
    for g in (0, 5):
        want = po.expected_from_linearity(0, seed_b, scalars[g * per:(g + 1) * per], first=g * per)
        assert (po.to_affine(0, partials[g]) == want).all()


@pytest.mark.parametrize("ranks,transport", [(1, ffi.MULTI_RCCL), (1, ffi.MULTI_LOOPBACK), (4, ffi.MULTI_LOOPBACK), (8, ffi.MULTI_LOOPBACK)])
def test_c_abi_multi_gpu_msm(gm, ranks, transport):
    """panda_msm_execute_bn254_multi (csrc/multi_gpu.hip): base ranges, one worker thread per rank, partials gathered (ncclAllGather on
    the RCCL transport -- one rank on this one-GPU box --, device copies on the loopback transport, where cuda:0 plays every rank) and
    combined.  Total by linearity over all scalars, Jacobian and homogeneous; called twice (the workers' arenas are reused)."""
    k = 16
    n, per = 1 << k, (1 << k) // ranks
    lib = ffi.load()
    seed_b, seed_s = 0xC0DE + ranks, 0xFACE + ranks
    db, ds = DeviceBuffer(n * 64), DeviceBuffer(n * 32)
    results = [DeviceBuffer(96) for _ in range(ranks)]
    ffi.check(lib.panda_gen_bases(0, seed_b, 0, n, db.ptr, NULL_STREAM), "gen")
    ffi.check(lib.panda_gen_scalars(0, seed_s, 0, n, ds.ptr, NULL_STREAM), "gen")
    scalars = ds.to_host().reshape(n, 8)
    want = po.expected_from_linearity(0, seed_b, scalars)
    mg = multi_gpu.MultiGpu([0] * ranks, transport)
    try:
        for coord in (pgm.JACOBIAN, pgm.PROJECTIVE, pgm.JACOBIAN):
            cfgs = [ffi.MSMConfiguration(ffi.PandaMemPool(), ffi.PandaStream(), C.c_void_p(db.ptr.value + r * per * 64), C.c_void_p(ds.ptr.value + r * per * 32),
                                         results[r].ptr, k - (ranks.bit_length() - 1), coord) for r in range(ranks)]
            total = mg.msm(cfgs)
            assert (affine_of(0, total, coord) == want).all()
        # each rank's buffer holds the Jacobian partial of its own range
        r = ranks - 1
        assert (po.to_affine(0, results[r].to_host()) == po.expected_from_linearity(0, seed_b, scalars[r * per:(r + 1) * per], first=r * per)).all()
        lib.panda_msm_set_phase_timing(1)
        try:
            mg.msm(cfgs)
            assert mg.phases(r)[7] > 0
        finally:
            lib.panda_msm_set_phase_timing(0)
        bad = [ffi.MSMConfiguration(ffi.PandaMemPool(), ffi.PandaStream(), None, ds.ptr, results[0].ptr, 4, pgm.JACOBIAN)] * ranks
        with pytest.raises(ffi.PandaGpuError):
            mg.msm(bad)  # a failing rank fails the call, the handle stays usable
        assert (affine_of(0, mg.msm(cfgs), pgm.JACOBIAN) == want).all()
    finally:
        mg.close()
        for d in [db, ds] + results:
            d.free()


@pytest.mark.parametrize("ranks,transport,pinned,tabled", [(1, ffi.MULTI_RCCL, True, True), (2, ffi.MULTI_LOOPBACK, True, True), (4, ffi.MULTI_LOOPBACK, False, True),
                                                          (8, ffi.MULTI_LOOPBACK, True, False), (8, ffi.MULTI_LOOPBACK, False, True)])
def test_c_abi_multi_gpu_msm_from_host(gm, ranks, transport, pinned, tabled):
    """panda_msm_execute_bn254_from_host_multi: the scalars start on the HOST (pinned, or pageable) and every rank's worker uploads its
    own shard inside the call, in point ranges beside its kernels (registered bases: with tables and without; 2^17 points per rank so
    that the ranges are real).  Total by linearity over all scalars; the device staging buffers end up holding the scalars."""
    k_per = 17
    per, n = 1 << k_per, ranks << k_per
    lib = ffi.load()
    seed_b, seed_s = 0xBA5E + ranks, 0x5CA1 + ranks
    db = DeviceBuffer(n * 64)
    ffi.check(lib.panda_gen_bases(0, seed_b, 0, n, db.ptr, NULL_STREAM), "gen")
    scalars = po.gen_scalars(po.F_BN254_FR, seed_s, n)
    want = po.expected_from_linearity(0, seed_b, scalars)
    host_ptr = C.c_void_p()
    if pinned:
        ffi.check(lib.panda_malloc_host(C.byref(host_ptr), n * 32), "malloc_host")
        C.memmove(host_ptr, scalars.ctypes.data, n * 32)
        host0 = host_ptr.value
    else:
        host0 = scalars.ctypes.data
    staging = [DeviceBuffer(per * 32) for _ in range(ranks)]
    results = [DeviceBuffer(96) for _ in range(ranks)]
    base_ptrs = [C.c_void_p(db.ptr.value + r * per * 64) for r in range(ranks)]
    for bp in base_ptrs:
        if tabled:
            ffi.check(lib.panda_msm_precompute_bases(0, bp, k_per, 0, NULL_STREAM), "precompute")
        else:
            ffi.check(lib.panda_msm_register_bases(0, bp, k_per, NULL_STREAM), "register")
    mg = multi_gpu.MultiGpu([0] * ranks, transport)
    try:
        for coord, ranges in ((pgm.JACOBIAN, 2), (pgm.PROJECTIVE, 1), (pgm.JACOBIAN, 4)):
            cfgs = [ffi.MSMConfiguration(ffi.PandaMemPool(), ffi.PandaStream(), base_ptrs[r], staging[r].ptr, results[r].ptr, k_per, coord) for r in range(ranks)]
            total = mg.msm_from_host(cfgs, [host0 + r * per * 32 for r in range(ranks)], ranges)
            assert (affine_of(0, total, coord) == want).all()
        r = ranks - 1
        assert (staging[r].to_host().reshape(per, 8) == scalars[r * per:(r + 1) * per]).all()
        with pytest.raises(ffi.PandaGpuError):
            mg.msm_from_host(cfgs, [host0] * (ranks - 1) + [0], 2)  # a missing host source is refused before anything is enqueued
        assert (affine_of(0, mg.msm(cfgs), pgm.JACOBIAN) == want).all()  # the resident-scalars call on what the upload left behind
    finally:
        mg.close()
        for bp in base_ptrs:
            lib.panda_msm_unregister_bases(bp)
        if pinned:
            lib.panda_free_host(host_ptr)
        for d in [db] + staging + results:
            d.free()


@pytest.mark.parametrize("ranks,transport,log_n", [(1, ffi.MULTI_RCCL, 12), (2, ffi.MULTI_LOOPBACK, 9), (4, ffi.MULTI_LOOPBACK, 14), (8, ffi.MULTI_LOOPBACK, 21)])
def test_c_abi_multi_gpu_ntt(gm, ranks, transport, log_n):
    """panda_ntt_execute_bn254_multi / _inverse_multi: step 1 -> all-to-all -> step 2 behind one C call (grouped ncclSend / ncclRecv on
    the RCCL transport, device copies on the loopback one); output layout and flags as for multi_gpu.ntt_sharded; forward vs the
    oracle's plain transform, inverse back to the decimated input slabs."""
    fid = po.F_BN254_FR
    n = 1 << log_n
    m = n // ranks
    om = po.root_of_unity(fid, log_n)
    x = po.gen_scalars(fid, 0x51AD + log_n, n)
    slabs = [DeviceBuffer.from_host(multi_gpu.slab_of(x, ranks, r)) for r in range(ranks)]
    scratches = [DeviceBuffer(m * 32) for _ in range(ranks)]
    mg = multi_gpu.MultiGpu([0] * ranks, transport)
    try:
        for _ in range(2):  # the second round hits the workers' twiddle caches
            for r in range(ranks):
                part = multi_gpu.slab_of(x, ranks, r)  # held while the copy runs
                ffi.check(ffi.load().panda_memcpy(slabs[r].ptr, C.c_void_p(part.ctypes.data), m * 32), "copy")
            flags = mg.ntt([b.ptr.value for b in slabs], [b.ptr.value for b in scratches], om, log_n)
            outs = [(scratches[r] if flags[r] else slabs[r]).to_host().reshape(m, 8) for r in range(ranks)]
            assert (multi_gpu.natural_from_slab_outputs(outs) == po.ntt(fid, x, om, log_n)).all()
        if ranks > 1 or log_n >= 2:
            src = [(scratches[r], slabs[r]) if flags[r] else (slabs[r], scratches[r]) for r in range(ranks)]
            back = mg.ntt([a.ptr.value for a, _ in src], [b.ptr.value for _, b in src], om, log_n, inverse=True)
            for r in range(ranks):
                got = (src[r][1] if back[r] else src[r][0]).to_host().reshape(m, 8)
                assert (got == multi_gpu.slab_of(x, ranks, r)).all(), r
    finally:
        mg.close()
        for d in slabs + scratches:
            d.free()


@pytest.mark.parametrize("ranks,transport", [(1, ffi.MULTI_RCCL), (4, ffi.MULTI_LOOPBACK)])
def test_c_abi_multi_gpu_msm_other_curves(gm, ranks, transport):
    """panda_msm_execute_bls12_381_multi / panda_msm_execute_bn254_g2_multi and their *_from_host_multi forms (VERDICT r3 missing 7): the same base-range
    sharding with 144- and 192-byte partials.  BLS12-381 G1 by linearity over all scalars, BN254 G2 against the Python reference's expectation."""
    k = 12
    n, per = 1 << k, (1 << k) // ranks
    lib = ffi.load()
    mg = multi_gpu.MultiGpu([0] * ranks, transport)
    bufs = []
    try:
        # BLS12-381 G1 (curve 2): device-generated bases, resident scalars and scalars from pageable host memory
        seed_b, seed_s = 0x3810 + ranks, 0x3811 + ranks
        db, ds = DeviceBuffer(n * 96), DeviceBuffer(n * 32)
        res = [DeviceBuffer(144) for _ in range(ranks)]
        bufs += [db, ds] + res
        ffi.check(lib.panda_gen_bases(2, seed_b, 0, n, db.ptr, NULL_STREAM), "gen")
        ffi.check(lib.panda_gen_scalars(2, seed_s, 0, n, ds.ptr, NULL_STREAM), "gen")
        scalars = ds.to_host().reshape(n, 8)
        want = po.expected_from_linearity(2, seed_b, scalars)
        for coord in (pgm.JACOBIAN, pgm.PROJECTIVE):
            cfgs = [ffi.MSMConfiguration(ffi.PandaMemPool(), ffi.PandaStream(), C.c_void_p(db.ptr.value + r * per * 96), C.c_void_p(ds.ptr.value + r * per * 32),
                                         res[r].ptr, k - (ranks.bit_length() - 1), coord) for r in range(ranks)]
            total = mg.msm(cfgs, curve=2)
            assert total.size == 144 and (affine_of(2, total, coord) == want).all()
        staging = [DeviceBuffer(per * 32) for _ in range(ranks)]
        bufs += staging
        cfgs = [ffi.MSMConfiguration(ffi.PandaMemPool(), ffi.PandaStream(), C.c_void_p(db.ptr.value + r * per * 96), staging[r].ptr, res[r].ptr,
                                     k - (ranks.bit_length() - 1), pgm.JACOBIAN) for r in range(ranks)]
        host = np.ascontiguousarray(scalars)
        total = mg.msm_from_host(cfgs, [host.ctypes.data + r * per * 32 for r in range(ranks)], 2, curve=2)
        assert (affine_of(2, total) == want).all()
        # BN254 G2 (curve 3)
        g2b = _g2_device_bases(0xE700 + ranks, n)
        g2s = po.gen_scalars(po.F_BN254_FR, 0xE701 + ranks, n)
        want2 = _g2_expected(0xE700 + ranks, g2s)
        d2b, d2s = DeviceBuffer.from_host(g2b), DeviceBuffer.from_host(g2s)
        res2 = [DeviceBuffer(192) for _ in range(ranks)]
        bufs += [d2b, d2s] + res2
        for coord in (pgm.JACOBIAN, pgm.PROJECTIVE):
            cfgs = [ffi.MSMConfiguration(ffi.PandaMemPool(), ffi.PandaStream(), C.c_void_p(d2b.ptr.value + r * per * 128), C.c_void_p(d2s.ptr.value + r * per * 32),
                                         res2[r].ptr, k - (ranks.bit_length() - 1), coord) for r in range(ranks)]
            total = mg.msm(cfgs, curve=3)
            assert total.size == 192 and _g2_decode(total, coord) == want2
        stag2 = [DeviceBuffer(per * 32) for _ in range(ranks)]
        bufs += stag2
        cfgs = [ffi.MSMConfiguration(ffi.PandaMemPool(), ffi.PandaStream(), C.c_void_p(d2b.ptr.value + r * per * 128), stag2[r].ptr, res2[r].ptr,
                                     k - (ranks.bit_length() - 1), pgm.JACOBIAN) for r in range(ranks)]
        total = mg.msm_from_host(cfgs, [g2s.ctypes.data + r * per * 32 for r in range(ranks)], 2, curve=3)
        assert _g2_decode(total, pgm.JACOBIAN) == want2
    finally:
        mg.close()
        for d in bufs:
            d.free()


@pytest.mark.parametrize("ranks,transport,log_n,count", [(1, ffi.MULTI_RCCL, 12, 3), (2, ffi.MULTI_LOOPBACK, 10, 4), (4, ffi.MULTI_LOOPBACK, 14, 3), (8, ffi.MULTI_LOOPBACK, 20, 5)])
def test_c_abi_multi_gpu_ntt_batch(gm, ranks, transport, log_n, count):
    """panda_ntt_execute_bn254_multi_batch / _inverse_multi_batch: `count` sharded transforms of different inputs pipelined over a compute and an
    exchange stream per device (the all-to-all of transform t behind step 1 of t + 1 and step 2 of t - 1).  Every transform against the oracle's
    plain transform, the inverse batch back to the decimated input slabs; run twice (the second round reuses the handle's events and the workers'
    twiddle caches)."""
    fid = po.F_BN254_FR
    n = 1 << log_n
    m = n // ranks
    om = po.root_of_unity(fid, log_n)
    xs = [po.gen_scalars(fid, 0xBA7C + 16 * t + log_n, n) for t in range(count)]
    wants = [po.ntt(fid, x, om, log_n) for x in xs]
    slabs = [[DeviceBuffer(m * 32) for _ in range(ranks)] for _ in range(count)]
    scratches = [[DeviceBuffer(m * 32) for _ in range(ranks)] for _ in range(count)]
    mg = multi_gpu.MultiGpu([0] * ranks, transport)
    try:
        for _ in range(2):
            for t in range(count):
                for r in range(ranks):
                    part = multi_gpu.slab_of(xs[t], ranks, r)  # held while the copy runs
                    ffi.check(ffi.load().panda_memcpy(slabs[t][r].ptr, C.c_void_p(part.ctypes.data), m * 32), "copy")
            flags = mg.ntt_batch([[b.ptr.value for b in row] for row in slabs], [[b.ptr.value for b in row] for row in scratches], om, log_n)
            for t in range(count):
                outs = [(scratches[t][r] if flags[t][r] else slabs[t][r]).to_host().reshape(m, 8) for r in range(ranks)]
                assert (multi_gpu.natural_from_slab_outputs(outs) == wants[t]).all(), t
            src = [[(scratches[t][r], slabs[t][r]) if flags[t][r] else (slabs[t][r], scratches[t][r]) for r in range(ranks)] for t in range(count)]
            back = mg.ntt_batch([[a.ptr.value for a, _ in row] for row in src], [[b.ptr.value for _, b in row] for row in src], om, log_n, inverse=True)
            for t in range(count):
                for r in range(ranks):
                    got = (src[t][r][1] if back[t][r] else src[t][r][0]).to_host().reshape(m, 8)
                    assert (got == multi_gpu.slab_of(xs[t], ranks, r)).all(), (t, r)
        lib = ffi.load()
        assert lib.panda_ntt_execute_bn254_multi_batch(mg.handle, None, 1) == 1  # panda_error_invalid_value
        one = (ffi.NttSlabConfiguration * ranks)(*[ffi.NttSlabConfiguration(ffi.PandaStream(), slabs[0][r].ptr, scratches[0][r].ptr, C.c_void_p(om.ctypes.data), log_n,
                                                                              ranks.bit_length() - 1, r, None) for r in range(ranks)])
        assert lib.panda_ntt_execute_bn254_multi_batch(mg.handle, one, 0) == 1
    finally:
        mg.close()
        for row in slabs + scratches:
            for d in row:
                d.free()


class _OnDevice:
    """allocations, generators and copies of one rank run with that rank's device current"""

    def __init__(self, dev):
        self.dev, self.prev = dev, C.c_int(0)

    def __enter__(self):
        lib = ffi.load()
        ffi.check(lib.panda_get_device(C.byref(self.prev)), "get_device")
        ffi.check(lib.panda_set_device(self.dev), "set_device")

    def __exit__(self, *exc):
        ffi.load().panda_set_device(self.prev.value)


def _box_devices():
    """every device of the box, up to eight, rounded down to a power of two (the slab decomposition wants one)"""
    n = C.c_int(0)
    ffi.check(ffi.load().panda_get_device_number(C.byref(n)), "device_number")
    g = max(1, min(n.value, 8))
    return list(range(1 << (g.bit_length() - 1)))


def _sharded_over(devs, transport, what):
    """One sharded operation through the single-process C entry points with rank r's buffers ON devs[r], checked as the one-device tests
    check it: MSM by linearity over every rank's scalars, NTT against the oracle's plain transform and back through the inverse."""
    lib = ffi.load()
    G = len(devs)
    g = G.bit_length() - 1
    bufs = []

    def dbuf(r, nbytes=None, host=None):
        with _OnDevice(devs[r]):
            b = DeviceBuffer.from_host(host) if host is not None else DeviceBuffer(nbytes)
        bufs.append((devs[r], b))
        return b

    def fetch(r, b, shape):
        with _OnDevice(devs[r]):
            return b.to_host().reshape(shape)

    mg = multi_gpu.MultiGpu(devs, transport)
    pinned = []
    try:
        if what in ("msm", "msm_from_host"):
            k = 15
            per = (1 << k) >> g
            seed_b, seed_s = 0xB0C5 + G, 0xB0C6 + G
            db, ds, dr = [dbuf(r, per * 64) for r in range(G)], [dbuf(r, per * 32) for r in range(G)], [dbuf(r, 96) for r in range(G)]
            for r in range(G):
                with _OnDevice(devs[r]):
                    ffi.check(lib.panda_gen_bases(0, seed_b, r * per, per, db[r].ptr, NULL_STREAM), "gen")
                    ffi.check(lib.panda_gen_scalars(0, seed_s, r * per, per, ds[r].ptr, NULL_STREAM), "gen")
                    if what == "msm_from_host":
                        ffi.check(lib.panda_msm_precompute_bases(0, db[r].ptr, k - g, 0, NULL_STREAM), "precompute")
            scalars = np.concatenate([fetch(r, ds[r], (per, 8)) for r in range(G)])
            want = po.expected_from_linearity(0, seed_b, scalars)
            cfgs = [ffi.MSMConfiguration(ffi.PandaMemPool(), ffi.PandaStream(), db[r].ptr, ds[r].ptr, dr[r].ptr, k - g, pgm.JACOBIAN) for r in range(G)]
            if what == "msm":
                for _ in range(2):
                    assert (affine_of(0, mg.msm(cfgs)) == want).all()
                r = G - 1  # every rank's own buffer holds the partial of its own range
                assert (po.to_affine(0, fetch(r, dr[r], (24,))) == po.expected_from_linearity(0, seed_b, scalars[r * per:], first=r * per)).all()
            else:
                hosts = []
                for r in range(G):
                    hp = C.c_void_p()
                    ffi.check(lib.panda_malloc_host(C.byref(hp), per * 32), "malloc_host")
                    pinned.append(hp)
                    C.memmove(hp, scalars[r * per:(r + 1) * per].ctypes.data, per * 32)
                    with _OnDevice(devs[r]):
                        ffi.check(lib.panda_memset(ds[r].ptr, 0, per * 32), "memset")  # the device copies must come from the host
                    hosts.append(hp.value)
                for ranges in (1, 3):
                    assert (affine_of(0, mg.msm_from_host(cfgs, hosts, ranges)) == want).all()
                for r in range(G):
                    with _OnDevice(devs[r]):
                        lib.panda_msm_unregister_bases(db[r].ptr)
        else:
            field = {"ntt_bls12_377": 1, "ntt_bls12_381": 2}.get(what, 0)
            fid = po.FR_OF[field]
            log_n = 13 + g
            n, m = 1 << log_n, (1 << log_n) >> g
            om = po.root_of_unity(fid, log_n)
            count = 3 if what == "ntt_batch" else 1
            xs = [po.gen_scalars(fid, 0xB0C7 + 16 * t + G, n) for t in range(count)]
            slabs = [[dbuf(r, host=multi_gpu.slab_of(xs[t], G, r)) for r in range(G)] for t in range(count)]
            scr = [[dbuf(r, m * 32) for r in range(G)] for t in range(count)]
            ptr = lambda rows: [[b.ptr.value for b in row] for row in rows]
            if what == "ntt_batch":
                flags = mg.ntt_batch(ptr(slabs), ptr(scr), om, log_n)
            else:
                flags = [mg.ntt(ptr(slabs)[0], ptr(scr)[0], om, log_n, field=field)]
            outs = [[(scr[t][r] if flags[t][r] else slabs[t][r], slabs[t][r] if flags[t][r] else scr[t][r]) for r in range(G)] for t in range(count)]
            for t in range(count):
                got = multi_gpu.natural_from_slab_outputs([fetch(r, outs[t][r][0], (m, 8)) for r in range(G)])
                assert (got == po.ntt(fid, xs[t], om, log_n)).all(), t
            src, oth = [[o[0].ptr.value for o in row] for row in outs], [[o[1].ptr.value for o in row] for row in outs]
            if what == "ntt_batch":
                back = mg.ntt_batch(src, oth, om, log_n, inverse=True)
            else:
                back = [mg.ntt(src[0], oth[0], om, log_n, inverse=True, field=field)]
            for t in range(count):
                for r in range(G):
                    res = outs[t][r][1] if back[t][r] else outs[t][r][0]
                    assert (fetch(r, res, (m, 8)) == multi_gpu.slab_of(xs[t], G, r)).all(), (t, r)
    finally:
        mg.close()
        for hp in pinned:
            lib.panda_free_host(hp)
        for dev, b in bufs:
            with _OnDevice(dev):
                b.free()


@pytest.mark.parametrize("what", ["msm", "msm_from_host", "ntt", "ntt_batch", "ntt_bls12_377", "ntt_bls12_381"])
@pytest.mark.parametrize("where", ["every_device_rccl", "two_loopback_ranks"])
def test_c_abi_multi_gpu_widens_with_the_box(gm, where, what):
    """The sharded entry points with one rank per DEVICE of the box over RCCL (ncclCommInitAll over several devices, the in-place
    ncclAllGather, the grouped ncclSend / ncclRecv all-to-all, the two-stream batch schedule): skipped on a one-GPU box, a parity result
    on the first box that has more.  The same body runs with two loopback ranks on device 0, so that it is itself tested everywhere."""
    if where == "every_device_rccl":
        devs = _box_devices()
        if len(devs) < 2:
            pytest.skip("one GPU on this box: RCCL with more than one rank cannot run here")
        _sharded_over(devs, ffi.MULTI_RCCL, what)
    else:
        _sharded_over([0, 0], ffi.MULTI_LOOPBACK, what)


def test_c_abi_multi_gpu_handles_from_two_threads(gm):
    """Two panda_multi_gpu handles alive at once (two sets of worker threads), each driven from its own host thread while the other is
    busy, created and destroyed three times over: every call returns the MSM of its own inputs."""
    import threading
    lib = ffi.load()
    k, ranks = 14, 2
    n, per = 1 << k, (1 << k) // ranks
    problems = []
    for t in range(2):
        seed_b, seed_s = 0xAB00 + t, 0xCD00 + t
        db, ds = DeviceBuffer(n * 64), DeviceBuffer(n * 32)
        res = [DeviceBuffer(96) for _ in range(ranks)]
        ffi.check(lib.panda_gen_bases(0, seed_b, 0, n, db.ptr, NULL_STREAM), "gen")
        ffi.check(lib.panda_gen_scalars(0, seed_s, 0, n, ds.ptr, NULL_STREAM), "gen")
        want = po.expected_from_linearity(0, seed_b, ds.to_host().reshape(n, 8))
        cfgs = [ffi.MSMConfiguration(ffi.PandaMemPool(), ffi.PandaStream(), C.c_void_p(db.ptr.value + r * per * 64), C.c_void_p(ds.ptr.value + r * per * 32),
                                     res[r].ptr, k - 1, pgm.JACOBIAN) for r in range(ranks)]
        problems.append((db, ds, res, cfgs, want))
    errors = []

    def worker(t):
        try:
            for _ in range(3):
                mg = multi_gpu.MultiGpu([0] * ranks, ffi.MULTI_LOOPBACK)
                try:
                    for _ in range(4):
                        total = mg.msm(problems[t][3])
                        if not (affine_of(0, total) == problems[t][4]).all():
                            errors.append(f"thread {t}: wrong sum")
                finally:
                    mg.close()
        except Exception as e:  # noqa: BLE001
            errors.append(f"thread {t}: {e!r}")

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    for db, ds, res, _, _ in problems:
        for d in [db, ds] + res:
            d.free()
    assert not errors, errors


def test_c_abi_multi_gpu_bad_arguments(gm):
    lib = ffi.load()
    h = ffi.PandaMultiGpu()
    one = (C.c_int * 1)(0)
    two = (C.c_int * 2)(0, 0)
    assert lib.panda_multi_gpu_create(C.byref(h), one, 0, ffi.MULTI_RCCL) == 1
    assert lib.panda_multi_gpu_create(C.byref(h), two, 2, ffi.MULTI_RCCL) == 1       # RCCL: one rank per device
    assert lib.panda_multi_gpu_create(C.byref(h), (C.c_int * 1)(99), 1, ffi.MULTI_RCCL) == 1
    assert lib.panda_multi_gpu_create(C.byref(h), one, 1, 7) == 1
    mg = multi_gpu.MultiGpu([0, 0, 0], ffi.MULTI_LOOPBACK)                           # three ranks: fine for an MSM, not for the slab NTT
    try:
        d = DeviceBuffer(3 << 12)
        om = po.root_of_unity(po.F_BN254_FR, 9)
        flag = C.c_uint(0)
        cfgs = (ffi.NttSlabConfiguration * 3)(*[ffi.NttSlabConfiguration(ffi.PandaStream(), d.ptr, d.ptr, C.c_void_p(om.ctypes.data), 9, 1, r, C.pointer(flag))
                                                for r in range(3)])
        assert lib.panda_ntt_execute_bn254_multi(mg.handle, cfgs) == 1
        assert lib.panda_ntt_execute_bn254_multi(mg.handle, None) == 1
        assert lib.panda_msm_execute_bn254_multi(mg.handle, None, None) == 1
        d.free()
    finally:
        mg.close()


def test_cpp_gpu_manager_mirror():
    """The C++ mirror of the reference's Rust gpu_manager + its integration test binary (panda_amd/csrc/tests/manager_test.cpp,
    the counterpart of tests/test.rs): device MSM vs the CPU entry point, cached variants, NTT round trips."""
    import subprocess
    for name in ("manager_test", "manager_test_static"):  # shared object, and the static archive build.rs links
        exe = os.path.join(os.path.dirname(ffi.LIB_PATH), "tests", name)
        assert os.path.exists(exe), "build it with make -C panda_amd/csrc"
        r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
        assert "manager_test: all ok" in r.stdout


@pytest.mark.parametrize("cid,k,coord", [(0, 24, pgm.JACOBIAN), (1, 24, pgm.PROJECTIVE), (0, 25, pgm.JACOBIAN), (0, 26, pgm.JACOBIAN)])
def test_msm_baseline_full_sizes(gm, cid, k, coord):
    """BASELINE.json's full sizes on one GPU: BN254 2^24 (the headline metric), BLS12-377 2^24 with Projective output
    (config 5) and BN254 2^26 (config 4's total).  No CPU MSM reaches these sizes; the size-independent check is the
    linearity identity over bases m_i*G generated in HBM."""
    n = 1 << k
    lib = ffi.load()
    lc = po.LC_Q[cid]
    db, ds, dr = DeviceBuffer(n * 2 * lc * 4), DeviceBuffer(n * 32), DeviceBuffer(3 * lc * 4)
    seed_b, seed_s = 0x70616E6461 ^ (k * 16 + cid), 0xFEED + k
    ffi.check(lib.panda_gen_bases(cid, seed_b, 0, n, db.ptr, NULL_STREAM), "gen")
    ffi.check(lib.panda_gen_scalars(cid, seed_s, 0, n, ds.ptr, NULL_STREAM), "gen")
    cfg = ffi.MSMConfiguration(gm.mem_pool, gm.exec_stream.raw, db.ptr, ds.ptr, dr.ptr, k, coord)
    ffi.check((lib.panda_msm_execute_bn254, lib.panda_msm_execute_bls12_377, lib.panda_msm_execute_bls12_381)[cid](cfg), "msm")
    out = dr.to_host()
    scalars = ds.to_host().reshape(n, 8)
    for d in (db, ds, dr):
        d.free()
    got = po.hom_to_affine(cid, out) if coord == pgm.PROJECTIVE else po.to_affine(cid, out)
    assert (got == po.expected_from_linearity(cid, seed_b, scalars)).all()


def test_committed_golden_fixtures(gm, golden_dir):
    """The HIP path against the committed fixtures (tests/golden/msm_cases.json, ntt_cases.json): seeds in, expected
    affine bytes / output digests out.  No oracle call on this path."""
    import hashlib
    import json
    import sys
    sys.path.insert(0, golden_dir)
    import make_golden
    for c in json.load(open(os.path.join(golden_dir, "msm_cases.json"))):
        cid, n = c["curve"], 1 << c["log_n"]
        bases = np.tile(po.generator(0), (n, 1)) if c.get("bases") == "all_generator" else po.gen_bases(cid, c["bases_seed"], n)
        scalars = make_golden.scalar_set(c["scalars"], cid, n, c["scalars_seed"])
        out = pgm.panda_msm_bn254_gpu(gm, scalars, bases, curve=cid)
        assert affine_of(cid, out).tobytes().hex() == c["affine_hex"], c
        assert (not out.view(np.uint32)[2 * po.LC_Q[cid]:].any()) == c["identity"]
        idx = gm.add_cached_bases(bases)  # the same fixture through the precomputed-table path
        gm.precompute_cached_bases(idx, curve=cid)
        out = pgm.panda_msm_bn254_gpu_with_cached_bases(gm, scalars, idx, curve=cid)
        assert affine_of(cid, out).tobytes().hex() == c["affine_hex"], ("tables", c)
        assert (not out.view(np.uint32)[2 * po.LC_Q[cid]:].any()) == c["identity"]
    for c in json.load(open(os.path.join(golden_dir, "ntt_cases.json"))):
        x = po.gen_scalars(po.F_BN254_FR, c["seed"], 1 << c["log_n"])
        om = np.frombuffer(bytes.fromhex(c["omega_hex"]), dtype=np.uint32).copy()
        buf = x.copy()
        pgm.panda_ntt_bn254_gpu_v1(gm, buf, om, c["log_n"])
        assert hashlib.sha256(buf.tobytes()).hexdigest() == c["sha256"], c["log_n"]
        assert buf[0].tobytes().hex() == c["first_hex"] and buf[-1].tobytes().hex() == c["last_hex"]


@pytest.mark.parametrize("k", [0, 1, 2, 3, 5, 7, 9])
def test_msm_tiny_sizes(gm, k):
    """n = 1 ... 512: degenerate geometry (one tile, one chunk, a single partition)."""
    n = 1 << k
    for cid in (0, 1, 2):
        bases = po.gen_bases(cid, 900 + k, n)
        scalars = po.gen_scalars(po.FR_OF[cid], 950 + k, n)
        out = pgm.panda_msm_bn254_gpu(gm, scalars, bases, curve=cid)
        assert (affine_of(cid, out) == po.to_affine(cid, po.msm_naive(cid, bases, scalars))).all()


@pytest.mark.parametrize("log_n", [0, 1, 3, 8, 10, 11, 12, 16, 17, 18, 19, 20])
def test_ntt_bls12_377_fr(gm, log_n):
    """The NTT kernels instantiated for the BLS12-377 scalar field (bls12_377/paramter.cuh:130-181; protocol fft.cu:171-216): every kind of
    plan -- the LDS kernel below 2^11, radix-256 passes with short and long last passes, the radix-512 plans of 2^17 / 2^18, three passes
    with two-factor inter-pass tables at 2^19 / 2^20 -- against the oracle, forward and inverse."""
    fid = po.F_BLS377_FR
    om = po.root_of_unity(fid, log_n)
    x = po.gen_scalars(fid, 4000 + log_n, 1 << log_n)
    buf = x.copy()
    flag = pgm.panda_ntt_bls12_377_gpu_v1(gm, buf, om, log_n)
    assert flag == ntt_passes(log_n) % 2
    assert (buf == po.ntt(fid, x, om, log_n)).all()
    pgm.panda_ntt_bls12_377_gpu_v1(gm, buf, om, log_n, inverse=True)
    assert (buf == x).all()


@pytest.mark.parametrize("cid,log_n", [(1, 0), (1, 3), (1, 9), (1, 12), (1, 17), (1, 18), (1, 20), (2, 3), (2, 12), (2, 17), (2, 20)])
def test_ntt_bls12_377_bit_reversed_orderings_and_coset(gm, cid, log_n):
    """panda_ntt_execute_bls12_{377,381}_{bitrev_out,inverse_bitrev_in,coset,coset_inverse}: the BN254 variants' semantics over the two BLS scalar fields."""
    fid = po.FR_OF[cid]
    c = pyref.CURVES[cid]
    field = ("", "bls12_377", "bls12_381")[cid]
    n = 1 << log_n
    om = po.root_of_unity(fid, log_n)
    x = po.gen_scalars(fid, 4300 + log_n, n)
    want = po.ntt(fid, x, om, log_n)
    perm = np.array([int(format(k, f"0{log_n}b")[::-1], 2) if log_n else 0 for k in range(n)])
    buf = x.copy()
    pgm.panda_ntt_bls12_377_gpu_bitrev(gm, buf, om, log_n, field=field)
    assert (buf[perm] == want).all()  # buf[bitrev(k)] = y[k]
    pgm.panda_ntt_bls12_377_gpu_bitrev(gm, buf, om, log_n, inverse=True, field=field)
    assert (buf == x).all()
    if log_n <= 12:  # coset: y = NTT(x[j] g^j), g^j by Python integers
        g = pyref.int_to_limbs(7 * c.Rr % c.r, 8)
        pw = np.stack([pyref.int_to_limbs(pow(7, j, c.r) * c.Rr % c.r, 8) for j in range(n)])
        buf = x.copy()
        pgm.panda_coset_ntt_bls12_377_gpu(gm, buf, om, g, log_n, field=field)
        assert (buf == po.ntt(fid, po.f_vec(fid, po.OP_MUL, x, pw), om, log_n)).all()
        pgm.panda_coset_ntt_bls12_377_gpu(gm, buf, om, g, log_n, inverse=True, field=field)
        assert (buf == x).all()


def test_msm_and_ntt_from_two_host_threads():
    """The library is callable from any host thread (SURVEY 8b threading): scratch is per thread, MSM has no global state."""
    import threading
    results, errors = {}, []

    def worker(tid):
        try:
            g = pgm.PandaGpuManager(0)
            for rep in range(3):
                n = 1 << 13
                bases = po.gen_bases(0, 7000 + tid, n)
                scalars = po.gen_scalars(po.F_BN254_FR, 7100 + tid * 10 + rep, n)
                out = pgm.panda_msm_bn254_gpu(g, scalars, bases)
                results[(tid, rep)] = (affine_of(0, out), po.expected_from_linearity(0, 7000 + tid, scalars))
                om = po.root_of_unity(po.F_BN254_FR, 12)
                x = po.gen_scalars(po.F_BN254_FR, 7200 + tid, 1 << 12)
                buf = x.copy()
                pgm.panda_ntt_bn254_gpu_v1(g, buf, om, 12)
                results[(tid, rep, "ntt")] = (buf, po.ntt(po.F_BN254_FR, x, om, 12))
            g.deinit()
        except Exception as e:  # pragma: no cover
            errors.append(repr(e))

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    assert len(results) == 12
    for got, want in results.values():
        assert (got == want).all()


def test_ntt_twiddle_cache_reuse_and_invalidation(gm):
    """Twiddle tables are cached per host thread between calls: same root twice (hit), then another primitive root of the
    same size, then the inverse with the first root (misses) -- every result must still match the oracle."""
    fid, log_n = po.F_BN254_FR, 13
    c = pyref.CURVES[0]
    om = po.root_of_unity(fid, log_n)
    w = pyref.decode_scalar(c, om)
    om3 = pyref.int_to_limbs(pow(w, 3, c.r) * c.Rr % c.r, 8)  # another primitive 2^13-th root
    for seed, root in ((1, om), (2, om), (3, om3), (4, om), (5, om3), (6, om3)):
        x = po.gen_scalars(fid, 8000 + seed, 1 << log_n)
        buf = x.copy()
        pgm.panda_ntt_bn254_gpu_v1(gm, buf, root, log_n)
        assert (buf == po.ntt(fid, x, root, log_n)).all(), seed
        if seed in (2, 5):
            pgm.panda_intt_bn254_gpu(gm, buf, root, log_n)
            assert (buf == x).all()
            pgm.panda_intt_bn254_gpu(gm, buf, root, log_n)  # inverse again on a hit: x scaled through the inverse transform
            om_inv = po.f_vec(fid, po.OP_INV, root[None])[0]
            n_inv = pyref.int_to_limbs(pow(1 << log_n, -1, c.r) * c.Rr % c.r, 8)
            assert (buf == po.f_scale(fid, po.ntt(fid, x, om_inv, log_n), n_inv)).all()


def test_msm_registered_cached_bases(gm):
    """panda_msm_register_bases: same answers with the conversion cached; a different buffer is unaffected; unregister works."""
    lib = ffi.load()
    n = 1 << 13
    for cid in (0, 1):
        bases = po.gen_bases(cid, 9100 + cid, n)
        other = po.gen_bases(cid, 9200 + cid, n)
        idx = gm.add_cached_bases(bases)
        gm.register_cached_bases(idx, curve=cid)
        for rep in range(3):
            scalars = po.gen_scalars(po.FR_OF[cid], 9300 + rep, n)
            out = pgm.panda_msm_bn254_gpu_with_cached_bases(gm, scalars, idx, curve=cid)
            assert (affine_of(cid, out) == po.expected_from_linearity(cid, 9100 + cid, scalars)).all()
            out2 = pgm.panda_msm_bn254_gpu(gm, scalars, other, curve=cid)  # unregistered path still converts per call
            assert (affine_of(cid, out2) == po.expected_from_linearity(cid, 9200 + cid, scalars)).all()
    assert lib.panda_msm_unregister_bases(C.c_void_p(12345)) != 0


@pytest.mark.parametrize("tabled", [False, True])
def test_msm_registration_does_not_outlive_its_buffer(gm, tabled):
    """A registration is keyed on the raw device address.  (1) panda_free drops it, so a new allocation at the same address
    with other bases gets the right answer; (2) a buffer recycled behind the library's back (overwritten in place, as a
    caching allocator would hand it out again) is caught by the rows sampled at registration: the stale entry is dropped
    and the call answered from the buffer as it is now."""
    lib = ffi.load()
    k = 12
    n = 1 << k
    old, new = po.gen_bases(0, 9970, n), po.gen_bases(0, 9971, n)
    scalars = po.gen_scalars(po.F_BN254_FR, 9972, n)
    ds, dr = DeviceBuffer.from_host(scalars), DeviceBuffer(96)
    reg = lib.panda_msm_precompute_bases if tabled else lib.panda_msm_register_bases
    args = (0, gm.exec_stream.raw) if tabled else (gm.exec_stream.raw,)

    def run(db):
        cfg = ffi.MSMConfiguration(gm.mem_pool, gm.exec_stream.raw, db.ptr, ds.ptr, dr.ptr, k, pgm.JACOBIAN)
        ffi.check(lib.panda_msm_execute_bn254(cfg), "msm")
        return po.to_affine(0, dr.to_host())

    # (1) free + reallocate
    db = DeviceBuffer.from_host(old)
    ffi.check(reg(0, db.ptr, k, *args), "register")
    assert (run(db) == po.expected_from_linearity(0, 9970, scalars)).all()
    addr = db.ptr.value
    db.free()
    assert lib.panda_msm_registered_info(C.c_void_p(addr), None, None, None) != 0  # gone with the buffer
    db2 = DeviceBuffer.from_host(new)  # usually the same address again; the result must be right either way
    assert (run(db2) == po.expected_from_linearity(0, 9971, scalars)).all()
    # (2) recycled in place while registered
    ffi.check(reg(0, db2.ptr, k, *args), "register")
    assert (run(db2) == po.expected_from_linearity(0, 9971, scalars)).all()
    ffi.check(lib.panda_memcpy(db2.ptr, C.c_void_p(old.ctypes.data), old.nbytes), "memcpy")
    assert (run(db2) == po.expected_from_linearity(0, 9970, scalars)).all()
    assert lib.panda_msm_registered_info(db2.ptr, None, None, None) != 0  # the stale entry was dropped
    # (3) ONE row changed, outside the rows sampled at registration: the per-call sample cannot see it (documented), the strict
    # check -- the hash of the whole buffer -- does: on demand, and in front of every execute in paranoid mode
    sampled = {0, n - 1} | {((t * 0x9E3779B97F4A7C15) >> 20) & (n - 1) for t in range(2, 64)}
    row = next(i for i in range(1, n) if i not in sampled)
    mixed = old.copy()
    mixed.reshape(n, -1)[row] = new.reshape(n, -1)[row]
    mixed_want = po.msm_affine(0, mixed.reshape(-1), scalars, window_bits=10)
    ffi.check(reg(0, db2.ptr, k, *args), "register")  # db2 holds `old`
    assert lib.panda_msm_verify_registered(db2.ptr, gm.exec_stream.raw) == 0
    one = np.ascontiguousarray(new.reshape(n, -1)[row])
    ffi.check(lib.panda_memcpy(C.c_void_p(db2.ptr.value + row * one.nbytes), C.c_void_p(one.ctypes.data), one.nbytes), "memcpy")
    assert lib.panda_msm_verify_registered(db2.ptr, gm.exec_stream.raw) != 0  # detected ...
    assert lib.panda_msm_registered_info(db2.ptr, None, None, None) != 0  # ... and dropped
    assert (run(db2) == mixed_want).all()
    ffi.check(lib.panda_memcpy(db2.ptr, C.c_void_p(old.ctypes.data), old.nbytes), "memcpy")
    ffi.check(reg(0, db2.ptr, k, *args), "register")
    ffi.check(lib.panda_memcpy(C.c_void_p(db2.ptr.value + row * one.nbytes), C.c_void_p(one.ctypes.data), one.nbytes), "memcpy")
    ffi.check(lib.panda_msm_set_paranoid(1), "paranoid")
    try:
        assert (run(db2) == mixed_want).all()
    finally:
        lib.panda_msm_set_paranoid(0)
    assert lib.panda_msm_registered_info(db2.ptr, None, None, None) != 0
    for d in (db2, ds, dr):
        d.free()


def test_msm_unregister_while_another_thread_executes(gm):
    """panda_msm_unregister_bases from one host thread while another is inside panda_msm_execute_bn254 with that registration:
    the executing call keeps the tables alive; every result is right whichever path served it."""
    import threading
    lib = ffi.load()
    k = 14
    n = 1 << k
    bases = po.gen_bases(0, 9980, n)
    scalars = po.gen_scalars(po.F_BN254_FR, 9981, n)
    want = po.expected_from_linearity(0, 9980, scalars)
    db, ds = DeviceBuffer.from_host(bases), DeviceBuffer.from_host(scalars)
    errors, done = [], threading.Event()

    def executor():
        try:
            g = pgm.PandaGpuManager(0)
            dr = DeviceBuffer(96)
            cfg = ffi.MSMConfiguration(g.mem_pool, g.exec_stream.raw, db.ptr, ds.ptr, dr.ptr, k, pgm.JACOBIAN)
            for _ in range(40):
                ffi.check(lib.panda_msm_execute_bn254(cfg), "msm")
                if not (po.to_affine(0, dr.to_host()) == want).all():
                    errors.append("wrong result")
            dr.free()
            g.deinit()
        except Exception as e:  # pragma: no cover
            errors.append(repr(e))
        finally:
            done.set()

    t = threading.Thread(target=executor)
    t.start()
    flips = 0
    while not done.is_set() and flips < 200:
        if lib.panda_msm_precompute_bases(0, db.ptr, k, 0, gm.exec_stream.raw) == 0:
            lib.panda_msm_unregister_bases(db.ptr)
            flips += 1
    t.join()
    lib.panda_msm_unregister_bases(db.ptr)
    db.free()
    ds.free()
    assert not errors, errors
    assert flips > 0


def _edge_scalars(kind, n):
    c = pyref.CURVES[0]
    rng = np.random.default_rng(9)
    mont = lambda v: pyref.int_to_limbs(v * c.Rr % c.r, 8)
    if kind == "zeros":
        return np.zeros((n, 8), np.uint32)
    if kind == "ones":
        return np.tile(mont(1), (n, 1))
    if kind == "minus_one":
        return np.tile(mont(c.r - 1), (n, 1))
    if kind == "small":
        return np.stack([mont(int(v)) for v in rng.integers(0, 1 << 16, n)])
    if kind == "all_equal":
        return np.tile(po.gen_scalars(po.F_BN254_FR, 77, 1), (n, 1))
    if kind == "half_zero":
        s = po.gen_scalars(po.F_BN254_FR, 78, n)
        s[::2] = 0
        return s
    return np.stack([mont((1 << 253) + int(v)) for v in rng.integers(0, 1 << 30, n)])


@pytest.mark.parametrize("cid,k,wbits", [(0, 10, 10), (0, 12, 12), (0, 13, 16), (0, 13, 0), (0, 14, 18), (0, 16, 0), (1, 12, 13), (1, 13, 17)])
def test_msm_precomputed_tables(gm, cid, k, wbits):
    """panda_msm_precompute_bases: window tables 2^lo[k]*P, one shared bucket space, three-level sort.
    Same group element as the plain path for several table geometries; other buffers are unaffected."""
    n = 1 << k
    bases = po.gen_bases(cid, 9400 + k, n)
    other = po.gen_bases(cid, 9500 + k, n)
    idx = gm.add_cached_bases(bases)
    tables, bits, held = gm.precompute_cached_bases(idx, curve=cid, window_bits=wbits)
    assert tables >= 2 and (wbits == 0 or 0 < bits <= wbits) and held == tables * n * 2 * po.LC_Q[cid] * 4
    for rep in range(2):
        scalars = po.gen_scalars(po.FR_OF[cid], 9600 + rep, n)
        keep = scalars.copy()
        coord = pgm.PROJECTIVE if rep else pgm.JACOBIAN
        gm.set_config(coord)
        try:
            out = pgm.panda_msm_bn254_gpu_with_cached_bases(gm, scalars, idx, curve=cid)
        finally:
            gm.set_config(pgm.JACOBIAN)
        assert (affine_of(cid, out, coord) == po.expected_from_linearity(cid, 9400 + k, scalars)).all()
        assert (scalars == keep).all()
    out2 = pgm.panda_msm_bn254_gpu(gm, scalars, other, curve=cid)
    assert (affine_of(cid, out2) == po.expected_from_linearity(cid, 9500 + k, scalars)).all()


@pytest.mark.parametrize("kind", EDGE_SETS)
def test_msm_precomputed_tables_edge_scalars(gm, kind):
    n = 1 << 12
    bases = po.gen_bases(0, 55, n)
    idx = gm.add_cached_bases(bases)
    gm.precompute_cached_bases(idx, curve=0, window_bits=14)
    scalars = _edge_scalars(kind, n)
    out = pgm.panda_msm_bn254_gpu_with_cached_bases(gm, scalars, idx)
    assert (affine_of(0, out) == po.expected_from_linearity(0, 55, scalars)).all()
    if kind == "zeros":
        assert not out.view(np.uint32)[16:].any()  # Z == 0: the identity


def test_msm_precomputed_tables_degenerate_bases(gm):
    """identity rows stay identity in every table; repeated points and P, -P pairs meet in the same bucket."""
    n = 1 << 10
    lc = 8
    bases = po.gen_bases(0, 66, n)
    scalars = po.gen_scalars(po.F_BN254_FR, 67, n)
    bases[3::7, :lc] = 0
    bases[100:200] = bases[100]
    bases[301] = bases[300]
    bases[301, lc:] = po.f_vec(0, po.OP_SUB, np.zeros((1, lc), np.uint32), bases[300:301, lc:])[0]
    scalars[301] = scalars[300]
    scalars[100:150] = scalars[100]
    idx = gm.add_cached_bases(bases)
    gm.precompute_cached_bases(idx, curve=0, window_bits=11)
    out = pgm.panda_msm_bn254_gpu_with_cached_bases(gm, scalars, idx)
    assert (affine_of(0, out) == po.msm_affine(0, bases, scalars, window_bits=9)).all()


def _msm_precomputed_on_device(gm, cid, k, wbits, seed_b, seed_s):
    n = 1 << k
    lib = ffi.load()
    db = DeviceBuffer(n * 2 * po.LC_Q[cid] * 4)
    ds = DeviceBuffer(n * 32)
    dr = DeviceBuffer(3 * po.LC_Q[cid] * 4)
    ffi.check(lib.panda_gen_bases(cid, seed_b, 0, n, db.ptr, NULL_STREAM), "gen")
    ffi.check(lib.panda_gen_scalars(cid, seed_s, 0, n, ds.ptr, NULL_STREAM), "gen")
    ffi.check(lib.panda_msm_precompute_bases(cid, db.ptr, k, wbits, gm.exec_stream.raw), "precompute")
    tables, bits = C.c_uint(0), C.c_uint(0)
    ffi.check(lib.panda_msm_registered_info(db.ptr, C.byref(tables), C.byref(bits), None), "info")
    cfg = ffi.MSMConfiguration(gm.mem_pool, gm.exec_stream.raw, db.ptr, ds.ptr, dr.ptr, k, pgm.JACOBIAN)
    fn = (lib.panda_msm_execute_bn254, lib.panda_msm_execute_bls12_377, lib.panda_msm_execute_bls12_381)[cid]
    ffi.check(fn(cfg), "msm")
    out = dr.to_host()
    scalars = ds.to_host().reshape(n, 8)
    ffi.check(lib.panda_msm_unregister_bases(db.ptr), "unregister")
    for d in (db, ds, dr):
        d.free()
    return out, scalars, tables.value, bits.value


def test_msm_precomputed_tables_wide_words_2_18(gm):
    """2^18 points with 22-bit windows: 2^21 shared buckets (7 + 7 + 7 key bits over the three sort levels)."""
    out, scalars, tables, bits = _msm_precomputed_on_device(gm, 0, 18, 22, 0x70616E6461 ^ 7, 0x5CA1A7)
    assert bits == 22 and tables == 12
    assert (po.to_affine(0, out) == po.expected_from_linearity(0, 0x70616E6461 ^ 7, scalars)).all()


@pytest.mark.parametrize("cid,k", [(0, 20), (0, 24), (1, 22), (0, 26)])
def test_msm_precomputed_tables_baseline_sizes(gm, cid, k):
    """BASELINE configs 2 (BN254 2^20, cached bases) and the headline 2^24 with the built-in table policy; linearity."""
    seed_b = 0x70616E6461 ^ (8 + k)
    out, scalars, tables, bits = _msm_precomputed_on_device(gm, cid, k, 0, seed_b, 0x5CA1A8 + k)
    assert tables >= 2
    assert (po.to_affine(cid, out) == po.expected_from_linearity(cid, seed_b, scalars)).all()


@pytest.mark.parametrize("kind", ["all_equal", "ones", "half_zero"])
def test_msm_precomputed_tables_skewed_2_15(gm, kind):
    """2^15 equal scalars put a whole window into one bucket: level-2 segments of many tiles, level-3 cells beyond the
    register-resident size, buckets cut into hundreds of chunk pieces."""
    n = 1 << 15
    bases = po.gen_bases(0, 56, n)
    idx = gm.add_cached_bases(bases)
    gm.precompute_cached_bases(idx, curve=0, window_bits=16)
    scalars = _edge_scalars(kind, n)
    out = pgm.panda_msm_bn254_gpu_with_cached_bases(gm, scalars, idx)
    assert (affine_of(0, out) == po.expected_from_linearity(0, 56, scalars)).all()


@pytest.mark.parametrize("cid,k,wbits,front,wgs,kind", [
    (0, 13, 19, 64, 6, "uniform"), (0, 14, 22, 40, 6, "all_equal"), (1, 14, 20, 32, 0, "uniform"),
    _soak(0, 15, 20, 32, 4, "uniform"), _soak(0, 16, 22, 8, 6, "uniform"), _soak(0, 16, 22, 120, 2, "uniform"), _soak(0, 14, 21, 1, 64, "uniform"),
    _soak(0, 14, 22, 40, 6, "zeros"), _soak(0, 14, 22, 40, 6, "ones"), _soak(0, 14, 22, 40, 6, "minus_one"), _soak(0, 14, 22, 40, 6, "small"),
    _soak(0, 14, 22, 40, 6, "half_zero"), _soak(0, 14, 22, 40, 6, "top_bit"), _soak(0, 18, 22, 32, 6, "uniform"), _soak(0, 20, 0, 32, 6, "uniform"),
    _soak(2, 13, 19, 64, 0, "uniform"), _soak(1, 2, 22, 121, 8, "uniform"), _soak(1, 2, 20, 113, 6, "uniform"), _soak(0, 3, 22, 100, 64, "uniform"), _soak(0, 5, 21, 127, 8, "uniform"),
    _soak(0, 6, 22, 64, 64, "small_tiny")])
def test_msm_sort_of_the_rest_beside_the_accumulation_of_the_front(gm, cid, k, wbits, front, wgs, kind):
    """panda_msm_set_overlap: with tables, levels 2 and 3 of the sort for the back of the bucket space run on a second stream beside the
    accumulation of the front (a launch of a few workgroups per CU), then the rest is accumulated.  Same group element as the oracle
    says and as the one-stream schedule returns (not byte for byte: the order of a bucket's entries, and with it the Jacobian
    representative, follows the merge kernel's LDS atomics from run to run): for every way the entries can fall on the two sides of
    the cut -- uniform, everything in a handful of buckets, nothing at all."""
    lib = ffi.load()
    n = 1 << k
    lq = po.LC_Q[cid]
    db, ds, dr = DeviceBuffer(n * 2 * lq * 4), DeviceBuffer(n * 32), DeviceBuffer(3 * lq * 4)
    seed_b = 0x0E11A9 + 31 * k + cid
    ffi.check(lib.panda_gen_bases(cid, seed_b, 0, n, db.ptr, NULL_STREAM), "gen")
    if kind == "uniform":
        ffi.check(lib.panda_gen_scalars(cid, 0x5CA1AB + k, 0, n, ds.ptr, NULL_STREAM), "gen")
    elif kind == "small_tiny":  # a handful of entries, all in the front part: the list's only chunk is short and ends where the rest begins
        c0 = pyref.CURVES[0]
        tiny = np.stack([pyref.int_to_limbs((3 + j) * c0.Rr % c0.r, 8) for j in range(n)])
        ffi.check(lib.panda_memcpy(ds.ptr, C.c_void_p(tiny.ctypes.data), n * 32), "memcpy")
    else:
        ffi.check(lib.panda_memcpy(ds.ptr, C.c_void_p(np.ascontiguousarray(_edge_scalars(kind, n)).ctypes.data), n * 32), "memcpy")
    scalars = ds.to_host().reshape(n, 8)
    ffi.check(lib.panda_msm_precompute_bases(cid, db.ptr, k, wbits, gm.exec_stream.raw), "precompute")
    cfg = ffi.MSMConfiguration(gm.mem_pool, gm.exec_stream.raw, db.ptr, ds.ptr, dr.ptr, k, pgm.JACOBIAN)
    fn = (lib.panda_msm_execute_bn254, lib.panda_msm_execute_bls12_377, lib.panda_msm_execute_bls12_381)[cid]
    try:
        ffi.check(lib.panda_msm_set_overlap(0, 0), "set_overlap")
        ffi.check(fn(cfg), "msm")
        serial = dr.to_host().copy()
        ffi.check(lib.panda_msm_set_overlap(front, wgs), "set_overlap")
        for _ in range(2):  # the second call meets the first one's offsets and cell positions in the arena
            ffi.check(lib.panda_memset(dr.ptr, 0, 3 * lq * 4), "memset")
            ffi.check(fn(cfg), "msm")
            split = dr.to_host().copy()
            assert (po.to_affine(cid, split) == po.to_affine(cid, serial)).all()
    finally:
        lib.panda_msm_set_overlap(0xFFFFFFFF, 0)
        lib.panda_msm_unregister_bases(db.ptr)
    assert (ds.to_host().reshape(n, 8) == scalars).all()
    assert (po.to_affine(cid, split) == po.expected_from_linearity(cid, seed_b, scalars)).all()
    assert lib.panda_msm_set_overlap(128, 0) != 0 and lib.panda_msm_set_overlap(1, 65) != 0
    for d in (db, ds, dr):
        d.free()


_LDS_ROW_CASES = [(12, 0, "uniform"), (15, 16, "uniform"), (18, 0, "uniform"), (13, 14, "all_equal"), (13, 14, "half_zero"), (13, 14, "small"), (13, 0, "plain")]


@pytest.mark.parametrize("variant,k,wbits,kind", [(0,) + c for c in _LDS_ROW_CASES]  # variant 0 is the built-in kernel: identity rows, skew, plain path
                         + [(1, 15, 16, "uniform"), (2, 13, 14, "all_equal"), (3, 13, 0, "plain"), (4, 15, 16, "uniform"), (5, 15, 16, "uniform"), (5, 13, 14, "all_equal"), (5, 13, 0, "plain")]
                         + [_soak(v, *c) for v in (1, 2, 3, 4, 5) for c in _LDS_ROW_CASES
                            if (v,) + c not in ((1, 15, 16, "uniform"), (2, 13, 14, "all_equal"), (3, 13, 0, "plain"), (4, 15, 16, "uniform"), (5, 15, 16, "uniform"), (5, 13, 14, "all_equal"), (5, 13, 0, "plain"))])
def test_msm_accumulate_with_the_row_staged_in_lds(gm, variant, k, wbits, kind):
    """panda_msm_set_accumulate_variant: k_accumulate with the next entry's row staged in LDS (global_load_lds) at five / four waves per
    SIMD (1, 2), rows fetched four lanes to a row (3), the sorted words in 64-byte sectors through LDS forced (4: the built-in kernel of the 9-limb fields) or forbidden (5: rounds 2-5) -- the same group element as the
    oracle says, with tables and on the plain path, uniform and skewed scalars, identity rows"""
    lib = ffi.load()
    n = 1 << k
    db, ds, dr = DeviceBuffer(n * 64), DeviceBuffer(n * 32), DeviceBuffer(96)
    seed_b = 0x1D5 + k
    bases = po.gen_bases(0, seed_b, n)
    bases[5::97, :8] = 0  # identity rows (x == 0)
    ffi.check(lib.panda_memcpy(db.ptr, C.c_void_p(bases.ctypes.data), n * 64), "memcpy")
    scalars = po.gen_scalars(po.F_BN254_FR, 0x1D6 + k, n) if kind in ("uniform", "plain") else _edge_scalars(kind, n)
    ffi.check(lib.panda_memcpy(ds.ptr, C.c_void_p(np.ascontiguousarray(scalars).ctypes.data), n * 32), "memcpy")
    if kind != "plain":
        ffi.check(lib.panda_msm_precompute_bases(0, db.ptr, k, wbits, gm.exec_stream.raw), "precompute")
    cfg = ffi.MSMConfiguration(gm.mem_pool, gm.exec_stream.raw, db.ptr, ds.ptr, dr.ptr, k, pgm.JACOBIAN)
    try:
        ffi.check(lib.panda_msm_set_accumulate_variant(variant), "variant")
        for _ in range(2):
            ffi.check(lib.panda_msm_execute_bn254(cfg), "msm")
        got = po.to_affine(0, dr.to_host())
    finally:
        lib.panda_msm_set_accumulate_variant(0)
        lib.panda_msm_unregister_bases(db.ptr)
    assert (got == po.msm_affine(0, bases, scalars, window_bits=11)).all()
    assert lib.panda_msm_set_accumulate_variant(6) != 0
    for d in (db, ds, dr):
        d.free()


_WIDE_MERGE_CASES = [(15, 14, "all_equal"), (16, 14, "all_equal"), (15, 0, "ones"), (14, 12, "small"), (16, 0, "half_zero"), (15, 16, "top_bit"), (17, 0, "uniform")]
_WIDE_MERGE_KEPT = [(1, 15, 14, "all_equal"), (1, 16, 14, "all_equal"), (2, 16, 14, "all_equal"), (3, 17, 0, "uniform")]  # mode 1 is what the policy picks from 2^24 points up


@pytest.mark.parametrize("mode,k,wbits,kind", _WIDE_MERGE_KEPT + [_soak(m, *c) for m in (1, 2, 3) for c in _WIDE_MERGE_CASES if (m,) + c not in _WIDE_MERGE_KEPT])
def test_msm_level3_merge_wide_variant(gm, mode, k, wbits, kind):
    """panda_msm_set_wide_merge: every level-3 cell through the variant of k3_merge that reads up to 32 k entries per cell once and writes
    them in two rounds (mode 1), or none (mode 2: large cells take the two-pass path).  2^15 equal scalars put exactly 32 k entries into
    one cell (two rounds), 2^16 put 64 k (the two-pass path inside the wide kernel); the policy itself only picks the variant from 2^24 points
    up (test_msm_baseline_full_sizes)."""
    lib = ffi.load()
    n = 1 << k
    db, ds, dr = DeviceBuffer(n * 64), DeviceBuffer(n * 32), DeviceBuffer(96)
    seed_b = 0x51DE + k
    ffi.check(lib.panda_gen_bases(0, seed_b, 0, n, db.ptr, NULL_STREAM), "gen")
    scalars = po.gen_scalars(po.F_BN254_FR, 0x51DF + k, n) if kind == "uniform" else _edge_scalars(kind, n)
    scalars = np.ascontiguousarray(scalars, dtype=np.uint32)
    ffi.check(lib.panda_memcpy(ds.ptr, C.c_void_p(scalars.ctypes.data), n * 32), "memcpy")
    ffi.check(lib.panda_msm_precompute_bases(0, db.ptr, k, wbits, gm.exec_stream.raw), "precompute")
    cfg = ffi.MSMConfiguration(gm.mem_pool, gm.exec_stream.raw, db.ptr, ds.ptr, dr.ptr, k, pgm.JACOBIAN)
    try:
        ffi.check(lib.panda_msm_set_wide_merge(mode), "mode")
        for _ in range(2):
            ffi.check(lib.panda_memset(dr.ptr, 0, 96), "memset")
            ffi.check(lib.panda_msm_execute_bn254(cfg), "msm")
            assert (po.to_affine(0, dr.to_host()) == po.expected_from_linearity(0, seed_b, scalars)).all()
    finally:
        lib.panda_msm_set_wide_merge(0)
        lib.panda_msm_unregister_bases(db.ptr)
    assert lib.panda_msm_set_wide_merge(4) != 0
    for d in (db, ds, dr):
        d.free()


@pytest.mark.parametrize("cid,k,tables", [(1, 14, True), (3, 11, False), _soak(2, 13, True), _soak(3, 12, True), _soak(1, 12, False), _soak(0, 16, True)])
def test_msm_accumulate_with_rows_fetched_four_lanes_to_a_row(gm, cid, k, tables):
    """panda_msm_set_accumulate_variant(3) on every curve (rows of 64, 96 and 128 bytes: 4, 6, 8 pieces a row), tables and plain path"""
    lib = ffi.load()
    n = 1 << k
    pt, res = ((64, 96), (96, 144), (96, 144), (128, 192))[cid]
    fn = (lib.panda_msm_execute_bn254, lib.panda_msm_execute_bls12_377, lib.panda_msm_execute_bls12_381, lib.panda_msm_execute_bn254_g2)[cid]
    db, ds, dr = DeviceBuffer(n * pt), DeviceBuffer(n * 32), DeviceBuffer(res)
    seed_b = 0x5A4ED + k
    ffi.check(lib.panda_gen_bases(cid, seed_b, 0, n, db.ptr, NULL_STREAM), "gen")
    ffi.check(lib.panda_gen_scalars(cid, 0x5A4EE + k, 0, n, ds.ptr, NULL_STREAM), "gen")
    scalars = ds.to_host().reshape(n, 8)
    if tables:
        ffi.check(lib.panda_msm_precompute_bases(cid, db.ptr, k, 0, gm.exec_stream.raw), "precompute")
    cfg = ffi.MSMConfiguration(gm.mem_pool, gm.exec_stream.raw, db.ptr, ds.ptr, dr.ptr, k, pgm.JACOBIAN)
    try:
        ffi.check(lib.panda_msm_set_accumulate_variant(3), "variant")
        for _ in range(2):
            ffi.check(lib.panda_memset(dr.ptr, 0, res), "memset")
            ffi.check(fn(cfg), "msm")
            if cid == 3:
                assert _g2_decode(dr.to_host()) == _g2_expected(seed_b, scalars)
            else:
                assert (po.to_affine(cid, dr.to_host()) == po.expected_from_linearity(cid, seed_b, scalars)).all()
    finally:
        lib.panda_msm_set_accumulate_variant(0)
        lib.panda_msm_unregister_bases(db.ptr)
    for d in (db, ds, dr):
        d.free()


@pytest.mark.parametrize("kind", ["all_equal", "half_zero", "three_values", "all_equal_from_host_5_ranges"])
def test_msm_skewed_scalars_full_size_2_22(gm, kind):
    """Skew at a size where it bites: 2^22 points with precomputed tables.  Equal scalars put 2^22 entries into ONE bucket per window
    (buckets cut into tens of thousands of chunk pieces: the k_fixup_long queue; level-2 / level-3 sort segments far beyond a tile),
    half-zero scalars halve every list, three distinct values load three buckets per window.  Checked by linearity over all scalars;
    the last case goes through panda_msm_execute_from_host with five upload ranges (range buckets merged by the fix-ups)."""
    lib = ffi.load()
    k = 22
    n = 1 << k
    seed_b = 0x70616E6461 ^ 0x5EED22
    db, ds, dr = DeviceBuffer(n * 64), DeviceBuffer(n * 32), DeviceBuffer(96)
    ffi.check(lib.panda_gen_bases(0, seed_b, 0, n, db.ptr, NULL_STREAM), "gen")
    if kind.startswith("all_equal"):
        scalars = np.tile(po.gen_scalars(po.F_BN254_FR, 771, 1), (n, 1))
    elif kind == "half_zero":
        ffi.check(lib.panda_gen_scalars(0, 772, 0, n, ds.ptr, NULL_STREAM), "gen")
        scalars = ds.to_host().reshape(n, 8)
        scalars[::2] = 0
    else:
        three = po.gen_scalars(po.F_BN254_FR, 773, 3)
        scalars = three[np.random.default_rng(774).integers(0, 3, n)]
    scalars = np.ascontiguousarray(scalars, dtype=np.uint32)
    ffi.check(lib.panda_memcpy(ds.ptr, C.c_void_p(scalars.ctypes.data), n * 32), "copy")
    ffi.check(lib.panda_msm_precompute_bases(0, db.ptr, k, 0, gm.exec_stream.raw), "precompute")
    cfg = ffi.MSMConfiguration(gm.mem_pool, gm.exec_stream.raw, db.ptr, ds.ptr, dr.ptr, k, pgm.JACOBIAN)
    try:
        if kind.endswith("from_host_5_ranges"):
            ffi.check(lib.panda_memset(ds.ptr, 0, n * 32), "memset")  # the call fills the device buffer itself
            ffi.check(lib.panda_msm_execute_from_host(0, cfg, C.c_void_p(scalars.ctypes.data), 5, gm.h2d_stream.raw), "msm")
        else:
            ffi.check(lib.panda_msm_execute_bn254(cfg), "msm")
        got = po.to_affine(0, dr.to_host())
        assert (ds.to_host().reshape(n, 8) == scalars).all()  # scalars untouched (or uploaded intact)
    finally:
        lib.panda_msm_unregister_bases(db.ptr)
        for d in (db, ds, dr):
            d.free()
    assert (got == po.expected_from_linearity(0, seed_b, scalars)).all()


@pytest.mark.parametrize("tabled", [False, True])
def test_msm_batched_pipeline_over_cached_bases(gm, tabled):
    """upload of batch k+1 on the h2d stream under the execution of batch k; results in order and equal
    to the one-at-a-time calls."""
    n = 1 << 13
    bases = po.gen_bases(0, 9700, n)
    idx = gm.add_cached_bases(bases)
    if tabled:
        gm.precompute_cached_bases(idx, curve=0)
    batches = [po.gen_scalars(po.F_BN254_FR, 9800 + j, n) for j in range(5)]
    outs = pgm.panda_msm_bn254_gpu_with_cached_bases_batched(gm, batches, idx)
    assert len(outs) == 5
    for j, out in enumerate(outs):
        assert (affine_of(0, out) == po.expected_from_linearity(0, 9700, batches[j])).all()
        single = pgm.panda_msm_bn254_gpu_with_cached_bases(gm, batches[j], idx)
        assert (affine_of(0, single) == affine_of(0, out)).all()  # the Jacobian representative depends on the addition order
    assert pgm.panda_msm_bn254_gpu_with_cached_bases_batched(gm, [], idx) == []
    with pytest.raises(ffi.PandaGpuError):
        pgm.panda_msm_bn254_gpu_with_cached_bases_batched(gm, batches[:1], 999)


@pytest.mark.parametrize("cid,k,tabled,chunks,source", [(0, 17, True, 2, "pageable"), (0, 19, True, 8, "pinned"), (0, 19, False, 4, "pinned"), (0, 20, True, 4, "resident"),
                                                         (1, 18, True, 4, "pageable"), (0, 18, True, 64, "pinned"), (0, 15, True, 4, "pinned"), (0, 18, None, 4, "pinned"),
                                                         (0, 18, True, 3, "pinned_same_stream"), (0, 18, False, 3, "pinned_null_stream")])
def test_msm_upload_pipeline_inside_one_call(gm, cid, k, tabled, chunks, source):
    """panda_msm_execute_from_host cuts one MSM into point ranges, uploads range r+1 while range r is accumulated against
    its own rows of the registered tables, merges the ranges' buckets on the device.  Same group element as the ordinary call, for
    tables / converted-only / unregistered bases (the last falls back to copy-then-execute), pageable, pinned and resident scalars,
    range counts the library has to clamp (64 ranges of 2^18 points, 4 ranges of 2^15), and skewed scalars."""
    lib = ffi.load()
    n = 1 << k
    lc = po.LC_Q[cid]
    seed_b, seed_s = 0x70616E6461 ^ (0xF200 + 16 * k + cid), 0xF2F2 + k
    db, ds, dr = DeviceBuffer(n * 2 * lc * 4), DeviceBuffer(n * 32), DeviceBuffer(3 * lc * 4)
    ffi.check(lib.panda_gen_bases(cid, seed_b, 0, n, db.ptr, NULL_STREAM), "gen")
    ffi.check(lib.panda_gen_scalars(cid, seed_s, 0, n, ds.ptr, NULL_STREAM), "gen")
    scalars = ds.to_host().reshape(n, 8)
    scalars[n // 3:n // 3 + 3000] = scalars[5]  # a run of equal scalars: one bucket per window takes thousands of entries in one range
    scalars[::7] = 0
    want = po.expected_from_linearity(cid, seed_b, scalars)
    if tabled is True:
        ffi.check(lib.panda_msm_precompute_bases(cid, db.ptr, k, 0, gm.exec_stream.raw), "precompute")
    elif tabled is False:
        ffi.check(lib.panda_msm_register_bases(cid, db.ptr, k, gm.exec_stream.raw), "register")
    pinned = C.c_void_p()
    h2d = gm.h2d_stream.raw
    if source == "pinned_same_stream":  # copies and kernels on ONE stream: no overlap, same answer
        h2d, source = gm.exec_stream.raw, "pinned"
    elif source == "pinned_null_stream":  # the legacy NULL stream as copy stream
        h2d, source = NULL_STREAM, "pinned"
    if source == "pinned":
        ffi.check(lib.panda_malloc_host(C.byref(pinned), n * 32), "malloc_host")
        C.memmove(pinned, scalars.ctypes.data, n * 32)
        h_ptr = pinned
        ffi.check(lib.panda_memset(ds.ptr, 0xEE, n * 32), "memset")  # the device buffer holds nothing useful before the call
    elif source == "pageable":
        h_ptr = C.c_void_p(scalars.ctypes.data)
        ffi.check(lib.panda_memset(ds.ptr, 0xEE, n * 32), "memset")
    else:
        h_ptr = None
        ffi.check(lib.panda_memcpy(ds.ptr, C.c_void_p(scalars.ctypes.data), n * 32), "memcpy")
    for coord in (pgm.JACOBIAN, pgm.PROJECTIVE):
        cfg = ffi.MSMConfiguration(gm.mem_pool, gm.exec_stream.raw, db.ptr, ds.ptr, dr.ptr, k, coord)
        ffi.check(lib.panda_msm_execute_from_host(cid, cfg, h_ptr, chunks, h2d), "msm")
        got = dr.to_host()
        assert ((po.hom_to_affine(cid, got) if coord == pgm.PROJECTIVE else po.to_affine(cid, got)) == want).all()
    assert (ds.to_host().reshape(n, 8) == scalars).all()  # the whole scalar set arrived on the device
    if tabled is not None:
        ffi.check(lib.panda_msm_unregister_bases(db.ptr), "unregister")
    if pinned:
        lib.panda_free_host(pinned)
    for d in (db, ds, dr):
        d.free()


@pytest.mark.parametrize("cid,k,R", [(0, 18, 4), (0, 19, 8), (1, 18, 2), (0, 17, 4)])
def test_msm_equal_point_ranges_inside_one_call(gm, cid, k, R):
    """panda_msm_execute_from_host with ranges = 0x100 | R and no host source: R equal point ranges one after the other on resident scalars
    (the footprint experiment of profiles/r05_accumulate_table_footprint.txt); too small an input runs fewer ranges; same group element."""
    lib = ffi.load()
    n = 1 << k
    lq = po.LC_Q[cid]
    db, ds, dr = DeviceBuffer(n * 2 * lq * 4), DeviceBuffer(n * 32), DeviceBuffer(3 * lq * 4)
    seed_b = 0xE90A1 + k
    ffi.check(lib.panda_gen_bases(cid, seed_b, 0, n, db.ptr, NULL_STREAM), "gen")
    ffi.check(lib.panda_gen_scalars(cid, 0xE90A2 + k, 0, n, ds.ptr, NULL_STREAM), "gen")
    scalars = ds.to_host().reshape(n, 8)
    ffi.check(lib.panda_msm_precompute_bases(cid, db.ptr, k, 0, gm.exec_stream.raw), "precompute")
    cfg = ffi.MSMConfiguration(gm.mem_pool, gm.exec_stream.raw, db.ptr, ds.ptr, dr.ptr, k, pgm.JACOBIAN)
    try:
        for _ in range(2):
            ffi.check(lib.panda_memset(dr.ptr, 0, 3 * lq * 4), "memset")
            ffi.check(lib.panda_msm_execute_from_host(cid, cfg, None, 0x100 | R, gm.exec_stream.raw), "msm in equal ranges")
            assert (po.to_affine(cid, dr.to_host()) == po.expected_from_linearity(cid, seed_b, scalars)).all()
    finally:
        lib.panda_msm_unregister_bases(db.ptr)
    assert (ds.to_host().reshape(n, 8) == scalars).all()
    for d in (db, ds, dr):
        d.free()


def test_msm_upload_pipeline_through_the_manager(gm):
    """panda_msm_bn254_gpu_with_cached_bases over a registered base set takes the pipelined path (2^19 points: two ranges)."""
    k = 19
    n = 1 << k
    lib = ffi.load()
    db = DeviceBuffer(n * 64)
    ffi.check(lib.panda_gen_bases(0, 0xF3, 0, n, db.ptr, NULL_STREAM), "gen")  # the oracle's generator takes too long for 2^19 points
    bases = db.to_host().reshape(n, 16)
    db.free()
    scalars = po.gen_scalars(po.F_BN254_FR, 0xF4, n)
    idx = gm.add_cached_bases(bases)
    gm.precompute_cached_bases(idx)
    assert pgm.pipeline_chunks(k) == 2
    keep = scalars.copy()
    out = pgm.panda_msm_bn254_gpu_with_cached_bases(gm, scalars, idx)
    assert (affine_of(0, out) == po.expected_from_linearity(0, 0xF3, scalars)).all()
    assert (scalars == keep).all()


def test_msm_ragged_lengths_and_bad_arguments(gm):
    """unit.rs:31: n = 2^floor(log2(len / 32)) -- a ragged scalar slice uses its leading power-of-two prefix (and as many bases).
    Null pointers and oversize counts come back as panda_error_invalid_value, never as a crash."""
    bases = po.gen_bases(0, 9900, 1024)
    scalars = po.gen_scalars(po.F_BN254_FR, 9901, 1000)  # 1000 -> n = 512
    out = pgm.panda_msm_bn254_gpu(gm, scalars, bases)
    assert (affine_of(0, out) == po.msm_affine(0, bases[:512], scalars[:512], window_bits=9)).all()
    out1 = pgm.panda_msm_bn254_gpu(gm, scalars[:1], bases[:1])  # a single point: log_n = 0
    assert (affine_of(0, out1) == po.msm_affine(0, bases[:1], scalars[:1], window_bits=4)).all()
    lib = ffi.load()
    d = DeviceBuffer(4096)
    for cfg in (ffi.MSMConfiguration(gm.mem_pool, gm.exec_stream.raw, None, d.ptr, d.ptr, 2, pgm.JACOBIAN),
                ffi.MSMConfiguration(gm.mem_pool, gm.exec_stream.raw, d.ptr, None, d.ptr, 2, pgm.JACOBIAN),
                ffi.MSMConfiguration(gm.mem_pool, gm.exec_stream.raw, d.ptr, d.ptr, None, 2, pgm.JACOBIAN),
                ffi.MSMConfiguration(gm.mem_pool, gm.exec_stream.raw, d.ptr, d.ptr, d.ptr, 27, pgm.JACOBIAN)):
        assert lib.panda_msm_execute_bn254(cfg) == 1
        assert lib.panda_msm_execute_bls12_377(cfg) == 1
    assert lib.panda_msm_precompute_bases(0, None, 10, 0, gm.exec_stream.raw) == 1
    assert lib.panda_msm_precompute_bases(4, d.ptr, 10, 0, gm.exec_stream.raw) == 1  # curve ids are 0 ... 3
    assert lib.panda_msm_precompute_bases(0, d.ptr, 3, 30, gm.exec_stream.raw) == 1
    assert lib.panda_msm_register_bases(0, d.ptr, 27, gm.exec_stream.raw) == 1
    flag = C.c_uint(0)
    bad = ffi.NttconfigurationV1(gm.mem_pool, gm.exec_stream.raw, None, d.ptr, None, 4, C.pointer(flag))
    assert lib.panda_ntt_execute_bn254_v1(bad) == 1
    # Buffers shorter than log_n implies are refused at the boundary instead of being read past their end.  The first line is the
    # call that took the process down in round 2: 2^10 G2 points are 128 KiB, the buffer has 4 KiB.
    assert lib.panda_msm_precompute_bases(3, d.ptr, 10, 0, gm.exec_stream.raw) == 1
    assert lib.panda_msm_register_bases(0, d.ptr, 10, gm.exec_stream.raw) == 1
    big = DeviceBuffer(1 << 20)
    om = po.root_of_unity(po.F_BN254_FR, 10)
    for b, s_, r in ((d, big, big), (big, d, big)):  # bases / scalars too short for 2^10 points
        cfg = ffi.MSMConfiguration(gm.mem_pool, gm.exec_stream.raw, b.ptr, s_.ptr, r.ptr, 10, pgm.JACOBIAN)
        assert lib.panda_msm_execute_bn254(cfg) == 1
        assert lib.panda_msm_execute_bls12_377(cfg) == 1
    short_result = C.c_void_p(big.ptr.value + (1 << 20) - 64)  # 64 bytes left in the allocation, a result needs 96
    cfg = ffi.MSMConfiguration(gm.mem_pool, gm.exec_stream.raw, big.ptr, big.ptr, short_result, 10, pgm.JACOBIAN)
    assert lib.panda_msm_execute_bn254(cfg) == 1
    for src, dst in ((d, big), (big, d)):  # 2^10 elements are 32 KiB
        cfg = ffi.NttconfigurationV1(gm.mem_pool, gm.exec_stream.raw, src.ptr, dst.ptr, C.c_void_p(om.ctypes.data), 10, C.pointer(flag))
        assert lib.panda_ntt_execute_bn254_v1(cfg) == 1
        assert lib.panda_ntt_execute_bn254_inverse(cfg) == 1
    slab = ffi.NttSlabConfiguration(gm.exec_stream.raw, d.ptr, big.ptr, C.c_void_p(om.ctypes.data), 12, 1, 0, C.pointer(flag))  # slab of 2^11
    assert lib.panda_ntt_slab_step1_bn254(slab) == 1
    # ... and buffers of exactly the right size are accepted (the check is not off by one)
    exact_b, exact_s, exact_r = DeviceBuffer(64 << 10), DeviceBuffer(32 << 10), DeviceBuffer(96)
    ffi.check(lib.panda_gen_bases(0, 77, 0, 1 << 10, exact_b.ptr, NULL_STREAM), "gen")
    ffi.check(lib.panda_gen_scalars(0, 78, 0, 1 << 10, exact_s.ptr, NULL_STREAM), "gen")
    cfg = ffi.MSMConfiguration(gm.mem_pool, gm.exec_stream.raw, exact_b.ptr, exact_s.ptr, exact_r.ptr, 10, pgm.JACOBIAN)
    assert lib.panda_msm_execute_bn254(cfg) == 0
    for b in (big, exact_b, exact_s, exact_r):
        b.free()
    d.free()


# ------------------------------------------------------------------ BLS12-381 (the curve the reference names in curve.cuh:12)

@pytest.mark.parametrize("k", [10, 13])
def test_msm_bls12_381(gm, k):
    """G1 MSM, device and host entry points, Jacobian and homogeneous output."""
    cid = pgm.BLS12_381
    n = 1 << k
    bases = po.gen_bases(cid, 1500 + k, n)
    scalars = po.gen_scalars(po.F_BLS381_FR, 1600 + k, n)
    keep = scalars.copy()
    want = po.expected_from_linearity(cid, 1500 + k, scalars)
    for coord in (pgm.JACOBIAN, pgm.PROJECTIVE):
        gm.set_config(coord)
        try:
            out = pgm.panda_msm_bn254_gpu(gm, scalars, bases, curve=cid)
        finally:
            gm.set_config(pgm.JACOBIAN)
        assert out.size == 144
        assert (affine_of(cid, out, coord) == want).all()
    assert (scalars == keep).all()
    if k == 10:
        assert (want == po.msm_affine(cid, bases, scalars, window_bits=8)).all()
        host = pgm.panda_msm_bn254_gpu_host(gm, scalars, bases, curve=cid)
        assert (affine_of(cid, host) == want).all()


@pytest.mark.parametrize("wbits", [0, 14])
def test_msm_bls12_381_precomputed_tables_and_edges(gm, wbits):
    cid = pgm.BLS12_381
    n = 1 << 12
    c = pyref.CURVES[cid]
    bases = po.gen_bases(cid, 1700, n)
    bases[3::7, :12] = 0  # identity bases
    bases[301] = bases[300]
    bases[301, 12:] = po.f_vec(po.F_BLS381_FQ, po.OP_SUB, np.zeros((1, 12), np.uint32), bases[300:301, 12:])[0]  # P, -P
    scalars = po.gen_scalars(po.F_BLS381_FR, 1701, n)
    scalars[301] = scalars[300]
    mont = lambda v: pyref.int_to_limbs(v * c.Rr % c.r, 8)
    for i, v in enumerate([0, 1, c.r - 1, 0xBEEF, (1 << 254) + 5]):
        scalars[10 + i] = mont(v % c.r)
    want = po.msm_affine(cid, bases, scalars, window_bits=9)
    idx = gm.add_cached_bases(bases)
    tables, bits, held = gm.precompute_cached_bases(idx, curve=cid, window_bits=wbits)
    assert tables >= 2 and held == tables * n * 96
    out = pgm.panda_msm_bn254_gpu_with_cached_bases(gm, scalars, idx, curve=cid)
    assert (affine_of(cid, out) == want).all()
    out = pgm.panda_msm_bn254_gpu(gm, scalars, bases, curve=cid)
    assert (affine_of(cid, out) == want).all()


def test_msm_bls12_381_2_18_linearity(gm):
    out, scalars = _msm_on_device_inputs(gm, 2, 18, 0x70616E6461 ^ 6, 0x5CA1A9)
    assert (po.to_affine(2, out) == po.expected_from_linearity(2, 0x70616E6461 ^ 6, scalars)).all()
    out, scalars, tables, bits = _msm_precomputed_on_device(gm, 2, 18, 0, 0x70616E6461 ^ 6, 0x5CA1AA)
    assert tables >= 2
    assert (po.to_affine(2, out) == po.expected_from_linearity(2, 0x70616E6461 ^ 6, scalars)).all()


@pytest.mark.parametrize("log_n", [0, 3, 8, 11, 17, 20])
def test_ntt_bls12_381_fr(gm, log_n):
    """The NTT kernels over the BLS12-381 scalar field (two-adicity 32, headroom R/p = 70: the earliest reductions)."""
    fid = po.F_BLS381_FR
    om = po.root_of_unity(fid, log_n)
    x = po.gen_scalars(fid, 4100 + log_n, 1 << log_n)
    buf = x.copy()
    pgm.panda_ntt_bls12_381_gpu_v1(gm, buf, om, log_n)
    assert (buf == po.ntt(fid, x, om, log_n)).all()
    pgm.panda_ntt_bls12_381_gpu_v1(gm, buf, om, log_n, inverse=True)
    assert (buf == x).all()


def test_msm_combine_bls12_381(gm):
    """multi-GPU half: the sum of two range partials equals the whole MSM."""
    cid = pgm.BLS12_381
    n = 1 << 11
    bases = po.gen_bases(cid, 1800, n)
    scalars = po.gen_scalars(po.F_BLS381_FR, 1801, n)
    parts = np.stack([pgm.panda_msm_bn254_gpu(gm, scalars[h * (n // 2):(h + 1) * (n // 2)], bases[h * (n // 2):(h + 1) * (n // 2)], curve=cid).view(np.uint32) for h in range(2)])
    total = multi_gpu.combine_partials(parts, curve=cid)
    assert (po.to_affine(cid, total.view(np.uint32)) == po.msm_affine(cid, bases, scalars, window_bits=9)).all()


@pytest.mark.parametrize("k", [0, 1, 3, 6])
def test_msm_precompute_tiny_base_sets(gm, k):
    """A handful of points: precompute either builds tables or quietly keeps the converted copy; results are right either way."""
    n = 1 << k
    bases = po.gen_bases(0, 9950 + k, n)
    scalars = po.gen_scalars(po.F_BN254_FR, 9960 + k, n)
    idx = gm.add_cached_bases(bases)
    tables, bits, held = gm.precompute_cached_bases(idx, curve=0)
    assert tables >= 1 and held == tables * n * 64
    out = pgm.panda_msm_bn254_gpu_with_cached_bases(gm, scalars, idx)
    assert (affine_of(0, out) == po.to_affine(0, po.msm_naive(0, bases, scalars))).all()


@pytest.mark.parametrize("seed", range(24))
def test_msm_precomputed_tables_random_geometry(gm, seed):
    """Random (size, window width, scalar pattern) combinations: every split of the bucket id over the three sort levels, ragged
    level-2 segments, empty windows, cells of every size."""
    rng = np.random.default_rng(1000 + seed)
    k = int(rng.integers(4, 15))
    n = 1 << k
    wbits = int(rng.choice([0, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22]))
    c = pyref.CURVES[0]
    mont = lambda v: pyref.int_to_limbs(v * c.Rr % c.r, 8)
    pattern = int(rng.integers(0, 5))
    scalars = po.gen_scalars(po.F_BN254_FR, 5000 + seed, n)
    if pattern == 1:  # most scalars zero
        scalars[rng.random(n) < 0.9] = 0
    elif pattern == 2:  # small scalars: the high windows are empty
        scalars = np.stack([mont(int(v)) for v in rng.integers(0, 1 << 40, n)])
    elif pattern == 3:  # a few distinct values: huge buckets
        vals = [mont(int.from_bytes(rng.bytes(32), "little") % c.r) for _ in range(3)]
        scalars = np.stack([vals[int(i)] for i in rng.integers(0, 3, n)])
    elif pattern == 4:  # values just below r and just above 0: carries through every window
        scalars = np.stack([mont((c.r - 1 - int(v)) % c.r if i % 2 else int(v)) for i, v in enumerate(rng.integers(0, 1 << 20, n))])
    bases = po.gen_bases(0, 6000 + seed, n)
    idx = gm.add_cached_bases(bases)
    try:
        tables, bits, held = gm.precompute_cached_bases(idx, curve=0, window_bits=wbits)
    except ffi.PandaGpuError:
        assert wbits and k + int(np.ceil(np.log2(np.ceil(256 / wbits)))) > 31  # only an impossible geometry may be refused
        return
    out = pgm.panda_msm_bn254_gpu_with_cached_bases(gm, scalars, idx)
    assert (affine_of(0, out) == po.expected_from_linearity(0, 6000 + seed, scalars)).all(), (k, wbits, pattern, tables, bits)


@pytest.mark.parametrize("log_n", [0, 1, 5, 10, 16, 18])
def test_coset_ntt_bn254(gm, log_n):
    """Additive coset transforms: y = NTT(x[j] * g^j), checked against the oracle NTT of the pre-scaled input; the inverse
    returns the original coefficients."""
    fid = po.F_BN254_FR
    c = pyref.CURVES[0]
    n = 1 << log_n
    om = po.root_of_unity(fid, log_n)
    x = po.gen_scalars(fid, 4200 + log_n, n)
    g = pyref.int_to_limbs(5 * c.Rr % c.r, 8)  # the shift 5 in Montgomery form
    pw = np.empty((n, 8), np.uint32)
    cur = pyref.int_to_limbs(c.Rr % c.r, 8)
    gi = 5
    acc = 1
    for j in range(min(n, 1 << 12)):  # g^j in Montgomery form, by big-int arithmetic
        pw[j] = pyref.int_to_limbs(acc * c.Rr % c.r, 8)
        acc = acc * gi % c.r
    if n > (1 << 12):  # the rest by oracle multiplications in blocks
        step = pyref.int_to_limbs(pow(5, 1 << 12, c.r) * c.Rr % c.r, 8)
        for b in range(1, n >> 12):
            pw[b << 12:(b + 1) << 12] = po.f_vec(fid, po.OP_MUL, pw[(b - 1) << 12:b << 12], np.tile(step, (1 << 12, 1)))
    scaled = po.f_vec(fid, po.OP_MUL, x, pw)
    want = po.ntt(fid, scaled, om, log_n)
    buf = x.copy()
    pgm.panda_coset_ntt_bn254_gpu(gm, buf, om, g, log_n)
    assert (buf == want).all()
    pgm.panda_coset_ntt_bn254_gpu(gm, buf, om, g, log_n, inverse=True)
    assert (buf == x).all()
    del cur


# ------------------------------------------------------------------ BN254 G2 (curve id 3, coordinates in Fq2)

def _g2_expected(seed_b, scalars, first=0):
    """(sum s_i m_i) * G2 for bases m_i * G2 generated by panda_gen_bases(3, seed_b): the linearity identity, evaluated with the
    oracle's scalar-field arithmetic and one Python scalar multiplication on the twist."""
    k = pyref.limbs_to_int(po.linear_combination(0, seed_b, scalars, first))
    return pyref.g2_mul(k, pyref.G2_GEN)


def _g2_decode(out, coord=pgm.JACOBIAN):
    w = np.asarray(out).view(np.uint32)
    return pyref.g2_decode_homogeneous(w) if coord == pgm.PROJECTIVE else pyref.g2_decode_jacobian(w)


def _g2_device_bases(seed, n, first=0):
    db = DeviceBuffer(n * 128)
    ffi.check(ffi.load().panda_gen_bases(3, seed, first, n, db.ptr, NULL_STREAM), "gen")
    bases = db.to_host().reshape(n, 32)
    db.free()
    return bases


def test_g2_device_generator_and_group_law_vs_python():
    """panda_gen_bases(3, ...) = m_i * G2 (checked against Python scalar multiplications), and the three group-law kernels over Fq2
    (madd / add / dbl incl. P + P, P + (-P), identity operands) against affine arithmetic over Python integers."""
    lib = ffi.load()
    c = pyref.CURVES[0]
    n = 48
    bases = _g2_device_bases(0xD2, n, first=5)
    pts = [pyref.g2_decode_affine(b) for b in bases]
    for i in (0, 1, 17, n - 1):
        assert pts[i] == pyref.g2_mul(po.gen_multiplier(0xD2, 5 + i), pyref.G2_GEN)
        assert pyref.g2_is_on_curve(pts[i])
    one = pyref._f2_to_wire((1, 0))
    jac = np.stack([np.concatenate([b, one]) for b in bases])  # (x, y, 1): the same points as Jacobian triples
    other = _g2_device_bases(0xD3, n)
    opts = [pyref.g2_decode_affine(b) for b in other]
    neg = bases.copy()
    for i in range(n):
        neg[i, 16:] = pyref._f2_to_wire(pyref.f2_sub((0, 0), pts[i][1], c.p))
    ident = np.zeros_like(jac)
    zero_base = other.copy()
    zero_base[::5, :16] = 0

    def run(op, A, B):
        dA, dB, dR = DeviceBuffer.from_host(A), DeviceBuffer.from_host(B), DeviceBuffer(n * 192)
        ffi.check(lib.panda_debug_curve_op(3, op, dR.ptr, dA.ptr, dB.ptr, n, NULL_STREAM), "op")
        r = dR.to_host().reshape(n, 48)
        for d in (dA, dB, dR):
            d.free()
        return [pyref.g2_decode_jacobian(x) for x in r]

    assert run(0, jac, other) == [pyref.g2_add(p, q) for p, q in zip(pts, opts)]
    assert run(0, jac, bases) == [pyref.g2_add(p, p) for p in pts]                      # P + P through the mixed addition
    assert run(0, jac, neg) == [None] * n                                               # P + (-P)
    assert run(0, ident, other) == opts
    assert run(0, jac, zero_base) == [p if i % 5 == 0 else pyref.g2_add(p, q) for i, (p, q) in enumerate(zip(pts, opts))]
    ojac = np.stack([np.concatenate([b, one]) for b in other])
    assert run(1, jac, ojac) == [pyref.g2_add(p, q) for p, q in zip(pts, opts)]
    assert run(1, jac, jac) == [pyref.g2_add(p, p) for p in pts]
    assert run(1, ident, ojac) == opts and run(1, jac, ident) == pts
    assert run(2, jac, jac) == [pyref.g2_add(p, p) for p in pts]


@pytest.mark.parametrize("k", [0, 1, 3, 6, 10, 13])
def test_msm_bn254_g2_sizes(gm, k):
    """G2 MSM through the C ABI: device path in both coordinate systems against the linearity identity, the library's CPU entry point
    (pinned to the Python reference in the CPU suite) at the smaller sizes, and the Python MSM itself at the smallest."""
    n = 1 << k
    bases = _g2_device_bases(0xE000 + k, n)
    scalars = po.gen_scalars(po.F_BN254_FR, 0xE100 + k, n)
    keep = scalars.copy()
    want = _g2_expected(0xE000 + k, scalars)
    for coord in (pgm.JACOBIAN, pgm.PROJECTIVE):
        gm.set_config(coord)
        try:
            out = pgm.panda_msm_bn254_gpu(gm, scalars, bases, curve=pgm.BN254_G2)
        finally:
            gm.set_config(pgm.JACOBIAN)
        assert out.size == 192
        assert _g2_decode(out, coord) == want
    assert (scalars == keep).all()
    if k <= 10:
        assert _g2_decode(pgm.panda_msm_bn254_gpu_host(gm, scalars, bases, curve=pgm.BN254_G2)) == want
    if k <= 3:
        assert pyref.g2_msm(bases, scalars) == want


@pytest.mark.parametrize("wbits", [0, 12])
def test_msm_bn254_g2_tables_edges_and_pipeline(gm, wbits):
    """Precomputed window tables over Fq2 rows, edge scalars, identity bases, P / -P pairs, and the point-range pipeline."""
    lib = ffi.load()
    c = pyref.CURVES[0]
    k = 12
    n = 1 << k
    bases = _g2_device_bases(0xE200, n)
    mult = [po.gen_multiplier(0xE200, i) for i in range(n)]
    bases[3::7, :16] = 0                                                   # identity bases
    p300 = pyref.g2_decode_affine(bases[300])
    bases[301, :16] = bases[300, :16]
    bases[301, 16:] = pyref._f2_to_wire(pyref.f2_sub((0, 0), p300[1], c.p))  # -P next to P
    scalars = po.gen_scalars(po.F_BN254_FR, 0xE201, n)
    scalars[301] = scalars[300]
    mont = lambda v: pyref.int_to_limbs(v * c.Rr % c.r, 8)
    for i, v in enumerate([0, 1, c.r - 1, 0xBEEF, (1 << 253) + 5]):
        scalars[10 + i] = mont(v % c.r)
    # expected: sum over the bases that are still m_i * G2 (identity rows and the P / -P pair drop out)
    total = 0
    for i in range(n):
        if i % 7 == 3 or i in (300, 301):
            continue
        total = (total + pyref.decode_scalar(c, scalars[i]) * mult[i]) % c.r
    want = pyref.g2_mul(total, pyref.G2_GEN)
    assert _g2_decode(pgm.panda_msm_bn254_gpu(gm, scalars, bases, curve=pgm.BN254_G2)) == want
    idx = gm.add_cached_bases(bases)
    tables, bits, held = gm.precompute_cached_bases(idx, curve=pgm.BN254_G2, window_bits=wbits)
    assert tables >= 2 and held == tables * n * 128
    assert _g2_decode(pgm.panda_msm_bn254_gpu_with_cached_bases(gm, scalars, idx, curve=pgm.BN254_G2)) == want
    # the same registered set through the upload pipeline (2^12 points: the library lowers the ranges to one; then 2^17 below)
    ds, dr = DeviceBuffer(n * 32), DeviceBuffer(192)
    cfg = ffi.MSMConfiguration(gm.mem_pool, gm.exec_stream.raw, C.c_void_p(gm.d_bases[idx]), ds.ptr, dr.ptr, k, pgm.JACOBIAN)
    ffi.check(lib.panda_msm_execute_from_host(3, cfg, C.c_void_p(scalars.ctypes.data), 4, gm.h2d_stream.raw), "msm")
    assert _g2_decode(dr.to_host()) == want
    ds.free()
    dr.free()


def test_msm_bn254_g2_2_17_linearity_tables_and_ranges(gm):
    """2^17 G2 points generated in HBM: plain, with tables, and in two upload ranges; linearity."""
    lib = ffi.load()
    k = 17
    n = 1 << k
    db, ds, dr = DeviceBuffer(n * 128), DeviceBuffer(n * 32), DeviceBuffer(192)
    ffi.check(lib.panda_gen_bases(3, 0xE300, 0, n, db.ptr, NULL_STREAM), "gen")
    ffi.check(lib.panda_gen_scalars(3, 0xE301, 0, n, ds.ptr, NULL_STREAM), "gen")
    scalars = ds.to_host().reshape(n, 8)
    want = _g2_expected(0xE300, scalars)
    cfg = ffi.MSMConfiguration(gm.mem_pool, gm.exec_stream.raw, db.ptr, ds.ptr, dr.ptr, k, pgm.JACOBIAN)
    ffi.check(lib.panda_msm_execute_bn254_g2(cfg), "msm")
    assert _g2_decode(dr.to_host()) == want
    ffi.check(lib.panda_msm_precompute_bases(3, db.ptr, k, 0, gm.exec_stream.raw), "precompute")
    ffi.check(lib.panda_msm_execute_bn254_g2(cfg), "msm")
    assert _g2_decode(dr.to_host()) == want
    ffi.check(lib.panda_memset(ds.ptr, 0, n * 32), "memset")
    ffi.check(lib.panda_msm_execute_from_host(3, cfg, C.c_void_p(scalars.ctypes.data), 2, gm.h2d_stream.raw), "msm")
    assert _g2_decode(dr.to_host()) == want
    ffi.check(lib.panda_msm_unregister_bases(db.ptr), "unregister")
    for d in (db, ds, dr):
        d.free()


def test_process_exit_with_live_registration_and_scratch():
    """A process that ends without unregistering its bases or tearing the library down must exit cleanly: the registry is never
    destroyed from a static destructor (the HIP runtime's own exit handler may have run by then); per-thread scratch is."""
    import subprocess
    import sys
    code = (
        "import sys, ctypes as C\n"
        f"sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r}); sys.path.insert(0, {os.path.dirname(os.path.abspath(__file__))!r})\n"
        "from panda_amd import gpu_ffi as ffi\n"
        "from gpu_util import DeviceBuffer, NULL_STREAM\n"
        "lib = ffi.load(); n = 1 << 12\n"
        "db, ds, dr = DeviceBuffer(n * 64), DeviceBuffer(n * 32), DeviceBuffer(96)\n"
        "ffi.check(lib.panda_gen_bases(0, 1, 0, n, db.ptr, NULL_STREAM), 'gen')\n"
        "ffi.check(lib.panda_gen_scalars(0, 2, 0, n, ds.ptr, NULL_STREAM), 'gen')\n"
        "ffi.check(lib.panda_msm_precompute_bases(0, db.ptr, 12, 0, NULL_STREAM), 'pre')\n"
        "cfg = ffi.MSMConfiguration(ffi.PandaMemPool(), NULL_STREAM, db.ptr, ds.ptr, dr.ptr, 12, 0)\n"
        "ffi.check(lib.panda_msm_execute_bn254(cfg), 'msm')\n"
        "print('bye')\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "bye" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("log_n", [25, 26, _soak(27)])  # 2^27 (4 GiB per buffer, 30 s of host-side evaluation) is beyond BASELINE's 2^20 .. 2^26: -m gpu_soak
def test_ntt_beyond_three_passes(gm, log_n):
    """2^25 .. 2^27 elements.  Natural order: three passes with radix-512 passes in front (9 + 8 + 8, 9 + 9 + 8, 9 + 9 + 9; k_ntt_pass9, inter-pass
    tables of up to 2^18 entries).  Bit-reversed output: the same plans at 2^25 / 2^26 (their last pass is the radix-256 kernel, which holds that
    address map); bit-reversed input: the radix-512 passes LAST (8 + 8 + 9, 8 + 9 + 9); 2^27 keeps the eight-bit plan in both (a fourth pass of
    degree 3 whose inter-pass twiddle has a 27-bit exponent).  Forward values by direct O(n) evaluation of three outputs, the whole
    inverse(forward(x)) buffer byte for byte, and the bit-reversed orderings against the natural ones at sampled positions."""
    fid = po.F_BN254_FR
    n = 1 << log_n
    lib = ffi.load()
    om = po.root_of_unity(fid, log_n)
    rng = np.random.default_rng(log_n)
    d_a, d_b = DeviceBuffer(n * 32), DeviceBuffer(n * 32)
    ffi.check(lib.panda_gen_scalars(0, 0x2600 + log_n, 0, n, d_a.ptr, NULL_STREAM), "gen")
    x = d_a.to_host().reshape(n, 8)
    flag = C.c_uint(9)
    cfg = ffi.NttconfigurationV1(gm.mem_pool, gm.exec_stream.raw, d_a.ptr, d_b.ptr, C.c_void_p(om.ctypes.data), log_n, C.pointer(flag))
    ffi.check(lib.panda_ntt_execute_bn254_v1(cfg), "ntt")
    assert ntt_passes(log_n) == 3 and flag.value == 1  # three passes: the result is in d_dst (fft.cu:211)
    fwd, other = (d_b, d_a) if flag.value else (d_a, d_b)
    ks = [1, int(rng.integers(0, n))] if log_n >= 26 else [1, n - 1, int(rng.integers(0, n))]  # O(n) on the host each: 5 s at 2^26
    vals = {k: fwd.to_host(nbytes=32, offset=k * 32) for k in ks}
    for k in ks:
        assert (vals[k] == po.ntt_eval_at(fid, x, om, log_n, k)).all(), k
    cfg2 = ffi.NttconfigurationV1(gm.mem_pool, gm.exec_stream.raw, fwd.ptr, other.ptr, C.c_void_p(om.ctypes.data), log_n, C.pointer(flag))
    ffi.check(lib.panda_ntt_execute_bn254_inverse(cfg2), "intt")
    res, spare = (other, fwd) if flag.value else (fwd, other)
    assert np.array_equal(res.to_host().reshape(n, 8), x)
    # bit-reversed output of the same input: y[k] sits at bitrev(k)
    cfg3 = ffi.NttconfigurationV1(gm.mem_pool, gm.exec_stream.raw, res.ptr, spare.ptr, C.c_void_p(om.ctypes.data), log_n, C.pointer(flag))
    ffi.check(lib.panda_ntt_execute_bn254_bitrev_out(cfg3), "ntt")
    br, spare = (spare, res) if flag.value else (res, spare)
    for k in ks:
        pos = int(format(k, f"0{log_n}b")[::-1], 2)
        assert (br.to_host(nbytes=32, offset=pos * 32) == vals[k]).all(), k
    cfg4 = ffi.NttconfigurationV1(gm.mem_pool, gm.exec_stream.raw, br.ptr, spare.ptr, C.c_void_p(om.ctypes.data), log_n, C.pointer(flag))
    ffi.check(lib.panda_ntt_execute_bn254_inverse_bitrev_in(cfg4), "intt")
    back = spare if flag.value else br
    assert np.array_equal(back.to_host().reshape(n, 8), x)
    d_a.free()
    d_b.free()
