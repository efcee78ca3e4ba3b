"""The C-ABI boundary without a GPU: every symbol include/panda_interface.h declares is exported by the built
library, the by-value structs have the reference's layout, and the CPU entry points of the library
(panda_msm_execute_*_host, panda_msm_combine_*) agree with the oracle.  No device compute calls."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import oracle as po
from panda_amd import gpu_ffi as ffi
from panda_amd import gpu_manager as pgm
from panda_amd import multi_gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "panda_interface.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(panda_[a-z0-9_]+)\s*\(", txt)))


def test_header_and_python_symbol_lists_agree():
    assert declared_symbols() == sorted(ffi.ALL_SYMBOLS)
    assert len(ffi.REFERENCE_SYMBOLS) == 36  # the reference defines 36


def test_library_exports_every_declared_symbol():
    out = subprocess.run(["nm", "-D", "--defined-only", ffi.LIB_PATH], check=True, capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (panda_[a-z0-9_]+)", out))
    missing = [s for s in declared_symbols() if s not in exported]
    assert not missing, missing
    lib = ffi.load()
    for s in ffi.ALL_SYMBOLS:
        assert hasattr(lib, s)
    assert b"gfx950" in lib.panda_version()


def test_multi_gpu_entry_points_link_rccl():
    """The single-process multi-GPU entry points (csrc/multi_gpu.hip) are part of the library and the exchange they issue is RCCL's:
    the shared object names librccl among its dependencies and imports the collectives it calls; the handle is one pointer."""
    dyn = subprocess.run(["readelf", "-d", ffi.LIB_PATH], check=True, capture_output=True, text=True).stdout
    assert re.search(r"NEEDED.*librccl\.so", dyn), dyn
    und = subprocess.run(["nm", "-D", "--undefined-only", ffi.LIB_PATH], check=True, capture_output=True, text=True).stdout
    for sym in ("ncclCommInitAll", "ncclAllGather", "ncclSend", "ncclRecv", "ncclGroupStart", "ncclGroupEnd"):
        assert re.search(rf"\b{sym}\b", und), sym
    assert C.sizeof(ffi.PandaMultiGpu) == C.sizeof(C.c_void_p)
    for s in ("panda_multi_gpu_create", "panda_msm_execute_bn254_multi", "panda_ntt_execute_bn254_multi", "panda_ntt_execute_bn254_inverse_multi"):
        assert s in ffi.ADDITIVE_SYMBOLS


def test_static_library_is_built_too():
    a = os.path.join(os.path.dirname(ffi.LIB_PATH), "libpanda-cuda.a")
    assert os.path.exists(a)  # `cargo:rustc-link-lib=static=panda-cuda`, build.rs:45


def test_struct_layouts_match_reference():
    # panda_interface.cuh:18-31,70-105 / gpu_ffi/common.rs:40-208 on LP64
    assert C.sizeof(ffi.PandaStream) == C.sizeof(ffi.PandaEvent) == C.sizeof(ffi.PandaMemPool) == 8
    assert C.sizeof(ffi.MSMConfiguration) == 48
    assert [f[0] for f in ffi.MSMConfiguration._fields_] == ["mem_pool", "stream", "bases", "scalars", "results", "log_scalars_count", "msm_result_coordinate_type"]
    assert ffi.MSMConfiguration.log_scalars_count.offset == 40 and ffi.MSMConfiguration.msm_result_coordinate_type.offset == 44
    assert C.sizeof(ffi.NTTConfiguration) == 48 and ffi.NTTConfiguration.log_n.offset == 32 and ffi.NTTConfiguration.flag.offset == 40
    assert C.sizeof(ffi.NttconfigurationV1) == 56 and ffi.NttconfigurationV1.omega.offset == 32 and ffi.NttconfigurationV1.flag.offset == 48


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(ffi, "_lib", None)
    monkeypatch.setattr(ffi, "LIB_PATH", "/nonexistent/libpanda-cuda.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ffi.load()


@pytest.mark.parametrize("cid", [0, 1, 2])
def test_host_debug_entry_point(cid):
    """panda_msm_execute_bn254_host (panda_interface.cu:162-165 -> msm_host.cuh:267-383): all-host pointers."""
    for k, coord in ((10, pgm.JACOBIAN), (11, pgm.PROJECTIVE)):
        n = 1 << k
        bases = po.gen_bases(cid, 700 + k, n)
        scalars = po.gen_scalars(po.FR_OF[cid], 800 + k, n)
        bases[3, : po.LC_Q[cid]] = 0
        scalars[5] = 0
        keep = scalars.copy()
        lib = ffi.load()
        out = np.zeros(3 * po.LC_Q[cid], dtype=np.uint32)
        cfg = ffi.MSMConfiguration(ffi.PandaMemPool(), ffi.PandaStream(), bases.ctypes.data, scalars.ctypes.data, out.ctypes.data, k, coord)
        fn = (lib.panda_msm_execute_bn254_host, lib.panda_msm_execute_bls12_377_host, lib.panda_msm_execute_bls12_381_host)[cid]
        assert fn(cfg) == 0
        got = po.hom_to_affine(cid, out) if coord == pgm.PROJECTIVE else po.to_affine(cid, out)
        assert (got == po.msm_affine(cid, bases, scalars, window_bits=9)).all()
        assert (scalars == keep).all()
    assert fn(ffi.MSMConfiguration()) != 0  # null pointers are rejected with an error code, not a crash


@pytest.mark.parametrize("cid", [0, 1, 2])
def test_combine_partials(cid):
    n = 1 << 10
    bases = po.gen_bases(cid, 900, n)
    scalars = po.gen_scalars(po.FR_OF[cid], 901, n)
    parts = []
    for r in range(4):
        first, cnt = multi_gpu.shard_range(n, 4, r)
        parts.append(pgm.panda_msm_bn254_gpu_host(None, scalars[first:first + cnt], bases[first:first + cnt], curve=cid))
    parts.append(np.zeros_like(parts[0]))  # an identity partial
    total = multi_gpu.combine_partials(np.stack(parts), curve=cid)
    assert (po.to_affine(cid, total.view(np.uint32)) == po.msm_affine(cid, bases, scalars, window_bits=9)).all()
    hom = multi_gpu.combine_partials(np.stack(parts), curve=cid, coordinate_type=pgm.PROJECTIVE)
    assert (po.hom_to_affine(cid, hom.view(np.uint32)) == po.to_affine(cid, total.view(np.uint32))).all()


def test_config1_host_entry_2_16_vs_oracle_16_bit_windows():
    """BASELINE config 1 (BN254 MSM 2^16 random scalars/bases via the CPU host-debug path, no GPU): the product's
    panda_msm_execute_bn254_host against the oracle's restatement of the reference algorithm at its own window width
    (BIT_S = 16, msm_config.cuh:7; msm_host.cuh:267-370), plus the linearity identity.  test_msm_bn254_correctness_host
    (tests/test.rs:115-194) stops at the same size."""
    n = 1 << 16
    bases = po.gen_bases(0, 116, n)
    scalars = po.gen_scalars(po.F_BN254_FR, 216, n)
    keep = scalars.copy()
    want = po.msm_affine(0, bases, scalars, window_bits=16, threads=8)
    assert (want == po.expected_from_linearity(0, 116, scalars)).all()
    out = pgm.panda_msm_bn254_gpu_host(None, scalars, bases)
    assert (po.to_affine(0, out.view(np.uint32)) == want).all()
    assert (scalars == keep).all()


def test_gpu_manager_helpers():
    assert pgm.log_2(1) == 0 and pgm.log_2(1024) == 10 and pgm.log_2(1500) == 10  # gpu_manager/common.rs:5-15
    assert pgm.FIELD_ELEMENT_LEN == 32


def test_header_is_plain_c_and_links(tmp_path):
    """include/panda_interface.h is a C header (the reference's is consumed by bindgen-style Rust declarations): it compiles as
    C11 with -Wall -Wextra -Werror, and a C translation unit that names every declared function links against the library."""
    import re
    import subprocess
    header = open(os.path.join(ROOT, "include", "panda_interface.h")).read()
    names = sorted(set(re.findall(r"\b(panda_[a-z0-9_]+)\s*\(", header)) - {"panda_error", "panda_stream", "panda_event", "panda_mem_pool"})
    assert set(names) == set(ffi.ALL_SYMBOLS)
    src = tmp_path / "abi.c"
    src.write_text('#include "panda_interface.h"\n#include <stdio.h>\nint main(void) {\n    const void *fns[] = {' +
                   ", ".join(f"(const void *){n}" for n in names) + "};\n    size_t n = sizeof(fns) / sizeof(fns[0]), i, ok = 0;\n"
                   "    for (i = 0; i < n; i++) ok += fns[i] != 0;\n    printf(\"%zu\\n\", ok);\n    return ok == n ? 0 : 1;\n}\n")
    libdir = os.path.dirname(ffi.LIB_PATH)
    exe = tmp_path / "abi"
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-Wno-pedantic", "-I", os.path.join(ROOT, "include"), str(src), "-L", libdir,
                    "-lpanda-cuda", "-L/opt/rocm/lib", "-lamdhip64", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout
    assert int(out) == len(names)


def test_g2_reference_is_pinned():
    """The pure-Python G2 arithmetic the G2 tests lean on: the standard generator lies on the twist y^2 = x^3 + 3/(9+u) and has
    order r; addition is consistent with scalar multiplication."""
    import pyref
    c = pyref.CURVES[0]
    G = pyref.G2_GEN
    assert pyref.g2_is_on_curve(G)
    assert pyref.g2_mul(c.r, G) is None
    assert pyref.g2_add(pyref.g2_mul(5, G), pyref.g2_mul(7, G)) == pyref.g2_mul(12, G)
    assert pyref.g2_add(pyref.g2_mul(c.r - 1, G), G) is None


def test_g2_host_entry_point_vs_python_reference():
    """panda_msm_execute_bn254_g2_host (product code: the same fe29 / Fq2 arithmetic as the kernels, compiled for the host) against
    the Python reference: random multiples of the generator incl. an identity base, P and -P, repeated points; edge scalars; both
    output coordinate systems; and the combine entry point."""
    import pyref
    c = pyref.CURVES[0]
    rng = np.random.default_rng(0x62)
    n = 64
    mult = [int(v) for v in rng.integers(1, 1 << 62, n)]
    pts = [pyref.g2_mul(m, pyref.G2_GEN) for m in mult]
    pts[7] = None                                          # identity base (x == 0 on the wire)
    pts[9] = pts[8]                                        # repeated point
    pts[11] = (pts[10][0], pyref.f2_sub((0, 0), pts[10][1], c.p))  # P, -P
    bases = np.stack([pyref.g2_encode_affine(P) for P in pts])
    scalars = po.gen_scalars(po.F_BN254_FR, 0x63, n)
    mont = lambda v: pyref.int_to_limbs(v * c.Rr % c.r, 8)
    for i, v in enumerate([0, 1, c.r - 1, 2, (1 << 253) + 12345]):
        scalars[20 + i] = mont(v % c.r)
    scalars[11] = scalars[10]
    scalars[9] = scalars[8]
    want = pyref.g2_msm(bases, scalars)
    out = pgm.panda_msm_bn254_gpu_host(None, scalars, bases, curve=pgm.BN254_G2)
    assert out.size == 192
    assert pyref.g2_decode_jacobian(out.view(np.uint32)) == want
    lib = ffi.load()
    hom = np.zeros(192, np.uint8)
    cfg = ffi.MSMConfiguration(ffi.PandaMemPool(), ffi.PandaStream(), C.c_void_p(bases.ctypes.data), C.c_void_p(scalars.ctypes.data),
                               C.c_void_p(hom.ctypes.data), 6, pgm.PROJECTIVE)
    ffi.check(lib.panda_msm_execute_bn254_g2_host(cfg), "host")
    assert pyref.g2_decode_homogeneous(hom.view(np.uint32)) == want
    halves = np.stack([pgm.panda_msm_bn254_gpu_host(None, scalars[h * 32:(h + 1) * 32], bases[h * 32:(h + 1) * 32], curve=pgm.BN254_G2).view(np.uint32)
                       for h in range(2)])
    total = multi_gpu.combine_partials(halves, curve=pgm.BN254_G2)
    assert pyref.g2_decode_jacobian(total.view(np.uint32)) == want


def test_clock_delta_pairs_stamps_per_cu_and_reports_the_slowest_xcd():
    """panda_clock_delta is host code (no GPU needed): two stamp blocks of [xcc 0..7][256 CU slots][s_memtime, s_memrealtime] -> cycles of the XCD
    that showed the fewest, ticks, XCDs paired, mean over the XCDs, per-XCD cycles.  Stamps are paired per CU slot: a CU stamped on one side
    only is ignored, and CU counters with wildly different offsets (s_memtime is per CU) do not disturb the deltas."""
    import numpy as np

    lib = ffi.load()
    rng = np.random.default_rng(6)
    before = np.zeros((8, 256, 2), dtype=np.uint64)
    after = np.zeros((8, 256, 2), dtype=np.uint64)
    clocks = [2049, 1989, 2025, 1943, 1999, 1949, 2003, 1941]  # MHz per XCD
    ticks = 1_400_000  # 14 ms in 10 ns ticks
    for x in range(8):
        for cu in range(32):
            slot = (cu % 16) | ((cu // 16) << 5)            # some HW_ID[15:8] pattern
            base = int(rng.integers(1 << 40, 1 << 48))      # every CU's counter has its own offset
            t0 = 5_000_000_000 + int(rng.integers(0, 50))
            before[x, slot] = (base, t0)
            after[x, slot] = (base + clocks[x] * ticks // 100 + int(rng.integers(0, 2000)), t0 + ticks + int(rng.integers(0, 20)))
    before[3, 200] = (123, 456)          # stamped before only: ignored
    after[5, 201] = (999, 5_100_000_000)  # stamped after only: ignored
    out = (C.c_uint64 * ffi.CLOCK_WORDS)()
    assert lib.panda_clock_delta(C.c_void_p(before.ctypes.data), C.c_void_p(after.ctypes.data), out) == 0
    cycles, tk, xcds, mean = out[0], out[1], out[2], out[3]
    per = list(out[4:12])
    assert xcds == 8 and abs(tk - ticks) < 40
    for x in range(8):
        assert abs(per[x] - clocks[x] * ticks // 100) < 3000, (x, per[x])
    assert cycles == min(per) and per.index(cycles) == 7            # XCD 7 holds the lowest clock
    assert abs(mean - sum(per) // 8) <= 1
    assert abs(mean / tk * 100.0 - sum(clocks) / 8) < 1.0           # MHz of the box
    # nothing paired: all zeros, not garbage
    assert lib.panda_clock_delta(C.c_void_p(before.ctypes.data), C.c_void_p(before.ctypes.data), out) == 0 and list(out) == [0] * ffi.CLOCK_WORDS
    assert lib.panda_clock_delta(None, C.c_void_p(after.ctypes.data), out) == 1
