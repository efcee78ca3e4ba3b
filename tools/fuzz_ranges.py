"""Randomised soak of panda_msm_execute_from_host (point-range pipeline) against the linearity identity: sizes 2^16 ... 2^21, every
range count, tables / converted-only / unregistered bases, pinned / pageable / resident scalars, skewed scalar patterns, all four curve
ids.  Development aid, not part of the suites.   usage: fuzz_ranges.py <cases> [seed]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import oracle as po  # noqa: E402
import pyref  # noqa: E402
from gpu_util import NULL_STREAM, DeviceBuffer  # noqa: E402
from panda_amd import gpu_ffi as ffi  # noqa: E402
from panda_amd import gpu_manager as pgm  # noqa: E402

POINT = {0: 64, 1: 96, 2: 96, 3: 128}
RESULT = {0: 96, 1: 144, 2: 144, 3: 192}


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    gm = pgm.PandaGpuManager(0)
    lib = ffi.load()
    t0 = time.time()
    bad = 0
    for it in range(cases):
        cid = int(rng.choice([0, 0, 0, 1, 2, 3]))
        fr = 0 if cid == 3 else cid
        c = pyref.CURVES[fr]
        k = int(rng.integers(16, 22 if cid == 0 else 19))
        n = 1 << k
        seed = int(rng.integers(1, 1 << 40))
        db, ds, dr = DeviceBuffer(n * POINT[cid]), DeviceBuffer(n * 32), DeviceBuffer(RESULT[cid])
        ffi.check(lib.panda_gen_bases(cid, seed, 0, n, db.ptr, NULL_STREAM), "gen")
        ffi.check(lib.panda_gen_scalars(cid, seed + 1, 0, n, ds.ptr, NULL_STREAM), "gen")
        scalars = ds.to_host().reshape(n, 8)
        mont = lambda v: pyref.int_to_limbs(v * c.Rr % c.r, 8)
        pattern = int(rng.integers(0, 5))
        if pattern == 1:
            scalars[rng.random(n) < 0.8] = 0
        elif pattern == 2:  # a few distinct values: huge buckets in every range
            vals = [mont(int.from_bytes(rng.bytes(32), "little") % c.r) for _ in range(int(rng.integers(1, 4)))]
            scalars[:] = np.stack(vals)[rng.integers(0, len(vals), n)]
        elif pattern == 3:  # one range all zero, another all equal
            scalars[: n // 4] = 0
            scalars[n // 2:] = scalars[n // 2]
        elif pattern == 4:
            scalars[:] = mont(c.r - 1)
        mode = int(rng.integers(0, 3))  # 0 unregistered, 1 converted only, 2 tables
        wb = int(rng.choice([0, 0, 12, 14, 16, 18, 20]))
        if mode == 1:
            ffi.check(lib.panda_msm_register_bases(cid, db.ptr, k, gm.exec_stream.raw), "register")
        elif mode == 2 and lib.panda_msm_precompute_bases(cid, db.ptr, k, wb, gm.exec_stream.raw) != 0:
            mode = 0
        source = int(rng.integers(0, 3))  # 0 pageable, 1 pinned, 2 resident
        pinned = C.c_void_p()
        if source == 1:
            ffi.check(lib.panda_malloc_host(C.byref(pinned), n * 32), "pin")
            C.memmove(pinned, scalars.ctypes.data, n * 32)
            h = pinned
        elif source == 0:
            h = C.c_void_p(scalars.ctypes.data)
        else:
            h = None
            ffi.check(lib.panda_memcpy(ds.ptr, C.c_void_p(scalars.ctypes.data), n * 32), "copy")
        if source != 2:
            ffi.check(lib.panda_memset(ds.ptr, 0x5A, n * 32), "memset")
        ranges = int(rng.integers(1, 9))
        coord = int(rng.integers(0, 2))
        cfg = ffi.MSMConfiguration(gm.mem_pool, gm.exec_stream.raw, db.ptr, ds.ptr, dr.ptr, k, coord)
        ffi.check(lib.panda_msm_execute_from_host(cid, cfg, h, ranges, gm.h2d_stream.raw), "msm")
        out = dr.to_host()
        if cid == 3:
            kk = pyref.limbs_to_int(po.linear_combination(0, seed, scalars))
            want = pyref.g2_mul(kk, pyref.G2_GEN)
            got = pyref.g2_decode_homogeneous(out) if coord else pyref.g2_decode_jacobian(out)
            ok = got == want
        else:
            want = po.expected_from_linearity(cid, seed, scalars)
            got = po.hom_to_affine(cid, out) if coord else po.to_affine(cid, out)
            ok = bool((got == want).all())
        if not ok:
            bad += 1
            print("MISMATCH", dict(cid=cid, k=k, seed=seed, pattern=pattern, mode=mode, wb=wb, source=source, ranges=ranges, coord=coord), flush=True)
        if mode:
            lib.panda_msm_unregister_bases(db.ptr)
        if pinned:
            lib.panda_free_host(pinned)
        for d in (db, ds, dr):
            d.free()
        if it % 10 == 9:
            print(f"{it + 1} cases, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
    gm.deinit()
    print(f"done: {cases} cases, {bad} mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
