"""Large-size checks on one GPU (development aid): BN254 MSM 2^24 / 2^26 and BLS12-377 MSM 2^24 against the linearity
identity MSM(s, m*G) = (sum s_i m_i)*G, with timings."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import oracle as po  # noqa: E402
from gpu_util import NULL_STREAM, DeviceBuffer  # noqa: E402
from panda_amd import gpu_ffi as ffi  # noqa: E402
from panda_amd import gpu_manager as pgm  # noqa: E402


def run(gm, cid, k, check=True):
    lib = ffi.load()
    lib.panda_msm_set_phase_timing(2)
    n = 1 << k
    lc = po.LC_Q[cid]
    db, ds, dr = DeviceBuffer(n * 2 * lc * 4), DeviceBuffer(n * 32), DeviceBuffer(3 * lc * 4)
    ffi.check(lib.panda_gen_bases(cid, 0xB16 + k, 0, n, db.ptr, NULL_STREAM), "gen")
    ffi.check(lib.panda_gen_scalars(cid, 0x5CA + k, 0, n, ds.ptr, NULL_STREAM), "gen")
    cfg = ffi.MSMConfiguration(gm.mem_pool, gm.exec_stream.raw, db.ptr, ds.ptr, dr.ptr, k, 0)
    fn = lib.panda_msm_execute_bn254 if cid == 0 else lib.panda_msm_execute_bls12_377
    best = 1e9
    for r in range(3):
        t = time.time()
        ffi.check(fn(cfg), "msm")
        best = min(best, time.time() - t)
    ms = (C.c_float * 8)()
    lib.panda_msm_last_phase_ms(ms)
    out = dr.to_host()
    ok = None
    if check:
        t = time.time()
        scalars = ds.to_host().reshape(n, 8)
        want = po.expected_from_linearity(cid, 0xB16 + k, scalars)
        ok = bool((po.to_affine(cid, out) == want).all())
        tc = time.time() - t
    print(f"curve {cid} 2^{k}: {best*1e3:9.2f} ms  {n/best/1e6:8.1f} Mpts/s  accumulate {ms[3]:.2f} ms  parity(linearity)={ok}" + (f" (cpu check {tc:.1f}s)" if check else ""), flush=True)
    for d in (db, ds, dr):
        d.free()
    return ok


def main():
    gm = pgm.PandaGpuManager(0)
    cases = [(0, 24), (1, 20), (1, 24), (0, 26)]
    if len(sys.argv) > 1:
        cases = [tuple(int(x) for x in c.split(":")) for c in sys.argv[1].split(",")]
    bad = 0
    for cid, k in cases:
        if run(gm, cid, k) is False:
            bad += 1
    gm.deinit()
    sys.exit(bad)


if __name__ == "__main__":
    main()
