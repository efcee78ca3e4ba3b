"""The sort of the rest of the bucket space beside the accumulation of the front (panda_msm_set_overlap): wall time of the tabled
MSM per (front / 128, workgroups per CU) against the serial schedule, result bytes compared (development aid).
usage: overlap_bench.py <log_n[,log_n..]> <front:wgs[,front:wgs..]> [reps] [curve]      (0:0 = serial)
environment: PANDA_TIMING=0|1|2 (phase timers, default 0: the wall time is the honest figure)"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from gpu_util import NULL_STREAM, DeviceBuffer  # noqa: E402
from panda_amd import gpu_ffi as ffi  # noqa: E402
from panda_amd import gpu_manager as pgm  # noqa: E402


if os.environ.get("PANDA_LIB"):  # timing experiments: another build of the library
    ffi.LIB_PATH = os.environ["PANDA_LIB"]


def main():
    ks = [int(x) for x in sys.argv[1].split(",")]
    combos = [tuple(int(v) for v in x.split(":")) for x in sys.argv[2].split(",")]
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 7
    curve = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    pt, res = ((64, 96), (96, 144), (96, 144), (128, 192))[curve]
    lib = ffi.load()
    timing = int(os.environ.get("PANDA_TIMING", "0"))
    lib.panda_msm_set_phase_timing(timing)
    lib.panda_msm_set_accumulate_variant(int(os.environ.get("PANDA_ACC_VARIANT", "0")))
    fn = (lib.panda_msm_execute_bn254, lib.panda_msm_execute_bls12_377, lib.panda_msm_execute_bls12_381, lib.panda_msm_execute_bn254_g2)[curve]
    gm = pgm.PandaGpuManager(0)
    names = [lib.panda_msm_phase_name(i).decode() for i in range(8)]
    for k in ks:
        n = 1 << k
        db, ds, dr = DeviceBuffer(n * pt), DeviceBuffer(n * 32), DeviceBuffer(res)
        ffi.check(lib.panda_gen_bases(curve, 1, 0, n, db.ptr, NULL_STREAM), "gen")
        ffi.check(lib.panda_gen_scalars(curve, 2, 0, n, ds.ptr, NULL_STREAM), "gen")
        cfg = ffi.MSMConfiguration(gm.mem_pool, gm.exec_stream.raw, db.ptr, ds.ptr, dr.ptr, k, 0)
        ffi.check(lib.panda_msm_precompute_bases(curve, db.ptr, k, 0, gm.exec_stream.raw), "precompute")
        ref = None
        for front, wgs in combos:
            ffi.check(lib.panda_msm_set_overlap(front, wgs), "set_overlap")
            ts = []
            for r in range(reps + 2):
                t = time.perf_counter()
                ffi.check(fn(cfg), "msm")
                dt = time.perf_counter() - t
                if r >= 2:
                    ts.append(dt)
            ts.sort()
            ms = (C.c_float * 8)()
            lib.panda_msm_last_phase_ms(ms)
            out = dr.to_host().tobytes()
            if ref is None:
                ref = out
            ph = " ".join(f"{nm}={v:.3f}" for nm, v in zip(names, ms)) if timing else ""
            print(f"curve {curve} 2^{k} front {front:3d}/128 wgs/CU {wgs}: median {ts[len(ts)//2]*1e3:8.3f} ms  min {ts[0]*1e3:8.3f}  "
                  f"{'same bytes' if out == ref else 'DIFFERENT RESULT'}  {ph}", flush=True)
        lib.panda_msm_set_overlap(0xFFFFFFFF, 0)
        ffi.check(lib.panda_msm_unregister_bases(db.ptr), "unregister")
        for d in (db, ds, dr):
            d.free()
    gm.deinit()


if __name__ == "__main__":
    main()
