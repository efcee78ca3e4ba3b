"""Per-phase timing of the MSM with precomputed window tables against the plain registered path (development aid).
usage: tabled_bench.py <log_n[,log_n..]> <window_bits[,..] (0 = policy)> <unused> [reps] [curve 0|1|2] [chunk[,..] (0 = policy)]
environment: PANDA_TIMING=0|1|2 (phase timers a call records, default 2 here; the wall time is the honest figure at 0 / 1)"""
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


def run(lib, cfg, reps, names, fn=None):
    fn = fn or lib.panda_msm_execute_bn254
    best = None
    for r in range(reps + 1):
        t = time.time()
        ffi.check(fn(cfg), "msm")
        dt = time.time() - t
        ms = (C.c_float * 8)()
        lib.panda_msm_last_phase_ms(ms)
        if r and (best is None or dt < best[0]):
            best = (dt, list(ms))
    dt, ms = best
    return f"wall {dt*1e3:8.3f} ms | " + " ".join(f"{nm}={v:.3f}" for nm, v in zip(names, ms))


def main():
    ks = [int(x) for x in sys.argv[1].split(",")]
    wbs = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0").split(",")]
    groups = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "0").split(",")]
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    curve = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    chunks = [int(x) for x in (sys.argv[6] if len(sys.argv) > 6 else "0").split(",")]
    pt, res = (64, 96) if curve == 0 else (96, 144)
    lib = ffi.load()
    lib.panda_msm_set_phase_timing(int(os.environ.get("PANDA_TIMING", "2")))
    fn = (lib.panda_msm_execute_bn254, lib.panda_msm_execute_bls12_377, lib.panda_msm_execute_bls12_381)[curve]
    gm = pgm.PandaGpuManager(0)
    names = [lib.panda_msm_phase_name(i).decode() for i in range(8)]
    for k in ks:
        n = 1 << k
        db, ds, dr = DeviceBuffer(n * pt), DeviceBuffer(n * 32), DeviceBuffer(res)
        ffi.check(lib.panda_gen_bases(curve, 1, 0, n, db.ptr, NULL_STREAM), "gen")
        ffi.check(lib.panda_gen_scalars(curve, 2, 0, n, ds.ptr, NULL_STREAM), "gen")
        cfg = ffi.MSMConfiguration(gm.mem_pool, gm.exec_stream.raw, db.ptr, ds.ptr, dr.ptr, k, 0)
        ffi.check(lib.panda_msm_register_bases(curve, db.ptr, k, gm.exec_stream.raw), "register")
        for pc in [int(x) for x in os.environ.get("PANDA_PLAIN_C", "0").split(",")]:  # plain-path window widths to force (0 = policy)
            lib.panda_msm_set_window_bits(pc)
            print(f"curve {curve} 2^{k} plain registered c={pc:2d}: " + run(lib, cfg, reps, names, fn), flush=True)
        lib.panda_msm_set_window_bits(0)
        ref = dr.to_host().tobytes()
        ffi.check(lib.panda_msm_unregister_bases(db.ptr), "unregister")
        for wb in wbs:
            t = time.time()
            ffi.check(lib.panda_msm_precompute_bases(curve, db.ptr, k, wb, gm.exec_stream.raw), "precompute")
            tb = time.time() - t
            tables, bits, held = C.c_uint(0), C.c_uint(0), C.c_size_t(0)
            lib.panda_msm_registered_info(db.ptr, C.byref(tables), C.byref(bits), C.byref(held))
            for g in groups:
                for ck in chunks:
                    lib.panda_msm_set_chunk_entries(ck)
                    line = run(lib, cfg, reps, names, fn)
                    same = "same-jacobian" if dr.to_host().tobytes() == ref else "different representative"
                    print(f"2^{k} tables c={bits.value:2d} W={tables.value:2d} group={g:2d} chunk={ck:3d} ({held.value/2**30:.1f} GiB, built in {tb:.2f}s): {line}  [{same}]", flush=True)
            lib.panda_msm_set_chunk_entries(0)
            ffi.check(lib.panda_msm_unregister_bases(db.ptr), "unregister")
        for d in (db, ds, dr):
            d.free()
    gm.deinit()


if __name__ == "__main__":
    main()
