"""k_accumulate of the tabled BN254 MSM at 2^log_n points with ANY build of the library (PANDA_LIB=<path>: this round's, or an earlier
round's built from its commit): per launch the HIP-event milliseconds and -- when the build has them -- the in-kernel clock stamps
(panda_set_clock_stamps: shader cycles and 10 ns ticks around the launch).  Run under `rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace`
(tools/clock_check.sh) the counter gives the same launches' cycles from outside: GRBM_GUI_ACTIVE / 8 (the profiler sums the 8 XCDs).
Binds only the handful of symbols it needs, so that a round-4 library (fewer exports) loads.
usage: clock_check.py [log_n=24] [launches=12] [curve=0]   (curve 1: BLS12-377, 2: BLS12-381)"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.environ.get("PANDA_LIB") or os.path.join(ROOT, "panda_amd", "csrc", "libpanda-cuda.so")


class Handle(C.Structure):
    _fields_ = [("handle", C.c_void_p)]


class Cfg(C.Structure):
    _fields_ = [("mem_pool", Handle), ("stream", Handle), ("bases", C.c_void_p), ("scalars", C.c_void_p), ("results", C.c_void_p), ("log_scalars_count", C.c_uint),
                ("coord", C.c_int)]


def main():
    log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    launches = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    curve = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    point = (64, 96, 96)[curve]
    try:
        import torch  # noqa: F401  (one libamdhip64 in the process, as panda_amd.gpu_ffi does)
    except Exception:
        pass
    lib = C.CDLL(LIB)
    u, u64, vp = C.c_uint, C.c_uint64, C.c_void_p
    lib.panda_malloc.argtypes = [C.POINTER(vp), C.c_size_t]
    lib.panda_gen_bases.argtypes = lib.panda_gen_scalars.argtypes = [u, u64, u64, u64, vp, Handle]
    lib.panda_msm_precompute_bases.argtypes = [u, vp, u, u, Handle]
    execute = (lib.panda_msm_execute_bn254, lib.panda_msm_execute_bls12_377, lib.panda_msm_execute_bls12_381)[curve]
    execute.argtypes = [Cfg]
    lib.panda_msm_set_phase_timing.argtypes = [u]
    lib.panda_msm_last_phase_ms.argtypes = [C.POINTER(C.c_float)]
    lib.panda_stream_create.argtypes = [C.POINTER(Handle), C.c_bool]

    def ok(code, what):
        if code != 0:
            raise SystemExit(f"{what} failed: {code}")

    n = 1 << log_n
    db, ds, dr = vp(), vp(), vp()
    for p, size in ((db, n * point), (ds, n * 32), (dr, 144)):
        ok(lib.panda_malloc(C.byref(p), size), "malloc")
    null, stream = Handle(), Handle()
    ok(lib.panda_stream_create(C.byref(stream), False), "stream")
    ok(lib.panda_gen_bases(curve, 1, 0, n, db, null), "gen_bases")
    ok(lib.panda_gen_scalars(curve, 2, 0, n, ds, null), "gen_scalars")
    ok(lib.panda_msm_precompute_bases(curve, db, log_n, 0, stream), "precompute")
    stamps = hasattr(lib, "panda_set_clock_stamps")
    if stamps:
        lib.panda_set_clock_stamps.argtypes = [u]
        lib.panda_msm_last_clock.argtypes = [C.POINTER(u64)]
        lib.panda_set_clock_stamps(1)
    lib.panda_msm_set_phase_timing(1)
    cfg = Cfg(Handle(), stream, db, ds, dr, log_n, 0)
    ph, clk = (C.c_float * 8)(), (u64 * 12)()
    rows = []
    for i in range(launches + 3):
        ok(execute(cfg), "msm")
        lib.panda_msm_last_phase_ms(ph)
        row = {"k_accumulate_ms": round(ph[3], 4), "device_ms": round(ph[7], 4)}
        if stamps:
            lib.panda_msm_last_clock(clk)
            if clk[1]:
                row.update({"mcycles": round(clk[0] / 1e6, 4), "mcycles_mean": round(clk[3] / 1e6, 4), "sclk_mhz": round(clk[3] / clk[1] * 100.0, 1), "stamp_ms": round(clk[1] * 1e-5, 4),
                            "xcds": int(clk[2]), "mhz_min": round(min(clk[4:12]) / clk[1] * 100.0, 1), "mhz_max": round(max(clk[4:12]) / clk[1] * 100.0, 1)})
        if i >= 3:  # the first launches carry the chip from idle to its sustained clock
            rows.append(row)
    mean = {k: round(sum(r[k] for r in rows) / len(rows), 4) for k in rows[0]}
    print("CLOCK_CHECK " + json.dumps({"lib": os.path.basename(LIB), "log_n": log_n, "curve": curve, "launches": len(rows), "mean": mean, "rows": rows}), flush=True)


if __name__ == "__main__":
    main()
