"""[needs tools/xcd_balance.patch applied to the library: the experiment of profiles/r06_xcd_balance.txt was not kept]
XCD balance (panda_msm_set_xcd_balance: chunk lengths that follow the XCDs' clocks) against equal chunks in alternating blocks inside one process, tabled MSM:
wall time of the call, HIP-event time of k_accumulate, its cycles (slowest XCD) and the mean clock.  usage: xcd_balance_ab.py <mode: 1 | 2> <log_n[,..]> [rounds=6] [reps=5] [curve=0]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as po  # noqa: E402  (the checker: affine form of the results)
from gpu_util import NULL_STREAM, DeviceBuffer  # noqa: E402
from panda_amd import gpu_ffi as ffi  # noqa: E402
from panda_amd import gpu_manager as pgm  # noqa: E402


def med(v):
    v = sorted(v)
    return v[len(v) // 2]


def main():
    variant = int(sys.argv[1])
    ks = [int(x) for x in sys.argv[2].split(",")]
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
    curve = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    pt, res = ((64, 96), (96, 144), (96, 144), (128, 192))[curve]
    lib = ffi.load()
    fn = (lib.panda_msm_execute_bn254, lib.panda_msm_execute_bls12_377, lib.panda_msm_execute_bls12_381, lib.panda_msm_execute_bn254_g2)[curve]
    gm = pgm.PandaGpuManager(0)
    lib.panda_msm_set_phase_timing(1)
    lib.panda_set_clock_stamps(1)
    for k in ks:
        n = 1 << k
        db, ds, dr = DeviceBuffer(n * pt), DeviceBuffer(n * 32), DeviceBuffer(res)
        ffi.check(lib.panda_gen_bases(curve, 1, 0, n, db.ptr, NULL_STREAM), "gen")
        ffi.check(lib.panda_gen_scalars(curve, 2, 0, n, ds.ptr, NULL_STREAM), "gen")
        cfg = ffi.MSMConfiguration(gm.mem_pool, gm.exec_stream.raw, db.ptr, ds.ptr, dr.ptr, k, 0)
        ffi.check(lib.panda_msm_precompute_bases(curve, db.ptr, k, 0, gm.exec_stream.raw), "precompute")
        ph, clk = (C.c_float * 8)(), (C.c_uint64 * 12)()
        est = (C.c_uint * 8)()
        acc = {0: [], variant: []}
        ref = None
        for r in range(rounds):
            for v in ((0, variant) if r % 2 == 0 else (variant, 0)):
                saved = {}
                ffi.check(lib.panda_msm_set_xcd_balance(v), "mode")
                if v == 1:  # a new setting starts from equal chunks: let the estimate form (the first call is stamped)
                    ffi.check(fn(cfg), "msm")
                rows = []
                for i in range(reps + 2):
                    t = time.perf_counter()
                    ffi.check(fn(cfg), "msm")
                    dt = time.perf_counter() - t
                    lib.panda_msm_last_phase_ms(ph)
                    lib.panda_msm_last_clock(clk)
                    if i >= 2:
                        rows.append((dt * 1e3, ph[3], clk[0] / 1e6, ph[7], clk[3] / max(clk[1], 1) * 100.0))
                if v:
                    lib.panda_msm_last_chunk_lengths(est)
                out = po.to_affine(curve, dr.to_host()).tobytes()
                ref = ref or out
                assert out == ref, "DIFFERENT RESULT"
                acc[v].append(tuple(med([x[j] for x in rows]) for j in range(5)))
        lib.panda_msm_set_xcd_balance(1)
        print('   clocks of the XCDs as this thread last measured them (1/1000 of their mean):', list(est), flush=True)
        for v in (0, variant):
            m = [med([x[j] for x in acc[v]]) for j in range(5)]
            print(f"curve {curve} 2^{k} xcd balance {v}: wall {m[0]:8.4f} ms  k_accumulate {m[1]:8.4f} ms  {m[2]:8.4f} Mcycles (slowest XCD) at {m[4]:5.0f} MHz (mean)  device total {m[3]:8.4f} ms"
                  f"   (medians of {rounds} block medians; same point)", flush=True)
        ffi.check(lib.panda_msm_unregister_bases(db.ptr), "unregister")
        for d in (db, ds, dr):
            d.free()
    gm.deinit()


if __name__ == "__main__":
    main()
