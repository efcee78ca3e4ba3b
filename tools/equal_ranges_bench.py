"""Equal point ranges inside one call (panda_msm_execute_from_host with ranges = 0x100 | R and no host source) against the ordinary call, in
alternating blocks: every range gathers rows of W tables x n / R points -- a footprint of 12 GiB / R at 2^24 BN254 points --, and the memory
system serves random rows much faster below ~3.5 GiB (profiles/r05_ubench_gather_rate.txt).   usage: equal_ranges_bench.py <log_n> <R[,R..]> [rounds] [reps] [curve]"""
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

k = int(sys.argv[1])
Rs = [int(x) for x in sys.argv[2].split(",")]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 5
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 9
curve = int(sys.argv[5]) if len(sys.argv) > 5 else 0
pt, res = ((64, 96), (96, 144), (96, 144), (128, 192))[curve]
lib = ffi.load()
gm = pgm.PandaGpuManager(0)
n = 1 << k
db, ds, dr = DeviceBuffer(n * pt), DeviceBuffer(n * 32), DeviceBuffer(res)
ffi.check(lib.panda_gen_bases(curve, 1, 0, n, db.ptr, NULL_STREAM), "gen")
ffi.check(lib.panda_gen_scalars(curve, 2, 0, n, ds.ptr, NULL_STREAM), "gen")
ffi.check(lib.panda_msm_precompute_bases(curve, db.ptr, k, 0, gm.exec_stream.raw), "precompute")
cfg = ffi.MSMConfiguration(gm.mem_pool, gm.exec_stream.raw, db.ptr, ds.ptr, dr.ptr, k, 0)
fn = (lib.panda_msm_execute_bn254, lib.panda_msm_execute_bls12_377, lib.panda_msm_execute_bls12_381, lib.panda_msm_execute_bn254_g2)[curve]


def call(R):
    if R <= 1:
        ffi.check(fn(cfg), "msm")
    else:
        ffi.check(lib.panda_msm_execute_from_host(curve, cfg, None, 0x100 | R, gm.exec_stream.raw), "msm ranges")


for _ in range(10):
    call(1)
res_ms = {R: [] for R in Rs}
for r in range(rounds):
    for R in (Rs if r % 2 == 0 else Rs[::-1]):
        call(R)
        ts = []
        for _ in range(reps):
            t = time.perf_counter()
            call(R)
            ts.append(time.perf_counter() - t)
        ts.sort()
        res_ms[R].append(ts[len(ts) // 2] * 1e3)
for R in Rs:
    v = sorted(res_ms[R])
    print(f"curve {curve} 2^{k} equal ranges R = {R}: median of round medians {v[len(v)//2]:.3f} ms   rounds: " + " ".join(f"{x:.3f}" for x in res_ms[R]), flush=True)
