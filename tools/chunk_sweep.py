"""Sorted entries per k_accumulate thread (panda_msm_set_chunk_entries) against the policy, interleaved: the chip's clock drifts by several
per cent over a run of calls, so every candidate is timed in turn, round after round, and the medians of the rounds are compared.
usage: chunk_sweep.py <log_n> <chunk[,chunk..]> [rounds] [reps]      (0 = policy)"""
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
chunks = [int(x) for x in sys.argv[2].split(",")]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 6
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 15
lib = ffi.load()
gm = pgm.PandaGpuManager(0)
n = 1 << k
db, ds, dr = DeviceBuffer(n * 64), DeviceBuffer(n * 32), DeviceBuffer(96)
ffi.check(lib.panda_gen_bases(0, 1, 0, n, db.ptr, NULL_STREAM), "gen")
ffi.check(lib.panda_gen_scalars(0, 2, 0, n, ds.ptr, NULL_STREAM), "gen")
ffi.check(lib.panda_msm_precompute_bases(0, db.ptr, k, 0, gm.exec_stream.raw), "precompute")
cfg = ffi.MSMConfiguration(gm.mem_pool, gm.exec_stream.raw, db.ptr, ds.ptr, dr.ptr, k, 0)
for _ in range(30):
    lib.panda_msm_execute_bn254(cfg)
res = {c: [] for c in chunks}
for r in range(rounds):
    for c in (chunks if r % 2 == 0 else chunks[::-1]):
        lib.panda_msm_set_chunk_entries(c)
        lib.panda_msm_execute_bn254(cfg)
        ts = []
        for _ in range(reps):
            t = time.perf_counter()
            ffi.check(lib.panda_msm_execute_bn254(cfg), "msm")
            ts.append(time.perf_counter() - t)
        ts.sort()
        res[c].append(ts[len(ts) // 2] * 1e3)
lib.panda_msm_set_chunk_entries(0)
for c in chunks:
    v = sorted(res[c])
    print(f"2^{k} chunk {c:3d}: median of {rounds} round medians {v[len(v)//2]:.3f} ms   rounds: " + " ".join(f"{x:.3f}" for x in res[c]), flush=True)
