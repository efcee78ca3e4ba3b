"""Cost of the point-range schedule by itself: panda_msm_execute_from_host with h_scalars = NULL (resident scalars, nothing to
overlap) for R = 1 ... 6 ranges, BN254 2^24 with precomputed tables.  usage: python tools/range_bench.py [log_n]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from panda_amd import gpu_ffi as ffi  # noqa: E402

if os.environ.get("PANDA_LIB"):  # timing experiments: another build of the library
    ffi.LIB_PATH = os.environ["PANDA_LIB"]
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n = 1 << log_n
lib = ffi.load()
    lib.panda_msm_set_phase_timing(2)
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
ps = ffi.PandaStream(stream.cuda_stream)
bases = torch.empty(n * 64, dtype=torch.uint8, device=dev)
scalars = torch.empty(n * 32, dtype=torch.uint8, device=dev)
result = torch.zeros(96, dtype=torch.uint8, device=dev)
ffi.check(lib.panda_gen_bases(0, 1, 0, n, bases.data_ptr(), ps), "gen")
ffi.check(lib.panda_gen_scalars(0, 2, 0, n, scalars.data_ptr(), ps), "gen")
ffi.check(lib.panda_msm_precompute_bases(0, bases.data_ptr(), log_n, 0, ps), "pre")
cfg = ffi.MSMConfiguration(ffi.PandaMemPool(), ps, bases.data_ptr(), scalars.data_ptr(), result.data_ptr(), log_n, 0)
ref = None
for R in ([1] if os.environ.get("PANDA_LIB") else [1, 2, 3, 4, 5, 6]):
    best = 1e9
    for _ in range(4):
        torch.cuda.synchronize()
        t = time.perf_counter()
        ffi.check(lib.panda_msm_execute_from_host(0, cfg, None, R, ps), "msm")
        best = min(best, time.perf_counter() - t)
    ph = (C.c_float * 8)()
    lib.panda_msm_last_phase_ms(ph)
    print(f"ranges {R}: {best * 1e3:.3f} ms   last-range phases {[round(v, 3) for v in ph]}", flush=True)
