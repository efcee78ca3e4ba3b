"""[needs tools/xcd_balance.patch applied to the library: the experiment of profiles/r06_xcd_balance.txt was not kept]
Diagnostic (library built with -DPANDA_XCD_FINISH, PANDA_LIB=tools/bin/libpanda-xcdfinish.so, PANDA_XCD_FINISH_PRINT=1): per call, when the last
workgroup of every XCD left k_accumulate, with equal chunks and with chunk lengths that follow the XCDs' clocks.  usage: xcd_finish.py [log_n=24]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpu_util import NULL_STREAM, DeviceBuffer  # noqa: E402
from panda_amd import gpu_ffi as ffi  # noqa: E402
from panda_amd import gpu_manager as pgm  # noqa: E402

if os.environ.get("PANDA_LIB"):
    ffi.LIB_PATH = os.environ["PANDA_LIB"]
k = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n = 1 << k
lib = ffi.load()
gm = pgm.PandaGpuManager(0)
db, ds, dr = DeviceBuffer(n * 64), DeviceBuffer(n * 32), DeviceBuffer(96)
ffi.check(lib.panda_gen_bases(0, 1, 0, n, db.ptr, NULL_STREAM), "gen")
ffi.check(lib.panda_gen_scalars(0, 2, 0, n, ds.ptr, NULL_STREAM), "gen")
cfg = ffi.MSMConfiguration(gm.mem_pool, gm.exec_stream.raw, db.ptr, ds.ptr, dr.ptr, k, 0)
ffi.check(lib.panda_msm_precompute_bases(0, db.ptr, k, 0, gm.exec_stream.raw), "pre")
lib.panda_set_clock_stamps(1)
lib.panda_msm_set_phase_timing(1)
ph, clk = (C.c_float * 8)(), (C.c_uint64 * 12)()
for mode in (0, 1, 0, 1):
    lib.panda_msm_set_xcd_balance(mode)
    print(f"--- xcd balance {mode}", file=sys.stderr, flush=True)
    for i in range(6):
        ffi.check(lib.panda_msm_execute_bn254(cfg), "msm")
        lib.panda_msm_last_phase_ms(ph)
        lib.panda_msm_last_clock(clk)
        print(f"    k_accumulate {ph[3]:.3f} ms   per-XCD MHz " + " ".join(f"{clk[4 + x] / max(clk[1], 1) * 100:.0f}" for x in range(8)), file=sys.stderr, flush=True)
