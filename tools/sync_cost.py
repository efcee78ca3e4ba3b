import ctypes as C, os, sys, time
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
flag = int(sys.argv[1])
import torch
hip = C.CDLL("libamdhip64.so")
if flag >= 0:
    print("hipSetDeviceFlags", flag, "->", hip.hipSetDeviceFlags(flag))
from gpu_util import NULL_STREAM, DeviceBuffer
from panda_amd import gpu_ffi as ffi
from panda_amd import gpu_manager as pgm
lib = ffi.load(); gm = pgm.PandaGpuManager(0)
def med(v): v = sorted(v); return v[len(v)//2]
for k in (20, 24):
    n = 1 << k
    db, ds, dr = DeviceBuffer(n*64), DeviceBuffer(n*32), DeviceBuffer(96)
    ffi.check(lib.panda_gen_bases(0,1,0,n,db.ptr,NULL_STREAM),"g"); ffi.check(lib.panda_gen_scalars(0,2,0,n,ds.ptr,NULL_STREAM),"g")
    cfg = ffi.MSMConfiguration(gm.mem_pool, gm.exec_stream.raw, db.ptr, ds.ptr, dr.ptr, k, 0)
    ffi.check(lib.panda_msm_precompute_bases(0, db.ptr, k, 0, gm.exec_stream.raw), "pre")
    ph = (C.c_float*8)()
    lib.panda_msm_set_phase_timing(1)
    ts, dv = [], []
    for i in range(24):
        t = time.perf_counter(); ffi.check(lib.panda_msm_execute_bn254(cfg), "m"); dt = time.perf_counter()-t
        lib.panda_msm_last_phase_ms(ph)
        if i >= 4: ts.append(dt*1e3); dv.append(ph[7])
    print(f"flags {flag} 2^{k}: wall {med(ts):.4f} ms  device total {med(dv):.4f} ms  gap {med(ts)-med(dv):.4f} ms", flush=True)
    lib.panda_msm_unregister_bases(db.ptr)
    for d in (db, ds, dr): d.free()
