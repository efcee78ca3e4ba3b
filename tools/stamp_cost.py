import ctypes as C, os, sys, time
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
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
    res = {}
    for r in range(6):
        for mode in ((0,0),(1,0),(1,1)) if r % 2 == 0 else ((1,1),(1,0),(0,0)):
            lib.panda_msm_set_phase_timing(mode[0]); lib.panda_set_clock_stamps(mode[1])
            ts, ka = [], []
            for i in range(9):
                t = time.perf_counter(); ffi.check(lib.panda_msm_execute_bn254(cfg), "m"); dt = time.perf_counter()-t
                lib.panda_msm_last_phase_ms(ph)
                if i >= 2: ts.append(dt*1e3); ka.append(ph[3])
            res.setdefault(mode, []).append((med(ts), med(ka)))
    lib.panda_msm_set_phase_timing(0); lib.panda_set_clock_stamps(0)
    for mode, v in res.items():
        print(f"2^{k} timing={mode[0]} stamps={mode[1]}: wall {med([x[0] for x in v]):.4f} ms  k_accumulate(events) {med([x[1] for x in v]):.4f} ms", flush=True)
    lib.panda_msm_unregister_bases(db.ptr)
    for d in (db, ds, dr): d.free()
