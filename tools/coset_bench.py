import ctypes as C, os, sys, time
ROOT="/root/repo"; sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,"tests"))
from gpu_util import NULL_STREAM, DeviceBuffer
from panda_amd import gpu_ffi as ffi, gpu_manager as pgm
import oracle as po
lib=ffi.load(); gm=pgm.PandaGpuManager(0)
k=24; n=1<<k
om=po.root_of_unity(po.F_BN254_FR,k); g=po.gen_scalars(po.F_BN254_FR, 5, 1)[0].copy()
a,b=DeviceBuffer(n*32),DeviceBuffer(n*32)
ffi.check(lib.panda_gen_scalars(0,3,0,n,a.ptr,NULL_STREAM),"gen")
flag=C.c_uint(0)
cfg=ffi.NttconfigurationV1(gm.mem_pool, gm.exec_stream.raw, a.ptr, b.ptr, C.c_void_p(om.ctypes.data), k, C.pointer(flag))
for name,fn,extra in (("plain",lib.panda_ntt_execute_bn254_v1,()),("coset",lib.panda_ntt_execute_bn254_coset,(C.c_void_p(g.ctypes.data),)),("coset_inv",lib.panda_ntt_execute_bn254_coset_inverse,(C.c_void_p(g.ctypes.data),))):
    ts=[]
    for r in range(9):
        t=time.perf_counter(); ffi.check(fn(cfg,*extra),"ntt"); ts.append(time.perf_counter()-t)
    ts=sorted(ts[2:]); print(name, "median %.3f ms best %.3f"%(ts[len(ts)//2]*1e3, ts[0]*1e3))
