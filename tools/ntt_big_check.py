"""BN254 NTT at the largest supported sizes (2^27, 2^28: a fourth pass of degree 3 / 4): forward + inverse round trip compared on the
device, bit-reversed orderings, and one forward value by O(n) evaluation with the oracle.  Development aid (8 GiB per buffer at 2^28).
usage: ntt_big_check.py <log_n>"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import oracle as po  # noqa: E402
from panda_amd import gpu_ffi as ffi  # noqa: E402

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 27
n = 1 << log_n
lib = ffi.load()
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev)
ps = ffi.PandaStream(st.cuda_stream)
a = torch.empty(n * 32, dtype=torch.uint8, device=dev)
b = torch.empty_like(a)
ffi.check(lib.panda_gen_scalars(0, 0xB16 + log_n, 0, n, a.data_ptr(), ps), "gen")
keep = a.clone()
om = po.root_of_unity(po.F_BN254_FR, log_n)
flag = C.c_uint(9)


def run(fn, src, dst):
    cfg = ffi.NttconfigurationV1(ffi.PandaMemPool(), ps, src.data_ptr(), dst.data_ptr(), C.c_void_p(om.ctypes.data), log_n, C.pointer(flag))
    t = time.time()
    ffi.check(fn(cfg), "ntt")
    dt = time.time() - t
    return ((dst, src) if flag.value else (src, dst)) + (dt,)


fwd, other, t_f = run(lib.panda_ntt_execute_bn254_v1, a, b)
k = 0x1234567 & (n - 1)
y_k = fwd[k * 32:(k + 1) * 32].cpu().numpy().view(np.uint32)
x = keep.cpu().numpy().view(np.uint32).reshape(n, 8)
t = time.time()
want = po.ntt_eval_at(po.F_BN254_FR, x, om, log_n, k)
print(f"2^{log_n}: forward {t_f * 1e3:.2f} ms; y[{k}] by O(n) evaluation ({time.time() - t:.0f} s): {'ok' if (y_k == want).all() else 'MISMATCH'}", flush=True)
res, other, t_i = run(lib.panda_ntt_execute_bn254_inverse, fwd, other)
print(f"inverse {t_i * 1e3:.2f} ms; inverse(forward(x)) == x over the whole buffer: {bool(torch.equal(res, keep))}", flush=True)
br, other, _ = run(lib.panda_ntt_execute_bn254_bitrev_out, res, other)
pos = int(format(k, f"0{log_n}b")[::-1], 2)
print("bit-reversed output holds y[k] at bitrev(k):", bool((br[pos * 32:(pos + 1) * 32].cpu().numpy().view(np.uint32) == want).all()), flush=True)
back, other, _ = run(lib.panda_ntt_execute_bn254_inverse_bitrev_in, br, other)
print("inverse from bit-reversed input == x:", bool(torch.equal(back, keep)), flush=True)
