"""Large-size parity checks by the linearity identity for every curve, plain and with tables (development aid)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import oracle as po
from panda_amd import gpu_manager as pgm
import test_gpu_parity as T
gm = pgm.PandaGpuManager(0)
for cid, k in ((2, 24), (1, 24), (0, 25), (0, 23)):
    seed_b = 0x70616E6461 ^ (40 + k + cid)
    t = time.time()
    out, scalars, tables, bits = T._msm_precomputed_on_device(gm, cid, k, 0, seed_b, 0x5CA1B0 + k + cid)
    ok = (po.to_affine(cid, out) == po.expected_from_linearity(cid, seed_b, scalars)).all()
    print(f"curve {cid} 2^{k} tables={tables} bits={bits} parity={ok} ({time.time()-t:.1f}s)", flush=True)
    out, scalars = T._msm_on_device_inputs(gm, cid, k, seed_b, 0x5CA1B0 + k + cid)
    ok = (po.to_affine(cid, out) == po.expected_from_linearity(cid, seed_b, scalars)).all()
    print(f"curve {cid} 2^{k} plain parity={ok}", flush=True)
gm.deinit()
