"""Randomised soak of the NTT entry points against the CPU oracle (development aid, not part of the suites).
usage: fuzz_ntt.py <cases> [seed]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import oracle as po  # noqa: E402
from panda_amd import gpu_manager as pgm  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    gm = pgm.PandaGpuManager(0)
    calls = {po.F_BN254_FR: lambda b, om, k, inv: (pgm.panda_intt_bn254_gpu if inv else pgm.panda_ntt_bn254_gpu_v1)(gm, b, om, k),
             po.F_BLS377_FR: lambda b, om, k, inv: pgm.panda_ntt_bls12_377_gpu_v1(gm, b, om, k, inverse=inv),
             po.F_BLS381_FR: lambda b, om, k, inv: pgm.panda_ntt_bls12_381_gpu_v1(gm, b, om, k, inverse=inv)}
    bad = 0
    t0 = time.time()
    for it in range(cases):
        fid = [po.F_BN254_FR, po.F_BLS377_FR, po.F_BLS381_FR][int(rng.integers(0, 3))]
        k = int(rng.integers(0, 22))  # up to 2^21: the three-pass plans and their streamed inter-pass table included
        from panda_amd import gpu_ffi as _ffi
        streamed = int(rng.integers(0, 2))
        _ffi.load().panda_ntt_set_streamed_tables(streamed)  # flips at random: the two kinds of table share the thread's two cache entries
        om = po.root_of_unity(fid, k)
        x = po.gen_scalars(fid, int(rng.integers(1, 1 << 40)), 1 << k)
        if rng.random() < 0.2:
            x[rng.random(1 << k) < 0.7] = 0
        buf = x.copy()
        calls[fid](buf, om, k, False)
        ok = (buf == po.ntt(fid, x, om, k)).all()
        calls[fid](buf, om, k, True)
        ok = ok and (buf == x).all()
        if ok and fid == po.F_BN254_FR:  # the other BN254 orderings: bit-reversed output / input, and the sharded composition with its inverse
            want = po.ntt(fid, x, om, k)
            perm = np.array([int(format(i, f"0{k}b")[::-1], 2) if k else 0 for i in range(1 << k)])
            buf = x.copy()
            pgm.panda_ntt_bn254_gpu_bitrev(gm, buf, om, k)
            ok = (buf[perm] == want).all()
            pgm.panda_ntt_bn254_gpu_bitrev(gm, buf, om, k, inverse=True)
            ok = ok and (buf == x).all()
            g = int(rng.integers(1, 4))
            if ok and k >= 2 * g:
                import torch
                from panda_amd import multi_gpu
                G, m = 1 << g, (1 << k) >> g
                dev = torch.device("cuda", 0)
                slabs = [torch.from_numpy(multi_gpu.slab_of(x, G, r).view(np.uint8).reshape(-1).copy()).to(dev) for r in range(G)]
                outs = multi_gpu.ntt_sharded_one_process(slabs, [torch.empty_like(t) for t in slabs], om, k)
                y = multi_gpu.natural_from_slab_outputs([o.cpu().numpy().view(np.uint32).reshape(m, 8) for o in outs])
                ok = (y == want).all()
                back = multi_gpu.intt_sharded_one_process([o.clone() for o in outs], [torch.empty_like(o) for o in outs], om, k)
                ok = ok and all((back[r].cpu().numpy().view(np.uint32).reshape(m, 8) == multi_gpu.slab_of(x, G, r)).all() for r in range(G))
        if not ok:
            bad += 1
            print("MISMATCH", dict(fid=fid, k=k, streamed=streamed), flush=True)
        if it % 25 == 24:
            print(f"{it + 1} cases, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
    print(f"done: {cases} cases, {bad} mismatches")
    from panda_amd import gpu_ffi as _ffi
    _ffi.load().panda_ntt_set_streamed_tables(0xFFFFFFFF)
    gm.deinit()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
