"""Randomised soak of the MSM entry points against the linearity identity (development aid, not part of the suites).
usage: fuzz_msm.py <cases> [seed]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import oracle as po  # noqa: E402
import pyref  # noqa: E402
from panda_amd import gpu_ffi as ffi  # noqa: E402
from panda_amd import gpu_manager as pgm  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    gm = pgm.PandaGpuManager(0)
    lib = ffi.load()
    t0 = time.time()
    bad = 0
    for it in range(cases):
        cid = int(rng.integers(0, 3))
        c = pyref.CURVES[cid]
        k = int(rng.integers(0, 17))
        n = 1 << k
        seed = int(rng.integers(1, 1 << 40))
        mont = lambda v: pyref.int_to_limbs(v * c.Rr % c.r, 8)
        scalars = po.gen_scalars(po.FR_OF[cid], seed + 1, n)
        pattern = int(rng.integers(0, 6))
        if pattern == 1:
            scalars[rng.random(n) < 0.8] = 0
        elif pattern == 2:
            scalars = np.stack([mont(int(v)) for v in rng.integers(0, 1 << int(rng.integers(1, 60)), n)])
        elif pattern == 3:
            vals = [mont(int.from_bytes(rng.bytes(32), "little") % c.r) for _ in range(int(rng.integers(1, 4)))]
            scalars = np.stack([vals[int(i)] for i in rng.integers(0, len(vals), n)])
        elif pattern == 4:
            scalars = np.stack([mont((c.r - 1 - int(v)) % c.r if i % 2 else int(v)) for i, v in enumerate(rng.integers(0, 1 << 20, n))])
        elif pattern == 5:
            scalars[:] = mont(c.r - 1)
        bases = po.gen_bases(cid, seed, n)
        want = po.expected_from_linearity(cid, seed, scalars)
        mode = int(rng.integers(0, 3))
        # the round-5 options ride along at random: the split sort (any cut, work-queue or plain grid) and the accumulate with its row in LDS
        ov = (int(rng.integers(1, 128)), int(rng.choice([0, 2, 6, 8, 64]))) if rng.random() < 0.5 else (0, 0)
        variant = int(rng.integers(0, 4))  # 3: rows fetched four lanes to a row (every curve)
        lib.panda_msm_set_overlap(*ov)
        lib.panda_msm_set_accumulate_variant(variant)
        lib.panda_msm_set_wide_merge(int(rng.integers(0, 4)))  # level-3 merge: policy / every cell through the wide variant / none / every cell through the 256-thread one
        if mode == 1:  # registered bases, plain windows: forced widths incl. the three-level sort with a list per window
            lib.panda_msm_set_window_bits(int(rng.choice([0, 12, 16, 17, 19, 20])))
        wb = int(rng.choice([0, 0, 8, 10, 12, 14, 16, 18, 20, 22]))
        if mode == 0:
            lib.panda_msm_set_window_bits(int(rng.choice([0, 0, 4, 5, 7, 9, 11, 13, 16, 17, 18, 19, 20])))
            out = pgm.panda_msm_bn254_gpu(gm, scalars, bases, curve=cid)
            lib.panda_msm_set_window_bits(0)
        else:
            idx = gm.add_cached_bases(bases)
            try:
                if mode == 1:
                    gm.register_cached_bases(idx, curve=cid)
                else:
                    gm.precompute_cached_bases(idx, curve=cid, window_bits=wb)
            except ffi.PandaGpuError:
                continue
            out = pgm.panda_msm_bn254_gpu_with_cached_bases(gm, scalars, idx, curve=cid)
            lib.panda_msm_unregister_bases(gm.d_bases[idx])
            lib.panda_msm_set_window_bits(0)
        got = po.to_affine(cid, out.view(np.uint32))
        if not (got == want).all():
            bad += 1
            print("MISMATCH", dict(cid=cid, k=k, seed=seed, pattern=pattern, mode=mode, wb=wb, overlap=ov, variant=variant), flush=True)
        if it % 25 == 24:
            print(f"{it + 1} cases, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
    lib.panda_msm_set_overlap(0xFFFFFFFF, 0)
    lib.panda_msm_set_accumulate_variant(0)
    lib.panda_msm_set_wide_merge(0)
    print(f"done: {cases} cases, {bad} mismatches")
    gm.deinit()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
