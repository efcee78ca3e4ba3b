#!/usr/bin/env python3
"""Generates tests/golden/msm_cases.json and ntt_cases.json from the CPU oracle (after it has been pinned to the
reference's k13 vector and to tests/pyref.py by tests/test_oracle.py).  Inputs are regenerable from seeds, so only
seeds and the 64/96-byte answers (hex) or sha256 digests are stored.

    python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle as po  # noqa: E402
import pyref  # noqa: E402


def scalar_set(kind, cid, n, seed):
    c = pyref.CURVES[cid]
    fr = po.FR_OF[cid]
    mont = lambda v: pyref.int_to_limbs(v * c.Rr % c.r, 8)
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        return po.gen_scalars(fr, seed, n)
    if kind == "zeros":
        return np.zeros((n, 8), np.uint32)
    if kind == "ones":
        return np.tile(mont(1), (n, 1))
    if kind == "minus_one":
        return np.tile(mont(c.r - 1), (n, 1))
    if kind == "small16":
        return np.stack([mont(int(v)) for v in rng.integers(0, 1 << 16, n)])
    if kind == "all_equal":
        return np.tile(po.gen_scalars(fr, seed, 1), (n, 1))
    if kind == "half_zero":
        s = po.gen_scalars(fr, seed, n)
        s[::2] = 0
        return s
    raise ValueError(kind)


def main():
    msm = []
    for cid, k, kind, seed in [(0, 10, "uniform", 1), (0, 12, "uniform", 2), (0, 14, "uniform", 3), (1, 10, "uniform", 4), (1, 12, "uniform", 5),
                               (0, 11, "zeros", 6), (0, 11, "ones", 7), (0, 11, "minus_one", 8), (0, 11, "small16", 9), (0, 11, "all_equal", 10),
                               (0, 11, "half_zero", 11), (1, 11, "minus_one", 12), (1, 11, "all_equal", 13),
                               (2, 10, "uniform", 14), (2, 12, "uniform", 15), (2, 11, "minus_one", 16), (2, 11, "small16", 17)]:
        n = 1 << k
        bases = po.gen_bases(cid, 0xB000 + seed, n)
        scalars = scalar_set(kind, cid, n, 0x5000 + seed)
        aff = po.msm_affine(cid, bases, scalars, window_bits=10)
        jac = po.msm(cid, bases, scalars, window_bits=10)
        msm.append({"curve": cid, "log_n": k, "scalars": kind, "bases_seed": 0xB000 + seed, "scalars_seed": 0x5000 + seed,
                    "affine_hex": aff.tobytes().hex(), "identity": bool(not jac[2 * po.LC_Q[cid]:].any())})
    # degenerate bases: all equal (the structure of the reference's own k13 fixture), at another size
    n = 1 << 12
    bases = np.tile(po.generator(0), (n, 1))
    scalars = po.gen_scalars(po.F_BN254_FR, 0x5100, n)
    msm.append({"curve": 0, "log_n": 12, "scalars": "uniform", "bases": "all_generator", "scalars_seed": 0x5100,
                "affine_hex": po.msm_affine(0, bases, scalars, window_bits=10).tobytes().hex(), "identity": False})
    json.dump(msm, open(os.path.join(HERE, "msm_cases.json"), "w"), indent=1)

    ntt = []
    for log_n in (1, 4, 8, 9, 13, 16, 18):
        fid = po.F_BN254_FR
        om = po.root_of_unity(fid, log_n)
        x = po.gen_scalars(fid, 0x7000 + log_n, 1 << log_n)
        y = po.ntt(fid, x, om, log_n)
        ntt.append({"log_n": log_n, "seed": 0x7000 + log_n, "omega_hex": om.tobytes().hex(), "sha256": hashlib.sha256(y.tobytes()).hexdigest(),
                    "first_hex": y[0].tobytes().hex(), "last_hex": y[-1].tobytes().hex()})
    json.dump(ntt, open(os.path.join(HERE, "ntt_cases.json"), "w"), indent=1)
    print(f"wrote {len(msm)} MSM cases and {len(ntt)} NTT cases")


if __name__ == "__main__":
    main()
