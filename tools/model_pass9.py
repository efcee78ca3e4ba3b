"""Index-level model of the radix-512 pass (csrc/ntt_radix9.h): the thread -> element maps, the two LDS exchanges with their XOR
swizzles and the output addressing, run over a small prime field with numpy, against (a) a direct DFT of every sub-transform with the
pass's input / output / twiddle formulas and (b) the whole transform as a product of passes.  Also counts LDS bank conflicts of every
exchange access (32 banks, two 32-lane halves: ds_read_b32 / ds_write_b32).  Development aid; tests/test_ntt_model.py runs it on the CPU."""
import numpy as np

P = 2013265921  # 15 * 2^27 + 1
G = 31


def root(log_n):
    return pow(G, (P - 1) >> log_n, P)


def brev(v, bits):
    r = 0
    for b in range(bits):
        r |= ((v >> b) & 1) << (bits - 1 - b)
    return r


def dft_direct(x, w):
    n = len(x)
    idx = np.arange(n)
    pw = np.array([pow(w, int(t), P) for t in range(n)], dtype=object)
    out = np.zeros(n, dtype=object)
    for k in range(n):
        out[k] = int(sum(int(x[j]) * int(pw[(j * k) % n]) for j in idx) % P)
    return out


def ntt_ref(x, log_n):
    """plain iterative radix-2 transform, natural order in and out"""
    n = 1 << log_n
    w = root(log_n)
    a = [int(v) for v in x]
    # bit-reverse, then DIT
    a = [a[brev(i, log_n)] for i in range(n)] if log_n else a
    ln = 1
    while ln < n:
        wl = pow(w, n // (2 * ln), P)
        for s in range(0, n, 2 * ln):
            t = 1
            for j in range(ln):
                u, v = a[s + j], a[s + j + ln] * t % P
                a[s + j], a[s + j + ln] = (u + v) % P, (u - v) % P
                t = t * wl % P
        ln *= 2
    return a


def pass_functional(x, log_n, d, lgp, d2):
    """One pass by its definition: sub-transform blk reads x[blk + i S], S = n >> d, writes output i_out to
    (blk_hi << (lgp + d)) | (i_out << lgp) | k with blk = (blk_hi << lgp) | k, times W^(i2 k2) when d2 != 0 (not the last pass):
    W = w^(n >> (lgp + d + d2)), k2 = (i_out << lgp) | k, i2 = top d2 bits of blk_hi."""
    n = 1 << log_n
    w = root(log_n)
    S = n >> d
    wd = pow(w, n >> d, P)
    y = [0] * n
    W = pow(w, n >> (lgp + d + d2), P) if d2 else 1
    hi_bits = log_n - d - lgp
    # vectorised DFT of all sub-transforms through numpy object arrays would be slow; use the radix-2 reference per sub-transform
    tw = [pow(wd, t, P) for t in range(1 << d)]
    for blk in range(S):
        sub = [x[blk + i * S] for i in range(1 << d)]
        out = ntt_small(sub, d, tw)
        k = blk & ((1 << lgp) - 1)
        blk_hi = blk >> lgp
        i2 = (blk_hi >> (hi_bits - d2)) if d2 else 0
        for i_out in range(1 << d):
            k2 = (i_out << lgp) | k
            v = out[i_out]
            if d2:
                v = v * pow(W, i2 * k2, P) % P
            y[(blk_hi << (lgp + d)) | (i_out << lgp) | k] = v
    return y


def ntt_small(a, d, tw):
    n = 1 << d
    a = [a[brev(i, d)] for i in range(n)] if d else list(a)
    ln = 1
    while ln < n:
        step = n // (2 * ln)
        for s in range(0, n, 2 * ln):
            for j in range(ln):
                u, v = a[s + j], a[s + j + ln] * tw[j * step] % P
                a[s + j], a[s + j + ln] = (u + v) % P, (u - v) % P
        ln *= 2
    return a


def check_plan(log_n, radices, seed=1):
    rng = np.random.default_rng(seed)
    x = [int(v) for v in rng.integers(0, P, 1 << log_n)]
    want = ntt_ref(x, log_n)
    cur, lgp = x, 0
    for j, d in enumerate(radices):
        d2 = radices[j + 1] if j + 1 < len(radices) else 0
        cur = pass_functional(cur, log_n, d, lgp, d2)
        lgp += d
    assert cur == want, (log_n, radices)


# ---------------------------------------------------------------------------------------------------------------- thread-level model
br3 = lambda m: ((m & 1) << 2) | (m & 2) | ((m >> 2) & 1)


class Banks:
    def __init__(self):
        self.worst = 1

    def access(self, addrs_by_lane):
        """addrs_by_lane: 64 word addresses of one ds_read_b32 / ds_write_b32; returns the worst multiplicity over the two halves"""
        for half in (addrs_by_lane[:32], addrs_by_lane[32:]):
            per_bank = {}
            for a in half:
                per_bank.setdefault(a % 32, set()).add(a)
            self.worst = max(self.worst, max(len(v) for v in per_bank.values()))


def pass9_threads(x, log_n, lgp, d2, last, banks=None, tiles=None):
    """The pass as k_ntt_pass9 runs it: 256 threads per tile of four 512-point sub-transforms, eight elements per thread, three register
    blocks of three DIF rounds, two exchanges.  Returns y (only the tiles asked for are written)."""
    n = 1 << log_n
    w = root(log_n)
    S = n >> 9
    w512 = pow(w, n >> 9, P)
    pq = [pow(w512, t, P) for t in range(256)]
    hi_bits = log_n - 9 - lgp
    W = pow(w, n >> (lgp + 9 + d2), P) if d2 else 1
    y = [None] * n
    for tile in (tiles if tiles is not None else range(n // 2048)):
        blk0 = tile * 4
        lds = [None] * 2048
        # ---- block A: thread (s, i0) holds i = i0 + 64 m
        regs = {}
        for tid in range(256):
            lane, wave = tid & 63, tid >> 6
            s, i0 = lane & 3, (lane >> 2) | (wave << 4)
            e = [x[blk0 + s + (i0 + 64 * m) * S] for m in range(8)]
            for m in range(4):  # round 0, distance 256
                t = pq[i0 + 64 * m]
                a, b = e[m], e[m + 4]
                e[m], e[m + 4] = (a + b) % P, (a - b) * t % P
            for h in range(2):  # round 1, distance 128
                t = pq[2 * (i0 + 64 * h)]
                for o in (0, 4):
                    a, b = e[h + o], e[h + o + 2]
                    e[h + o], e[h + o + 2] = (a + b) % P, (a - b) * t % P
            t = pq[4 * i0]  # round 2, distance 64
            for m in range(0, 8, 2):
                a, b = e[m], e[m + 1]
                e[m], e[m + 1] = (a + b) % P, (a - b) * t % P
            regs[tid] = e
        # ---- exchange 1: element (s, i) at word s | ((i[2:0] ^ i[8:6]) << 2) | (i[5:0] << 5)
        for m in range(8):
            for wave in range(4):
                addrs = []
                for lane in range(64):
                    tid = lane | (wave << 6)
                    s, i0 = lane & 3, (lane >> 2) | (wave << 4)
                    wa = (s | ((i0 & 7) << 2) | (i0 << 5)) ^ (m << 2)
                    addrs.append(wa)
                    assert lds[wa] is None
                    lds[wa] = regs[tid][m]
                if banks:
                    banks.access(addrs)
        assert all(v is not None for v in lds)
        regs2 = {}
        for m in range(8):
            for wave in range(4):
                addrs = []
                for lane in range(64):
                    tid = lane | (wave << 6)
                    s, g, j = lane & 3, (lane >> 2) & 7, (lane >> 5) | (wave << 1)
                    ra = s | ((j ^ g) << 2) | (j << 5)
                    addrs.append(ra + m * 256)
                    regs2.setdefault(tid, [None] * 8)[m] = lds[ra + m * 256]
                if banks:
                    banks.access(addrs)
        # cross-check the exchange against the element identity: block B thread (s, g, j) register m must hold i = 64 g + 8 m + j
        # (verified through the arithmetic below: a wrong element gives a wrong transform)
        # ---- block B: rounds 3..5 (distances 32, 16, 8)
        for tid in range(256):
            lane, wave = tid & 63, tid >> 6
            j = (lane >> 5) | (wave << 1)
            e = regs2[tid]
            for m in range(4):
                t = pq[8 * (8 * m + j)]
                a, b = e[m], e[m + 4]
                e[m], e[m + 4] = (a + b) % P, (a - b) * t % P
            for h in range(2):
                t = pq[16 * (8 * h + j)]
                for o in (0, 4):
                    a, b = e[h + o], e[h + o + 2]
                    e[h + o], e[h + o + 2] = (a + b) % P, (a - b) * t % P
            t = pq[32 * j]
            for m in range(0, 8, 2):
                a, b = e[m], e[m + 1]
                e[m], e[m + 1] = (a + b) % P, (a - b) * t % P
        # ---- exchange 2: element (s, i) at word (i[8:6] ^ i[5:3]) | ((s ^ i[5:4]) << 3) | (i[5:0] << 5)
        lds = [None] * 2048
        for m in range(8):
            for wave in range(4):
                addrs = []
                for lane in range(64):
                    tid = lane | (wave << 6)
                    s, g, j = lane & 3, (lane >> 2) & 7, (lane >> 5) | (wave << 1)
                    wa = (g ^ m) | ((s ^ (m >> 1)) << 3) | (j << 5) | (m << 8)
                    addrs.append(wa)
                    assert lds[wa] is None
                    lds[wa] = regs2[tid][m]
                if banks:
                    banks.access(addrs)
        assert all(v is not None for v in lds)
        regs3 = {}
        thread_sq = {}
        for tid in range(256):
            lane, wave = tid & 63, tid >> 6
            if lgp == 0 and not last:  # a sub-transform's outputs are contiguous in i_out: lane = bitrev6(q), wave = s
                s, q = wave, brev(lane, 6)
            else:  # consecutive sub-transforms are contiguous
                s, q = lane & 3, (lane >> 2) | (wave << 4)
            thread_sq[tid] = (s, q)
        for m in range(8):
            for wave in range(4):
                addrs = []
                for lane in range(64):
                    tid = lane | (wave << 6)
                    s, q = thread_sq[tid]
                    ra = ((q >> 3) ^ (q & 7)) | ((s ^ ((q >> 1) & 3)) << 3) | ((q & 7) << 8)
                    addrs.append(ra + m * 32)
                    regs3.setdefault(tid, [None] * 8)[m] = lds[ra + m * 32]
                if banks:
                    banks.access(addrs)
        # ---- block C: rounds 6..8 (distances 4, 2, 1), then the outputs
        for tid in range(256):
            s, q = thread_sq[tid]
            e = regs3[tid]
            for m in range(4):
                t = pq[64 * m]
                a, b = e[m], e[m + 4]
                e[m], e[m + 4] = (a + b) % P, (a - b) * t % P
            for m in (0, 1, 4, 5):
                t = pq[128 * (m & 1)]
                a, b = e[m], e[m + 2]
                e[m], e[m + 2] = (a + b) % P, (a - b) * t % P
            for m in range(0, 8, 2):
                a, b = e[m], e[m + 1]
                e[m], e[m + 1] = (a + b) % P, (a - b) % P
            blk = blk0 + s
            k = blk & ((1 << lgp) - 1)
            blk_hi = blk >> lgp
            iq = brev(q, 6)
            i2 = (blk_hi >> (hi_bits - d2)) if d2 else 0
            for m in range(8):
                i_out = (br3(m) << 6) | iq
                v = e[m]
                if d2:
                    v = v * pow(W, i2 * ((i_out << lgp) | k), P) % P
                y[(blk_hi << (lgp + 9)) | (i_out << lgp) | k] = v
    return y


def check_pass9(log_n, lgp, d2, last, tiles, seed=2):
    rng = np.random.default_rng(seed)
    n = 1 << log_n
    x = [int(v) for v in rng.integers(0, P, n)]
    banks = Banks()
    got = pass9_threads(x, log_n, lgp, d2, last, banks, tiles)
    # functional pass restricted to the same tiles
    w = root(log_n)
    S = n >> 9
    wd = pow(w, n >> 9, P)
    tw = [pow(wd, t, P) for t in range(512)]
    hi_bits = log_n - 9 - lgp
    W = pow(w, n >> (lgp + 9 + d2), P) if d2 else 1
    for tile in tiles:
        for s in range(4):
            blk = tile * 4 + s
            out = ntt_small([x[blk + i * S] for i in range(512)], 9, tw)
            k, blk_hi = blk & ((1 << lgp) - 1), blk >> lgp
            i2 = (blk_hi >> (hi_bits - d2)) if d2 else 0
            for i_out in range(512):
                v = out[i_out]
                if d2:
                    v = v * pow(W, i2 * ((i_out << lgp) | k), P) % P
                pos = (blk_hi << (lgp + 9)) | (i_out << lgp) | k
                assert got[pos] == v, (log_n, lgp, d2, tile, s, i_out)
    return banks.worst


if __name__ == "__main__":
    for log_n, radices in ((6, (2, 2, 2)), (7, (3, 2, 2)), (8, (3, 3, 2)), (9, (3, 3, 3)), (10, (4, 3, 3)), (9, (4, 5)), (11, (4, 4, 3))):
        check_plan(log_n, radices)
    print("pass formulas: ok")
    for log_n, lgp, d2, last, tiles in ((18, 0, 9, False, (0, 5, 127)), (18, 9, 0, True, (0, 77)), (20, 9, 2, False, (3, 300)),
                                        (19, 0, 8, False, (1, 200)), (12, 0, 3, False, (0, 1)), (17, 8, 0, True, (0, 63)), (20, 8, 3, False, (5, 511))):
        worst = check_pass9(log_n, lgp, d2, last, tiles)
        print(f"pass9 threads log_n={log_n} lgp={lgp} d2={d2} last={last}: ok, worst LDS bank multiplicity {worst}")
