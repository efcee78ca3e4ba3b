#!/usr/bin/env python3
"""bench.py -- contract benchmark: BN254 MSM 2^24 points per GPU, inputs resident in HBM.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic input: panda_msm_execute_bn254 on this rank's
base range (2^24 points, cached bases, Jacobian output) and, for N > 1, the all-gather of the 96-byte partials over
RCCL plus the G - 1 point additions (weak scaling: per-GPU work is fixed).  Rank 0 prints ONE JSON line.

  value      whole-job MSM points/s = N * 2^24 * K / wall (max over ranks, barrier + synchronize on both sides)
  roofline   the dominant kernel (k_accumulate): algorithmic bytes per launch (96 B/point) / its
             average duration, measured with HIP events on the launch stream inside the timed region
  roofline_issue  the same kernel against the limit that actually binds it: v_mad_u64_u32 lane-operations per second
             over the measured peak of that instruction (profiles/r01_ubench_int_rates.txt)
  cpu_baseline  the CPU oracle (port of the reference's host-debug Pippenger, c = 16, one thread) timed on a bounded
             sample on this box's host cores; N = 1 only

Secondary figures, each measured in the same run but outside the timed region of `value` (every BASELINE.json config
gets a number the driver records):
  config4_msm_2_26   BN254 MSM 2^26 in total, base ranges of 2^26 / N points per rank, partials all-gathered and combined
                     (strong scaling: at N = 8 this is BASELINE config 4 as stated, 8 x 2^23)
  ntt                N = 1: BN254 NTT 2^24 forward and inverse (config 3)
  ntt_sharded        N > 1: the slab-sharded transform (step 1 -> RCCL all-to-all -> step 2), 2^24 in total (strong) and
                     2^24 per GPU (weak)
  config2_msm_2_20, config5_bls12_377_2_24_projective   N = 1
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s
BYTES_PER_POINT = {0: 96, 1: 128, 2: 128, 3: 160}  # 32 B scalar + affine base; 3 = BN254 G2
BYTES_PER_NTT_ELEM = 64
POINT_BYTES = {0: 64, 1: 96, 2: 96, 3: 128}
RESULT_BYTES = {0: 96, 1: 144, 2: 144, 3: 192}
# v_mad_u64_u32 lane-operations per second, whole chip, 8 waves per SIMD: profiles/r01_ubench_int_rates.txt ("mad_u64_u32 ... 28450.16 Gop/s")
MAD_PEAK_PER_S = 28.45e12
# ... and what one bare dependent chain in a single asm block sustains at four waves per SIMD (tools/ubench_nop.hip): the instruction's own rate,
# which no multiplication reaches -- every product column also needs a mask and a 64-bit shift on the same vector pipe
MAD_BARE_CHAIN_PER_S = 33.4e12
# multiply-adds of one XYZZ mixed addition on 9 x 29-bit limbs: 8 products of 162, 2 squarings of 126, one shared reduction
# (counted in the ISA of k_accumulate<Bn254Fq>)
MADS_PER_ADDITION_BN254 = 1467
# the same count for a 14-limb base field (BLS12-377 / BLS12-381): 8 * 2 * 14^2 + 2 * (14 * 15 / 2 + 14^2) - 14^2
MADS_PER_ADDITION = {0: 1467, 1: 3416, 2: 3542}  # BLS12-377: minus 9 reductions x 14 products by the zero top limb of p (fe29.h, fe29_reduce_col)
# Operation-count MODEL of the reference's CUDA MSM on this chip: per call
# n * W * 11 mulmods of Jacobian mixed additions (msm_cuda.cuh:373-409) + B * ~370 for weighting and reducing the B = W (2^c - 1)
# buckets (msm_cuda.cuh:411-449, 451-497) at its fixed c = 16, W = 16, priced at the measured rate of the reference's own kind of
# multiplication on gfx950: 8 x 32-bit CIOS Montgomery, 92.58 G mulmod/s at 8 waves per SIMD (profiles/r01_ubench_int_rates.txt,
# "montmul8x32").  A model, not a measurement: the reference cannot run here (no CUDA device) and publishes no number.
REF_MODEL_MULMOD_PER_S = 92.58e9


def reference_model_ms(log_n: int) -> float:
    n, w, c = 1 << log_n, 16, 16
    return (n * w * 11 + w * ((1 << c) - 1) * 370) / REF_MODEL_MULMOD_PER_S * 1e3
SEED = 0x70616E6461


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log-n", type=int, default=24, help="log2 of the points per GPU of the headline leg (contract: 24)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ntt", action="store_true")
    ap.add_argument("--no-ntt-sweep", action="store_true", help="NTT leg: 2^24 only (profiling runs: one size per kernel in the counters)")
    ap.add_argument("--no-config4", action="store_true", help="skip the 2^26-total strong-scaling leg")
    ap.add_argument("--no-extra-configs", action="store_true", help="skip the config 2 / config 5 legs (N = 1)")
    ap.add_argument("--cpu-sample-log-n", type=int, default=19)
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, the contract) or gloo (rehearsal of the N > 1 path on a 1-GPU box)")
    ap.add_argument("--all-on-device0", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--no-register", action="store_true", help="do not register the cached bases (plain drop-in call path)")
    ap.add_argument("--no-tables", action="store_true", help="register the cached bases without precomputed window tables")
    ap.add_argument("--no-compare", action="store_true", help="skip the extra without-tables and PCIe measurements (profiling runs)")
    ap.add_argument("--config4-total-log-n", type=int, default=26)
    ap.add_argument("--single-process", action="store_true",
                    help="ONE process drives all --gpus devices through the C entry points (panda_msm_execute_bn254_multi / panda_ntt_execute_bn254_multi: "
                         "a worker thread and an RCCL communicator per device inside the library) instead of one torch.distributed rank per GPU")
    ap.add_argument("--loopback", action="store_true", help="--single-process rehearsal on a one-GPU box: every rank on device 0, device copies instead of RCCL")
    ap.add_argument("--rccl-on-device0", action="store_true",
                    help="--single-process rehearsal on a one-GPU box: every rank on device 0 over the RCCL transport itself -- only under the test-only RCCL "
                         "interposer (LD_PRELOAD=tests/fake_rccl/libfake_rccl.so with PANDA_TEST_SHARED_DEVICE_RCCL=1 FAKE_RCCL_ALLOW_SHARED_DEVICE=1, tests/test_fake_rccl.py)")
    ap.add_argument("--no-c-abi-leg", action="store_true", help="N > 1: do not run the single-process C-ABI leg after the torch.distributed legs")
    ap.add_argument("--launcher-note", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--detail", default=None, help="also write the un-shortened record (every leg's full dictionary) to this file; "
                                                   "default: gpurun_out/bench_detail_n<N>.json when gpurun_out/ exists")
    return ap.parse_args()


def cpu_baseline(sample_log_n: int) -> dict:
    """The only place bench.py touches the oracle: as the thing timed beside the GPU, never as the thing shipped."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle as po

    n = 1 << sample_log_n
    bases = po.gen_bases(po.BN254, 0xC0FFEE, n)
    scalars = po.gen_scalars(po.F_BN254_FR, 0xC0FFEE + 1, n)
    t = time.time()
    po.msm(po.BN254, bases, scalars, window_bits=16, threads=1)
    dt = time.time() - t
    out = {"value": n / dt, "unit": "points/s", "cores": 1, "kind": "port",
           "sample": f"BN254 MSM 2^{sample_log_n} random bases/scalars, oracle/msm.c (reference host-debug algorithm, 16-bit windows), "
                     f"{dt:.1f} s on 1 of {os.cpu_count()} host threads"}
    # informative: the same algorithm with its 16 windows spread over 16 host threads (SURVEY 8d asks for both)
    t = time.time()
    po.msm(po.BN254, bases, scalars, window_bits=16, threads=16)
    dt16 = time.time() - t
    out["multi_thread"] = {"value": n / dt16, "unit": "points/s", "cores": 16, "seconds": round(dt16, 2)}
    return out


class Ctx:
    """What every leg needs: the library, this rank's device and stream, the process group."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist

        from panda_amd import gpu_ffi as ffi
        from panda_amd import multi_gpu

        self.torch, self.dist, self.ffi, self.multi_gpu, self.args = torch, dist, ffi, multi_gpu, args
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        local_rank = 0 if args.all_on_device0 else int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local_rank)
        self.dev = torch.device("cuda", local_rank)
        self.nccl = args.dist_backend == "nccl"
        if self.world > 1:
            if self.nccl:
                dist.init_process_group("nccl", device_id=self.dev)
            else:
                dist.init_process_group(args.dist_backend)
        self.lib = ffi.load()
        ffi.check(self.lib.panda_set_device(local_rank), "SetDeviceError")
        self.stream = torch.cuda.Stream(device=self.dev)
        self.pstream = ffi.PandaStream(self.stream.cuda_stream)

    def fence(self):
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()
        self.lib.panda_stream_sync(self.pstream)

    def max_over_ranks(self, dt: float) -> float:
        if self.world == 1:
            return dt
        t = self.torch.tensor([dt], dtype=self.torch.float64, device=self.dev if self.nccl else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def timed(self, step, warmup: int, steps: int) -> float:
        """The contract's protocol: W untimed steps, barrier + synchronize, K steps, barrier + synchronize, max over ranks."""
        for _ in range(warmup):
            step(False)
        self.fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(True)
        self.fence()
        return self.max_over_ranks(time.perf_counter() - t0)


def device_identity(ctx) -> str:
    """which device ran this line (arch, CUs, the tail of its UUID): with `sclk_mhz` and `k_accumulate_mcycles` beside the milliseconds, two
    records that differ can be read as another device, another clock or another kernel"""
    try:
        p = ctx.torch.cuda.get_device_properties(ctx.dev)
        uuid = str(getattr(p, "uuid", "") or "")
        return f"{getattr(p, 'gcnArchName', p.name).split(':')[0]}/{p.multi_processor_count}cu/{uuid[-12:] if uuid else 'no-uuid'}"
    except Exception as e:  # noqa: BLE001
        return f"unknown ({type(e).__name__})"


class MsmProblem:
    """One rank's share of an MSM: 2^log_n synthetic points and scalars generated in HBM (range [first, first + n) of the
    seeded stream), optionally registered / tabled, plus the exchange + combine step for N > 1."""

    def __init__(self, ctx: Ctx, curve: int, log_n: int, seed: int, coord: int, first: int = 0, register: bool = True, tables: bool = True):
        torch, ffi, lib = ctx.torch, ctx.ffi, ctx.lib
        self.ctx, self.curve, self.log_n, self.n = ctx, curve, log_n, 1 << log_n
        n = self.n
        self.bases = torch.empty(n * POINT_BYTES[curve], dtype=torch.uint8, device=ctx.dev)
        self.scalars = torch.empty(n * 32, dtype=torch.uint8, device=ctx.dev)
        self.result = torch.zeros(RESULT_BYTES[curve], dtype=torch.uint8, device=ctx.dev)
        ffi.check(lib.panda_gen_bases(curve, seed, first, n, self.bases.data_ptr(), ctx.pstream), "gen_bases")
        ffi.check(lib.panda_gen_scalars(curve, seed ^ 0xFFFF, first, n, self.scalars.data_ptr(), ctx.pstream), "gen_scalars")
        self.fn = (lib.panda_msm_execute_bn254, lib.panda_msm_execute_bls12_377, lib.panda_msm_execute_bls12_381, lib.panda_msm_execute_bn254_g2)[curve]
        self.cfg = ffi.MSMConfiguration(ffi.PandaMemPool(), ctx.pstream, self.bases.data_ptr(), self.scalars.data_ptr(), self.result.data_ptr(), log_n, coord)
        self.coord = coord
        self.tables, self.wbits, self.held, self.t_reg = 0, 0, 0, 0.0
        self.mode = "resident, plain pointer"
        self.registered = False
        if register:
            self.register(tables)
        self.total = None
        self.last_mcycles = self.last_sclk_mhz = None

    def register(self, tables: bool):
        """"cached bases" (BASELINE config): the base set stays on the device across MSMs and is registered once, so the
        library keeps its radix-converted copy -- and, with tables, the window tables 2^lo[k]*P built from it -- instead of
        re-deriving them in every call.  Built before any timed region; depends on the bases only."""
        ctx, lib, ffi = self.ctx, self.ctx.lib, self.ctx.ffi
        if self.registered:
            ffi.check(lib.panda_msm_unregister_bases(self.bases.data_ptr()), "unregister_bases")
        t = time.perf_counter()
        if tables and lib.panda_msm_precompute_bases(self.curve, self.bases.data_ptr(), self.log_n, 0, ctx.pstream) != 0:
            tables = False  # e.g. not enough free HBM for the tables: fall back to the converted copy alone
        if not tables:
            ffi.check(lib.panda_msm_register_bases(self.curve, self.bases.data_ptr(), self.log_n, ctx.pstream), "register_bases")
        self.t_reg = time.perf_counter() - t
        tb, wb, held = C.c_uint(0), C.c_uint(0), C.c_size_t(0)
        ffi.check(lib.panda_msm_registered_info(self.bases.data_ptr(), C.byref(tb), C.byref(wb), C.byref(held)), "registered_info")
        self.tables, self.wbits, self.held = tb.value, wb.value, held.value
        self.registered = True
        if self.tables > 1:
            self.mode = f"cached: {self.tables} window tables of {self.wbits} bits ({self.held / 2**30:.1f} GiB, built in {self.t_reg:.2f} s before the timed region)"
        else:
            self.mode = "cached: registered (panda_msm_register_bases), no tables"

    def execute(self):
        self.ctx.ffi.check(self.fn(self.cfg), "SchedulingErr")

    def exchange(self):
        """N > 1: all-gather of the result-sized partials (RCCL, device buffers) + the G - 1 point additions."""
        ctx = self.ctx
        if ctx.world == 1:
            return
        if ctx.nccl:
            gathered = ctx.torch.empty(ctx.world * self.result.numel(), dtype=ctx.torch.uint8, device=ctx.dev)
            ctx.dist.all_gather_into_tensor(gathered, self.result)
            partials = gathered.cpu().numpy().reshape(ctx.world, -1)
        else:
            partials = ctx.multi_gpu.allgather_partials(self.result.cpu().numpy())
        self.total = ctx.multi_gpu.combine_partials(partials, self.curve, self.coord)

    def phases(self):
        ph = (C.c_float * 8)()
        self.ctx.lib.panda_msm_last_phase_ms(ph)
        return list(ph)

    def all_phases(self, reps: int = 3):
        """Per-phase device times from `reps` extra, untimed calls with every phase timer on (the timed steps record the
        accumulate kernel and the call's total only: each further event costs a few microseconds of idle GPU)."""
        return self.timed_phases(2, reps)

    def timed_phases(self, level: int, reps: int = 3):
        """mean phase times of `reps` extra, untimed calls at phase-timer level `level` (1: the total and k_accumulate); the library's
        default -- what a timed step runs at unless it says otherwise -- records no device timers at all.  The same calls carry the clock
        stamps around k_accumulate: self.last_mcycles / self.last_sclk_mhz (means over the calls; None when nothing was stamped)."""
        lib = self.ctx.lib
        lib.panda_msm_set_phase_timing(level)
        lib.panda_set_clock_stamps(1)
        clk = (C.c_uint64 * self.ctx.ffi.CLOCK_WORDS)()
        try:
            rows, cyc = [], []
            for _ in range(reps):
                self.execute()
                rows.append(self.phases())
                lib.panda_msm_last_clock(clk)
                if clk[1] and clk[2]:
                    cyc.append((int(clk[0]) / 1e6, int(clk[3]) / int(clk[1]) * 100.0))
        finally:
            lib.panda_set_clock_stamps(0)
            lib.panda_msm_set_phase_timing(0)
        self.last_mcycles = sum(c[0] for c in cyc) / len(cyc) if cyc else None
        self.last_sclk_mhz = sum(c[1] for c in cyc) / len(cyc) if cyc else None
        return [sum(r[i] for r in rows) / len(rows) for i in range(8)]

    def release(self):
        if self.registered:
            self.ctx.lib.panda_msm_unregister_bases(self.bases.data_ptr())
            self.registered = False
        self.bases = self.scalars = self.result = None
        self.ctx.torch.cuda.empty_cache()


def single_process(args) -> dict:
    """--single-process: the sharded hot path as ONE C call per step, from one process -- what a Rust / C host gets.  Same workloads and
    the same timing protocol as the torch.distributed path (W untimed steps, K timed steps, synchronous calls); no torch involved."""
    import numpy as np  # noqa: F401

    from panda_amd import gpu_ffi as ffi
    from panda_amd import multi_gpu

    lib = ffi.load()
    lib.panda_msm_set_phase_timing(1)  # per-rank k_accumulate / device times are part of this mode's line
    G = args.gpus
    devices = [0] * G if (args.loopback or args.rccl_on_device0) else list(range(G))
    transport = ffi.MULTI_LOOPBACK if args.loopback else ffi.MULTI_RCCL
    mg = multi_gpu.MultiGpu(devices, transport)
    null = ffi.PandaStream()
    how = "device copies, all ranks on device 0 (rehearsal)" if args.loopback else ("RCCL call sites under the test interposer, all ranks on device 0 (rehearsal)" if args.rccl_on_device0 else "RCCL")

    def alloc(dev, nbytes):
        ffi.check(lib.panda_set_device(dev), "SetDeviceError")
        p = C.c_void_p()
        ffi.check(lib.panda_malloc(C.byref(p), nbytes), "CreateContextError")
        return p

    def free(dev, p):
        lib.panda_set_device(dev)
        lib.panda_free(p)

    def msm_leg(log_per, seed, warmup, steps, tables=True, from_host=None):
        per = 1 << log_per
        bufs, cfgs = [], []
        for d, dev in enumerate(devices):
            b, sc, r = alloc(dev, per * 64), alloc(dev, per * 32), alloc(dev, 96)
            ffi.check(lib.panda_gen_bases(0, seed, d * per, per, b, null), "gen")
            ffi.check(lib.panda_gen_scalars(0, seed ^ 0xFFFF, d * per, per, sc, null), "gen")
            if tables and lib.panda_msm_precompute_bases(0, b, log_per, 0, null) != 0:
                ffi.check(lib.panda_msm_register_bases(0, b, log_per, null), "register")
            bufs.append((dev, b, sc, r))
            cfgs.append(ffi.MSMConfiguration(ffi.PandaMemPool(), null, b, sc, r, log_per, ffi.JACOBIAN))
        for _ in range(warmup):
            mg.msm(cfgs)
        t0 = time.perf_counter()
        for _ in range(steps):
            mg.msm(cfgs)
        dt = time.perf_counter() - t0
        ph = [mg.phases(d) for d in range(G)]
        if from_host is not None:
            # the same shards with the scalars starting in pinned host memory: every worker uploads its own inside the call
            # (panda_msm_execute_bn254_from_host_multi); informative, never `value`
            hosts = []
            for dev, b, sc, r in bufs:
                hp = C.c_void_p()
                ffi.check(lib.panda_malloc_host(C.byref(hp), per * 32), "malloc_host")
                lib.panda_set_device(dev)
                ffi.check(lib.panda_memcpy(hp, sc, per * 32), "memcpy")
                hosts.append(hp)
            best = 1e9
            for _ in range(3):
                t1 = time.perf_counter()
                mg.msm_from_host(cfgs, [h.value for h in hosts], 5)
                best = min(best, time.perf_counter() - t1)
            # panda_msm_execute_bn254_from_host_multi: every rank's scalars cross PCIe from pinned host memory inside the call, in point
            # ranges beside its kernels; all ranks upload side by side
            from_host.update({"value": G * per / best, "unit": "points/s", "from_host_multi_ms": best * 1e3, "ranges": 5})
            for h in hosts:
                lib.panda_free_host(h)
        for dev, b, sc, r in bufs:
            lib.panda_set_device(dev)
            lib.panda_msm_unregister_bases(b)
            for p in (b, sc, r):
                free(dev, p)
        return dt, ph

    def ntt_leg(total_log, reps):
        g = G.bit_length() - 1
        m = (1 << total_log) >> g
        omega = _root_of_unity_host(total_log)
        slabs, scr = [], []
        for d, dev in enumerate(devices):
            a, b = alloc(dev, m * 32), alloc(dev, m * 32)
            ffi.check(lib.panda_gen_scalars(0, 0x4E5455, d * m, m, a, null), "gen")
            slabs.append(a)
            scr.append(b)
        out = {}
        for label, inv in (("ms", False), ("inverse_ms", True)):
            for _ in range(2):
                mg.ntt([p.value for p in slabs], [p.value for p in scr], omega, total_log, inverse=inv)
            t0 = time.perf_counter()
            for _ in range(reps):  # every transform's output is a valid input of the next: no refill inside the loop
                mg.ntt([p.value for p in slabs], [p.value for p in scr], omega, total_log, inverse=inv)
            out[label] = (time.perf_counter() - t0) / reps * 1e3
        # the same transform as a pipelined batch (panda_ntt_execute_bn254_multi_batch): the all-to-all of transform t behind the kernels of t + 1 / t - 1
        batch = 4
        more_s, more_b = [slabs], [scr]
        for _ in range(batch - 1):
            row_s, row_b = [], []
            for d, dev in enumerate(devices):
                a, b = alloc(dev, m * 32), alloc(dev, m * 32)
                ffi.check(lib.panda_gen_scalars(0, 0x4E5456, d * m, m, a, null), "gen")
                row_s.append(a)
                row_b.append(b)
            more_s.append(row_s)
            more_b.append(row_b)
        ps, pb = [[p.value for p in row] for row in more_s], [[p.value for p in row] for row in more_b]
        mg.ntt_batch(ps, pb, omega, total_log)
        t0 = time.perf_counter()
        for _ in range(reps):  # flags ignored: whichever buffer holds a transform's output, slab and scratch both hold valid field elements
            mg.ntt_batch(ps, pb, omega, total_log)
        out["batch_of_4_ms_per_transform"] = (time.perf_counter() - t0) / reps / batch * 1e3
        for row_s, row_b in zip(more_s, more_b):
            for dev, a, b in zip(devices, row_s, row_b):
                free(dev, a)
                free(dev, b)
        out.update({"value": (1 << total_log) / (out["ms"] * 1e-3), "log_n_total": total_log, "elements_per_gpu": m,
                    "exchange_bytes_per_gpu": m * 32 * (G - 1) // G})
        return out

    log_n = args.log_n
    n = 1 << log_n
    pcie = {}
    dt, ph = msm_leg(log_n, SEED, args.warmup, args.steps, tables=not args.no_tables, from_host=pcie)
    acc_ms = ph[0][3]
    achieved = BYTES_PER_POINT[0] * n / (acc_ms * 1e-3) / 1e9
    out = {
        "metric": "MSM points/s (BN254, 2^24)", "value": G * n * args.steps / dt, "unit": "points/s", "n_gpus": G, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": f"BN254 MSM 2^{log_n} points per GPU, Jacobian output, bases and scalars resident in HBM ({G} base range(s))", "curve": "bn254",
                   "log_points_per_gpu": log_n, "bases": "cached: registered with precomputed window tables" if not args.no_tables else "cached: registered",
                   "sharding": f"base-range x{G}, ONE process: panda_msm_execute_bn254_multi (worker thread per device inside the library)",
                   "exchange": f"ncclAllGather of 96 B partials ({how}) + host point additions"},
        "roofline": {"bound": "hbm", "kernel": "k_accumulate", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "algorithmic_bytes_per_launch": BYTES_PER_POINT[0] * n, "kernel_ms": acc_ms},
        "device_ms_by_rank": [round(p[7], 3) for p in ph],
        "pcie_inclusive": pcie,
    }
    if not args.no_config4 and G & (G - 1) == 0:
        log_per = args.config4_total_log_n - (G.bit_length() - 1)
        dt4, ph4 = msm_leg(log_per, SEED ^ 0xC4, 1, 3)
        out["config4_msm_2_26"] = {"value": (1 << args.config4_total_log_n) * 3 / dt4, "unit": "points/s", "ms_per_step": dt4 / 3 * 1e3, "n_gpus": G, "scaling": "strong",
                                   "log_points_per_gpu": log_per, "device_ms_by_rank": [round(p[7], 3) for p in ph4]}
    if not args.no_ntt and G & (G - 1) == 0:
        g = G.bit_length() - 1
        res = {"exchange": f"grouped ncclSend / ncclRecv all-to-all inside panda_ntt_execute_bn254_multi ({how})", "n_gpus": G, "unit": "elements/s"}
        res["strong_2_24_total"] = dict(ntt_leg(24, 5), scaling="strong")
        res["weak_2_24_per_gpu"] = dict(ntt_leg(min(24 + g, 28), 5), scaling="weak")
        res["value"] = res["strong_2_24_total"]["value"]
        out["ntt_sharded"] = res
    mg.close()
    return out


def c_abi_leg(args, world: int) -> dict:
    """N > 1: the same sharded workloads through the single-process C entry points, run by rank 0 in a child process over all N devices
    while the torch.distributed ranks wait on the host (their legs are finished and their buffers freed).  A child with a time limit:
    whatever happens in there, the contract line of this run is still printed."""
    import subprocess

    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(world), "--single-process", "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--log-n", str(args.log_n), "--config4-total-log-n", str(args.config4_total_log_n)]
    if args.all_on_device0:
        cmd.append("--loopback")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK", "LOCAL_WORLD_SIZE")}
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=420, env=env)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        raise RuntimeError(f"single-process child failed (rc {r.returncode}): {r.stdout[-300:]} {r.stderr[-500:]}")
    return json.loads(lines[-1])


def _sig(x, digits: int = 5):
    """numbers of the contract line carry five significant digits: the line has to fit the 8 KB the driver keeps of it"""
    if isinstance(x, bool) or not isinstance(x, float):
        return x
    return float(f"{x:.{digits}g}")


def _get(d, *path):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return None
        d = d[k]
    return d


def contract_line(full: dict) -> dict:
    """The ONE line the driver records, cut down from the run's full record (`--detail` keeps that): the contract's keys, then
    `configs_ms` -- milliseconds per call of every BASELINE.json configuration and sweep size, flat --, `roofline`, `cpu_baseline`, and the
    few numbers behind each.  No prose: what the keys mean is in DESIGN.md section 7."""
    keys = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    line = {k: _sig(full[k]) for k in keys if k in full}
    line["config"] = {k: v for k, v in full.get("config", {}).items() if not isinstance(v, dict)}
    if full.get("launcher"):
        line["launcher"] = full["launcher"]

    ms = {}

    def put(name, *path, scale=1.0):
        v = _get(full, *path)
        if isinstance(v, (int, float)) and v > 0:
            ms[name] = _sig(v * scale, 4)

    put("c1_host_msm_2_16_cpu", "config1_host_msm_2_16", "ms")
    put("c2_msm_2_20_tables", "config2_msm_2_20", "with_tables", "ms_per_step")
    put("c2_msm_2_20_registered", "config2_msm_2_20", "registered_only", "ms_per_step")
    put("c3_ntt_2_24_fwd", "ntt", "ms")
    put("c3_ntt_2_24_inv", "ntt", "inverse_ms")
    put("c4_msm_2_26_total", "config4_msm_2_26", "ms_per_step")
    put("c5_bls377_2_24_proj_tables", "config5_bls12_377_2_24_projective", "with_tables", "ms_per_step")
    put("c5_bls377_2_24_proj_registered", "config5_bls12_377_2_24_projective", "registered_only", "ms_per_step")
    put("msm_2_24_registered", "without_tables", "ms_per_step")
    put("msm_2_24_unregistered", "without_tables", "unregistered", "ms_per_step")
    put("msm_2_24_from_host_one_call", "pcie_inclusive", "single_call_pipelined", "ms")
    put("msm_2_24_upload_then_execute", "pcie_inclusive", "ms")
    put("msm_2_22_tables", "msm_2_22", "with_tables", "ms_per_step")
    put("msm_2_22_registered", "msm_2_22", "registered_only", "ms_per_step")
    for k in (20, 22, 26):
        put(f"ntt_2_{k}_fwd", "ntt", "sweep", f"2^{k}", "ms")
    put("ntt_bls377_2_24_fwd", "ntt_bls12_377", "ms")
    put("ntt_bls377_2_24_inv", "ntt_bls12_377", "inverse_ms")
    put("g2_msm_2_20_tables", "bn254_g2_msm_2_20", "with_tables", "ms_per_step")
    put("ntt_sharded_2_24_total", "ntt_sharded", "strong_2_24_total", "ms")
    put("ntt_sharded_2_24_total_inv", "ntt_sharded", "strong_2_24_total", "inverse_ms")
    put("ntt_sharded_2_24_per_gpu", "ntt_sharded", "weak_2_24_per_gpu", "ms")
    put("ntt_sharded_batch4_per_transform", "ntt_sharded", "strong_2_24_total", "batch_of_4_ms_per_transform")
    put("c_abi_msm_2_24_per_gpu", "c_abi_single_process", "ms_per_step")
    put("c_abi_c4_msm_2_26_total", "c_abi_single_process", "configs_ms", "c4_msm_2_26_total")
    put("c_abi_ntt_sharded_2_24_total", "c_abi_single_process", "configs_ms", "ntt_sharded_2_24_total")
    put("c_abi_msm_2_24_from_host", "c_abi_single_process", "configs_ms", "msm_2_24_from_host_one_call")
    put("msm_2_24_from_host_one_call", "pcie_inclusive", "from_host_multi_ms")  # --single-process: panda_msm_execute_bn254_from_host_multi
    line["configs_ms"] = ms
    c4 = full.get("config4_msm_2_26")
    if isinstance(c4, dict) and "log_points_per_gpu" in c4:
        line["c4_shape"] = {"n_gpus": c4.get("n_gpus"), "log_points_per_gpu": c4["log_points_per_gpu"], "scaling": "strong"}

    rf = full.get("roofline")
    if rf:
        line["roofline"] = {k: _sig(v) for k, v in rf.items()}
    if "cpu_baseline" in full:
        cb = full["cpu_baseline"]
        line["cpu_baseline"] = {k: _sig(cb[k]) for k in ("value", "unit", "cores", "kind", "sample", "error") if k in cb}
        if "multi_thread" in cb:
            line["cpu_baseline"]["value_16_threads"] = _sig(cb["multi_thread"]["value"])
    ri = full.get("roofline_issue")
    if ri:
        line["roofline_issue"] = {k: _sig(ri[k]) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "additions_per_launch", "mads_per_addition",
                                                           "k_accumulate_mcycles", "sclk_mhz") if k in ri}
        if ri.get("sclk_mhz_by_xcd"):
            line["roofline_issue"]["sclk_mhz_by_xcd"] = ri["sclk_mhz_by_xcd"]
        line["k_accumulate_mcycles"] = _sig(ri.get("k_accumulate_mcycles"))
    if "phases_ms" in full:
        line["phases_ms"] = {k: _sig(v, 4) for k, v in full["phases_ms"].items()}
    vr = {f"2^{k}": _sig(_get(full, leg, "vs_reference_model", "value"), 3) for k, leg in ((20, "config2_msm_2_20"), (22, "msm_2_22")) if _get(full, leg, "vs_reference_model", "value")}
    if _get(full, "vs_reference_model", "value"):
        vr["2^24"] = _sig(full["vs_reference_model"]["value"], 3)
    if vr:
        line["vs_reference_model"] = dict(vr, kind="model")
    for name in ("ntt", "ntt_bls12_377"):
        nt = full.get(name)
        if isinstance(nt, dict) and "ms" in nt:
            line[name] = {"value": _sig(nt["value"]), "unit": "elements/s", "ms": _sig(nt["ms"], 4), "passes": nt.get("passes"), "radix_bits": nt.get("radix_bits"),
                          "roofline": {"bound": "hbm", "achieved": _sig(_get(nt, "roofline", "achieved")), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": _sig(_get(nt, "roofline", "frac"), 4)},
                          "issue_frac": _sig(_get(nt, "roofline_issue", "frac"), 4), "mcycles": _sig(nt.get("mcycles")), "sclk_mhz": _sig(nt.get("sclk_mhz"))}
    kms = {}
    for short, leg in (("c2", "config2_msm_2_20"), ("2_22", "msm_2_22"), ("c5", "config5_bls12_377_2_24_projective")):
        v = _get(full, leg, "with_tables", "k_accumulate_ms")
        if v:
            kms[short] = _sig(v, 4)
            f = _get(full, leg, "with_tables", "roofline_issue", "frac")
            if f:
                kms[short + "_issue_frac"] = _sig(f, 3)
            c = _get(full, leg, "with_tables", "k_accumulate_mcycles")
            if c:
                kms[short + "_mcycles"] = _sig(c, 5)
    if kms:
        line["k_accumulate_ms"] = kms
    for k in ("device_ms_per_step", "device_ms_by_rank", "device"):
        if k in full:
            line[k] = _sig(full[k]) if isinstance(full[k], float) else full[k]
    for leg, v in full.items():
        if isinstance(v, dict) and "error" in v and leg not in line:
            line[leg] = {"error": v["error"][:200]}
    for k in ("failed_legs", "soft_failed_legs"):
        if k in full:
            line[k] = full[k]
    return line


def emit(full: dict, args) -> None:
    """print the contract line; keep the un-shortened record beside it where the builder's profiling runs pick it up"""
    if args.launcher_note:
        full["launcher"] = args.launcher_note
    path = args.detail
    if path is None and os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        path = os.path.join(ROOT, "gpurun_out", f"bench_detail_n{full.get('n_gpus', 1)}.json")
    if path:
        try:
            with open(path, "w") as f:
                json.dump(full, f)
        except OSError:
            pass
    print(json.dumps(contract_line(full)), flush=True)


def launch_ranks(args, argv=None, run=None) -> int:
    """`python bench.py --gpus N` typed without a launcher in front: start torch.distributed.run on this file as a CHILD process --
    before anything in this process has touched the GPU (a process that has never re-executes or replaces itself) --, pass its
    output through and return its exit code.  If the launcher is missing, or its run ends without a contract line, the same N devices
    are measured through the single-process C entry points instead (again in a child), and the line says so (`launcher`)."""
    import importlib.util
    import socket
    import subprocess

    argv = list(sys.argv[1:] if argv is None else argv)
    me = os.path.abspath(__file__)

    def child(cmd):
        """run `cmd`, echo its stdout line by line, return (rc, whether a contract line went by)"""
        if run is not None:
            return run(cmd)
        seen = False
        p = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, bufsize=1)
        for ln in p.stdout:
            seen = seen or (ln.startswith("{") and '"metric"' in ln)
            sys.stdout.write(ln)
            sys.stdout.flush()
        return p.wait(), seen

    why = "torch.distributed.run is not importable"
    if importlib.util.find_spec("torch") is not None and importlib.util.find_spec("torch.distributed.run") is not None:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        rc, seen = child([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), me] + argv)
        if seen:
            return rc
        why = f"torch.distributed.run ended with rc {rc} and no contract line"
    print(f"bench.py: {why}; measuring the {args.gpus} devices through the single-process C entry points", file=sys.stderr, flush=True)
    keep = [a for a in argv if a != "--no-c-abi-leg"]
    if "--all-on-device0" in keep and "--loopback" not in keep:
        keep.append("--loopback")  # the rehearsal's one device plays every rank there too
    rc, _ = child([sys.executable, me] + keep + ["--single-process", "--launcher-note", why])
    return rc


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.single_process and world == 1:
        emit(single_process(args), args)
        return
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit(launch_ranks(args))
        args.gpus = world
    ctx = Ctx(args)
    lib, rank = ctx.lib, ctx.rank

    log_n = args.log_n
    n = 1 << log_n
    prob = MsmProblem(ctx, 0, log_n, SEED, ctx.ffi.JACOBIAN, first=rank * n, register=not args.no_register, tables=not args.no_tables)
    acc_ms, acc_clock = [], []
    clk = (C.c_uint64 * ctx.ffi.CLOCK_WORDS)()

    def step(timed: bool):
        prob.execute()
        if timed:
            acc_ms.append(prob.phases())
            lib.panda_msm_last_clock(clk)
            acc_clock.append([int(v) for v in clk])
        prob.exchange()

    lib.panda_msm_set_phase_timing(1)  # the headline's roofline wants k_accumulate timed inside the timed steps (HIP events on its launch stream)
    # ... and its CYCLES: marker kernels around the launch read s_memtime / s_memrealtime on every CU (panda_set_clock_stamps), so the
    # record can tell a slower device (fewer MHz, same cycles) from a slower kernel (more cycles)
    lib.panda_set_clock_stamps(1)
    dt = ctx.timed(step, args.warmup, args.steps)
    lib.panda_set_clock_stamps(0)
    lib.panda_msm_set_phase_timing(0)

    out = None
    if rank == 0:
        names = [lib.panda_msm_phase_name(i).decode() for i in range(8)]
        mean = [sum(r[i] for r in acc_ms) / len(acc_ms) for i in range(8)]
        acc_kernel_ms = mean[3]  # HIP events around k_accumulate on its launch stream, inside the timed steps
        device_ms = mean[7]
        mean = prob.all_phases()  # every phase: separate untimed calls

        # per timed step: [0] cycles of the slowest-clocked XCD (the one the launch waits for: the cycles the code needed), [1] 10 ns ticks,
        # [2] XCDs stamped, [3] mean cycles over the XCDs (GRBM_GUI_ACTIVE / 8 of the same launch), [4..11] per XCD
        stamped = [c for c in acc_clock if c[2] and c[1]]
        acc_mcycles = sum(c[0] for c in stamped) / len(stamped) / 1e6 if stamped else None
        acc_sclk_mhz = sum(c[3] / c[1] for c in stamped) / len(stamped) * 100.0 if stamped else None
        acc_stamp_ms = sum(c[1] for c in stamped) / len(stamped) * 1e-5 if stamped else None
        acc_xcd_mhz = [round(sum(c[4 + x] / c[1] for c in stamped) / len(stamped) * 100.0) for x in range(8)] if stamped else None

        achieved = BYTES_PER_POINT[0] * n / (acc_kernel_ms * 1e-3) / 1e9
        traffic, traffic_src = None, None
        tr_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tr_path):
            try:
                tr = json.load(open(tr_path))
                if tr.get("log_n") == log_n:
                    traffic = tr.get("k_accumulate_hbm_bytes_per_launch")
                    traffic_src = f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, {tr.get('source')}"
            except Exception:
                traffic = None
        windows = prob.tables if prob.tables > 1 else None
        total_log = log_n + (world.bit_length() - 1)
        workload = f"BN254 MSM 2^{log_n} points per GPU, Jacobian output, bases and scalars resident in HBM"
        if world > 1:
            workload += f" ({world} base ranges, 2^{total_log} points in total)" if world & (world - 1) == 0 else f" ({world} base ranges)"
        out = {
            "metric": "MSM points/s (BN254, 2^24)",
            "value": world * n * args.steps / dt,
            "unit": "points/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": workload, "curve": "bn254", "log_points_per_gpu": log_n,
                       "bases": prob.mode, "sharding": f"base-range x{world}" if world > 1 else "none",
                       "exchange": f"all-gather of 96 B partials ({'RCCL' if ctx.nccl else args.dist_backend}) + host point additions" if world > 1 else "none"},
            "roofline": {"bound": "hbm", "kernel": "k_accumulate", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": BYTES_PER_POINT[0] * n, "kernel_ms": acc_kernel_ms,
                         "k_accumulate_mcycles": acc_mcycles, "sclk_mhz": acc_sclk_mhz},
            "phases_ms": {nm: round(v, 4) for nm, v in zip(names, mean)},
            "device_ms_per_step": device_ms,
            "device": device_identity(ctx),
        }
        if world == 1:
            ref_ms = reference_model_ms(log_n)
            out["vs_reference_model"] = {"value": ref_ms / (dt / args.steps * 1e3), "kind": "model", "reference_model_ms": ref_ms,
                                         "model": "reference CUDA algorithm's mulmod count at c = 16 (n W 11 + W (2^c - 1) 370, SURVEY 8a row a19; msm_cuda.cuh:373-497) "
                                                  "priced at the measured 8x32-bit Montgomery rate on this chip (92.58 G mulmod/s, profiles/r01_ubench_int_rates.txt) "
                                                  "divided by ms_per_step -- not a measurement of the reference, which cannot run without a CUDA device"}
        if windows:
            # what actually bounds k_accumulate: every sorted entry is one XYZZ mixed addition; a digit is zero
            # with probability 2^-c, so W * n additions per launch to within 1e-6 for random scalars
            additions = windows * n
            mads = additions * MADS_PER_ADDITION_BN254 / (acc_kernel_ms * 1e-3)
            out["roofline_issue"] = {"bound": "valu issue (v_mad_u64_u32)", "kernel": "k_accumulate", "achieved": mads / 1e12, "peak": MAD_PEAK_PER_S / 1e12,
                                     "unit": "T mad lane-ops/s", "frac": mads / MAD_PEAK_PER_S, "additions_per_launch": additions,
                                     "mads_per_addition": MADS_PER_ADDITION_BN254, "kernel_ms": acc_kernel_ms,
                                     # clock-normalised: shader cycles of the launch (s_memtime around it, median over the 8 XCDs, mean over the timed steps)
                                     # and the clock it ran at (cycles / s_memrealtime ticks x 100 MHz); cycles are the code's, MHz the box's
                                     "k_accumulate_mcycles": acc_mcycles, "sclk_mhz": acc_sclk_mhz, "stamp_ms": acc_stamp_ms, "sclk_mhz_by_xcd": acc_xcd_mhz,
                                     "mads_per_cycle_per_simd": (additions * MADS_PER_ADDITION_BN254 / 64.0 / (acc_mcycles * 1e6) / 1024.0) if acc_mcycles else None,
                                     "frac_of_bare_chain_rate": mads / MAD_BARE_CHAIN_PER_S,
                                     "valu_model": "4.7 cycles per multiply-add + 4 per other vector instruction on one vector pipe per SIMD reproduces the kernel's time "
                                                   "within 6 % (DESIGN.md section 7): the pipe is saturated, ~30 % of it by the masks, shifts and additions around the products",
                                     "peak_source": "profiles/r01_ubench_int_rates.txt: mad_u64_u32 at 8 waves/SIMD, 28450 Gop/s (a sub-millisecond launch at the nominal "
                                                    "2.4 GHz; this kernel sustains ~2.0 GHz at ~1240 W, profiles/r02_accumulate_stalls.txt)"}

    failed, soft_failed = [], []

    def leg(name, fn, soft=False):
        """Secondary figures never take the contract line down with them: a failure is reported in place of the number, the
        line is still printed, and the process then exits non-zero (soft: reported only -- the leg that cannot be rehearsed on
        the builder's one-GPU box must not cost an 8-GPU run its exit code)."""
        try:
            res = fn()
        except Exception as e:  # noqa: BLE001
            res = {"error": repr(e)[:300]}
            (soft_failed if soft else failed).append(name)
        if rank == 0 and res is not None:
            out[name] = res

    if world == 1 and not args.no_extra_configs:
        # the millisecond-sized configurations run straight after the headline: behind the 2^26 and BLS12-377 legs (seconds of table
        # building, tens of GB of traffic) the chip's sustained clock is ~10 % lower and they would be timed on a heat-soaked GPU
        leg("config2_msm_2_20", lambda: small_config(ctx, 0, 20, ctx.ffi.JACOBIAN, 20, "BN254 MSM 2^20, Jacobian output, cached bases (BASELINE config 2)"))
        leg("msm_2_22", lambda: small_config(ctx, 0, 22, ctx.ffi.JACOBIAN, 10, "BN254 MSM 2^22, Jacobian output, cached bases (north_star sweep 2^20 ... 2^26)"))
    if world == 1 and not args.no_ntt:
        leg("ntt", lambda: ntt_figure(ctx, sweep=not args.no_ntt_sweep))  # BASELINE config 3, also millisecond-sized: before the heavy legs, for the same reason
        if not args.no_ntt_sweep:
            leg("ntt_bls12_377", lambda: ntt_figure(ctx, sweep=False, field=1))
    if world == 1 and not args.no_compare:
        leg("pcie_inclusive", lambda: pcie_inclusive(ctx, prob))
        if prob.tables > 1:
            leg("without_tables", lambda: without_tables(ctx, prob, max(2, args.steps // 2)))
    prob.release()

    if not args.no_config4 and world & (world - 1) == 0 and world <= 64:
        leg("config4_msm_2_26", lambda: config4(ctx, args.config4_total_log_n))
    if not args.no_ntt and world > 1 and world & (world - 1) == 0:
        leg("ntt_sharded", lambda: ntt_sharded_figure(ctx))
    if world == 1 and not args.no_extra_configs:
        leg("config5_bls12_377_2_24_projective",
            lambda: small_config(ctx, 1, 24, ctx.ffi.PROJECTIVE, 5, "BLS12-377 MSM 2^24 + Projective-output conversion (BASELINE config 5)"))
        leg("bn254_g2_msm_2_20", lambda: small_config(ctx, 3, 20, ctx.ffi.JACOBIAN, 5, "BN254 G2 MSM 2^20 (SURVEY 8f-4; coordinates in Fq2), Jacobian output, cached bases"))
    if world == 1 and not args.no_cpu_baseline and rank == 0:
        leg("config1_host_msm_2_16", lambda: config1_host(ctx))
        leg("cpu_baseline", lambda: cpu_baseline(args.cpu_sample_log_n))
    if world > 1 and not args.no_c_abi_leg:
        # the other ranks wait on the HOST (a store key, not a collective: a rank parked in an RCCL barrier would spin on its GPU)
        ctx.torch.cuda.empty_cache()
        ctx.fence()
        store = ctx.dist.distributed_c10d._get_default_store()
        if rank == 0:
            leg("c_abi_single_process", lambda: c_abi_leg(args, world), soft=True)
            store.set("panda_c_abi_leg_done", "1")
        else:
            import datetime

            store.wait(["panda_c_abi_leg_done"], datetime.timedelta(seconds=900))
    if rank == 0:
        if failed:
            out["failed_legs"] = failed
        if soft_failed:  # legs the builder could not rehearse on real hardware (RCCL with more than one rank): reported, exit code kept
            out["soft_failed_legs"] = soft_failed
        emit(out, args)
    if world > 1:
        ctx.dist.barrier()
        ctx.dist.destroy_process_group()
    if failed:
        raise SystemExit(f"bench.py: secondary leg(s) failed: {failed}")


def config4(ctx: Ctx, total_log_n: int) -> dict:
    """BASELINE config 4: BN254 MSM 2^26, sharded by base range over the ranks, partial sums all-gathered and combined.
    Strong scaling: the total is fixed, rank r owns points [r 2^26/N, (r+1) 2^26/N) of one seeded stream."""
    g = ctx.world.bit_length() - 1
    log_per = total_log_n - g
    per = 1 << log_per
    prob = MsmProblem(ctx, 0, log_per, SEED ^ 0xC4, ctx.ffi.JACOBIAN, first=ctx.rank * per, register=True, tables=True)
    steps = 3

    def step(_timed):
        prob.execute()
        prob.exchange()

    dt = ctx.timed(step, 1, steps)
    ph = prob.timed_phases(1, 1)
    res = {"metric": f"MSM points/s (BN254, 2^{total_log_n} in total)", "value": (1 << total_log_n) * steps / dt, "unit": "points/s",
           "ms_per_step": dt / steps * 1e3, "steps": steps, "n_gpus": ctx.world, "scaling": "strong", "log_points_per_gpu": log_per,
           "workload": f"BN254 MSM 2^{total_log_n}, {ctx.world} base range(s) of 2^{log_per} points, all-gather of 96 B partials + combine",
           "bases": prob.mode, "k_accumulate_ms_rank0": ph[3], "device_ms_rank0": ph[7]}
    prob.release()
    return res


def small_config(ctx: Ctx, curve: int, log_n: int, coord: int, steps: int, what: str) -> dict:
    """One of BASELINE.json's single-GPU configurations, with and without precomputed tables."""
    prob = MsmProblem(ctx, curve, log_n, SEED ^ (0x100 + curve * 16 + log_n), coord, register=True, tables=True)
    res = {"workload": what, "unit": "points/s", "steps": steps}
    for label in ("with_tables", "registered_only"):
        if label == "registered_only":
            prob.register(False)
        dt = ctx.timed(lambda _t: prob.execute(), 2, steps) / steps  # no device timers inside these steps (the library's default)
        ph = prob.timed_phases(1, 3)                                 # k_accumulate and the device total: three separate calls
        res[label] = {"value": prob.n / dt, "ms_per_step": dt * 1e3, "bases": prob.mode, "k_accumulate_ms": ph[3], "device_ms": ph[7],
                      "k_accumulate_mcycles": prob.last_mcycles, "sclk_mhz": prob.last_sclk_mhz,
                      "roofline_frac_hbm": BYTES_PER_POINT[curve] * prob.n / (ph[3] * 1e-3) / 1e9 / HBM_PEAK_GBS}
        if curve in MADS_PER_ADDITION and ph[3] > 0:
            # the bound that binds: one mixed addition per sorted entry -- W n with tables, windows * n without
            additions = (prob.tables if prob.tables > 1 else plain_windows(ctx.lib, curve, log_n)) * prob.n
            mads = additions * MADS_PER_ADDITION[curve] / (ph[3] * 1e-3)
            res[label]["roofline_issue"] = {"bound": "valu issue (v_mad_u64_u32)", "achieved": mads / 1e12, "peak": MAD_PEAK_PER_S / 1e12, "unit": "T mad lane-ops/s",
                                            "frac": mads / MAD_PEAK_PER_S, "additions_per_launch": additions, "mads_per_addition": MADS_PER_ADDITION[curve]}
    if curve == 0:
        res["vs_reference_model"] = {"value": reference_model_ms(log_n) / res["with_tables"]["ms_per_step"], "kind": "model", "reference_model_ms": reference_model_ms(log_n)}
    res["value"] = res["with_tables"]["value"]
    prob.release()
    return res


def config1_host(ctx: Ctx, log_n: int = 16) -> dict:
    """BASELINE config 1: BN254 MSM 2^16 through the CPU host-debug entry point (panda_msm_execute_bn254_host: every pointer a host pointer,
    unit.rs:363-416 / msm_host.cuh:267-383) -- the product's own CPU path, one thread; the same inputs through the GPU call must give the same
    affine point (x / z^2, y / z^3 compared with Python integers)"""
    import numpy as np

    torch, lib, ffi = ctx.torch, ctx.lib, ctx.ffi
    n = 1 << log_n
    prob = MsmProblem(ctx, 0, log_n, SEED ^ 0xC1, ffi.JACOBIAN, register=False)
    prob.execute()
    gpu = prob.result.cpu().numpy().view(np.uint32).copy()
    bases, scalars = prob.bases.cpu().numpy(), prob.scalars.cpu().numpy()
    out = np.zeros(96, dtype=np.uint8)
    cfg = ffi.MSMConfiguration(ffi.PandaMemPool(), ffi.PandaStream(), bases.ctypes.data, scalars.ctypes.data, out.ctypes.data, log_n, ffi.JACOBIAN)
    t = time.perf_counter()
    ffi.check(lib.panda_msm_execute_bn254_host(cfg), "host msm")
    dt = time.perf_counter() - t
    p = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47

    def affine(w):  # Jacobian X || Y || Z in Montgomery form -> (x, y) up to the common radix factor, None for the identity
        X, Y, Z = (int.from_bytes(w[8 * i:8 * i + 8].tobytes(), "little") for i in range(3))
        if Z == 0:
            return None
        rinv = pow(1 << 256, -1, p)
        X, Y, Z = X * rinv % p, Y * rinv % p, Z * rinv % p
        zi = pow(Z, -1, p)
        return X * zi * zi % p, Y * zi * zi * zi % p

    same = affine(gpu) == affine(out.view(np.uint32))
    prob.release()
    if not same:
        raise RuntimeError("the CPU entry point and the GPU call disagree on the same inputs")
    return {"value": n / dt, "unit": "points/s", "ms": dt * 1e3, "cores": 1, "matches_gpu": True,
            "workload": f"BN254 MSM 2^{log_n} through panda_msm_execute_bn254_host (BASELINE config 1), one host thread"}


def plain_windows(lib, curve: int, log_n: int) -> int:
    """windows of the plain (no tables) path as the library's policy plans them (pick_window_bits, csrc/msm.hip)"""
    bits, windows = C.c_uint(0), C.c_uint(0)
    lib.panda_msm_plain_window_plan(curve, log_n, C.byref(bits), C.byref(windows))
    return windows.value


def without_tables(ctx: Ctx, prob: MsmProblem, steps: int) -> dict:
    """The same call with the bases registered but no window tables, and with nothing registered at all -- the reference's plain
    panda_msm_execute_bn254 on resident pointers, which converts the bases' radix inside every call (run after the timed region)."""
    prob.register(False)
    dt = ctx.timed(lambda _t: prob.execute(), 1, steps) / steps
    wb, wn = C.c_uint(0), C.c_uint(0)
    ctx.lib.panda_msm_plain_window_plan(prob.curve, prob.log_n, C.byref(wb), C.byref(wn))
    res = {"ms_per_step": dt * 1e3, "value": prob.n / dt, "unit": "points/s", "steps": steps, "k_accumulate_ms": prob.timed_phases(1, 2)[3],
           "windows": f"{wn.value} windows of {wb.value} bits"}
    ctx.ffi.check(ctx.lib.panda_msm_unregister_bases(prob.bases.data_ptr()), "unregister_bases")
    prob.registered = False
    dt = ctx.timed(lambda _t: prob.execute(), 1, steps) / steps
    res["unregistered"] = {"ms_per_step": dt * 1e3, "value": prob.n / dt, "unit": "points/s",
                           "note": "no panda_msm_register_bases: k_convert_bases runs inside every call (what an unmodified caller of the reference's API gets)"}
    return res


def pcie_inclusive(ctx: Ctx, prob: MsmProblem) -> dict:
    """Non-cached-scalars variant (unit.rs:103-188): scalars start in pinned host memory and cross PCIe inside the
    measurement.  Informative only -- never the headline `value` (inputs resident in HBM)."""
    torch, lib, ffi, pstream, cfg, n = ctx.torch, ctx.lib, ctx.ffi, ctx.pstream, prob.cfg, prob.n
    scalars = prob.scalars
    host = torch.empty(n * 32, dtype=torch.uint8).pin_memory()
    host.copy_(scalars.cpu())
    best = 1e9
    for _ in range(3):
        t = time.perf_counter()
        ffi.check(lib.panda_memcpy_async(scalars.data_ptr(), host.data_ptr(), n * 32, pstream), "AsyncMemcopyErr")
        ffi.check(lib.panda_msm_execute_bn254(cfg), "SchedulingErr")
        best = min(best, time.perf_counter() - t)
    out = {"value": n / best, "unit": "points/s", "ms": best * 1e3, "note": "scalars H2D (pinned, 32 B/point) + MSM, bases cached"}
    # the same with the caller-side pipeline the reference's three streams are meant for (wrapper.rs:12-14): the upload of
    # batch k+1 runs on an h2d stream while batch k executes; an event orders each execute after its own upload
    copy_stream = torch.cuda.Stream(device=ctx.dev)
    pcopy = ffi.PandaStream(copy_stream.cuda_stream)
    bufs = [scalars, torch.empty_like(scalars)]
    events = []
    for _ in range(2):
        ev = ffi.PandaEvent()
        ffi.check(lib.panda_event_create(C.byref(ev), True, True), "EventCreateErr")
        events.append(ev)
    reps = 6
    torch.cuda.synchronize()
    t = time.perf_counter()
    ffi.check(lib.panda_memcpy_async(bufs[0].data_ptr(), host.data_ptr(), n * 32, pcopy), "AsyncMemcopyErr")
    ffi.check(lib.panda_event_record(events[0], pcopy), "EventRecordErr")
    for k in range(reps):
        cur, nxt = k & 1, (k + 1) & 1
        if k + 1 < reps:
            ffi.check(lib.panda_memcpy_async(bufs[nxt].data_ptr(), host.data_ptr(), n * 32, pcopy), "AsyncMemcopyErr")
            ffi.check(lib.panda_event_record(events[nxt], pcopy), "EventRecordErr")
        ffi.check(lib.panda_stream_wait_event(pstream, events[cur]), "StreamWaitEventErr")
        c2 = ffi.MSMConfiguration(cfg.mem_pool, cfg.stream, cfg.bases, bufs[cur].data_ptr(), cfg.results, cfg.log_scalars_count, cfg.msm_result_coordinate_type)
        ffi.check(lib.panda_msm_execute_bn254(c2), "SchedulingErr")
    dt = (time.perf_counter() - t) / reps
    for ev in events:
        lib.panda_event_destroy(ev)
    out["pipelined"] = {"value": n / dt, "unit": "points/s", "ms": dt * 1e3, "note": "double-buffered scalar upload on a second stream, steady state over 6 batches"}
    # ONE call with the upload pipelined inside it (panda_msm_execute_from_host): the scalars cross PCIe in point ranges
    # while the previous range is already being sorted and accumulated; what gpu_manager's with_cached_bases path does
    single = {}
    for chunks in (3, 4, 5, 6):
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize()
            t = time.perf_counter()
            ffi.check(lib.panda_msm_execute_from_host(0, cfg, host.data_ptr(), chunks, pcopy), "SchedulingErr")
            best = min(best, time.perf_counter() - t)
        single[str(chunks)] = round(best * 1e3, 3)
    pick = min(single, key=single.get)
    out["single_call_pipelined"] = {"value": n / (single[pick] * 1e-3), "unit": "points/s", "ms": single[pick], "ranges": int(pick), "ms_by_ranges": single,
                                    "note": "one panda_msm_execute_from_host call: pinned host scalars uploaded in point ranges (n/2^(R-1), n/2^(R-1), ..., n/2) beside the execution, "
                                            "each range's buckets added into the running total by its fix-up"}
    return out


def ntt_one(ctx: Ctx, log_n: int, reps: int, field: int = 0) -> dict:
    """forward and inverse NTT of 2^log_n device-resident elements over BN254 Fr (field 0) or BLS12-377 Fr (field 1): warm-up calls,
    then the median of `reps`"""
    torch, lib, ffi = ctx.torch, ctx.lib, ctx.ffi
    n = 1 << log_n
    a = torch.empty(n * 32, dtype=torch.uint8, device=ctx.dev)
    b = torch.empty(n * 32, dtype=torch.uint8, device=ctx.dev)
    ffi.check(lib.panda_gen_scalars(field, 0x4E5454, 0, n, a.data_ptr(), ctx.pstream), "gen")
    omega = _root_of_unity_host(log_n, field)
    flag = C.c_uint(0)
    cfg = ffi.NttconfigurationV1(ffi.PandaMemPool(), ctx.pstream, a.data_ptr(), b.data_ptr(), C.c_void_p(omega.ctypes.data), log_n, C.pointer(flag))

    def timed(fn):
        """median of `reps` calls: (device time of the passes from the library's own HIP events on the launch stream -- SURVEY 8d's
        t_exec, first kernel to result ready --, host wall time of the synchronous call)"""
        dev_ts, wall_ts = [], []
        ms = C.c_float(0)
        # >= 3 warm-ups, median of >= 10.  The warm-up also has to carry the chip from idle to its sustained clock: the first ~20 ms of work
        # after an idle period run ~7 % slow (tools/ntt_order.py: 1.61 ms for the first block of 2^24 transforms, 1.50 ms for every later one), so
        # the untimed calls go on until 40 ms have passed
        t_warm, warm = time.perf_counter(), 0
        while warm < 3 or time.perf_counter() - t_warm < 0.04:
            ffi.check(fn(cfg), "ntt")
            warm += 1
        clk, cyc = (C.c_uint64 * ffi.CLOCK_WORDS)(), []
        for r in range(reps):
            t = time.perf_counter()
            ffi.check(fn(cfg), "ntt")
            w = time.perf_counter() - t
            ffi.check(lib.panda_ntt_last_device_ms(C.byref(ms)), "ntt_ms")
            lib.panda_ntt_last_clock(clk)
            if clk[1] and clk[2]:
                cyc.append((int(clk[0]), int(clk[3]) / int(clk[1]) * 100.0))  # cycles of the slowest-clocked XCD; mean clock of the XCDs
            wall_ts.append(w)
            dev_ts.append(ms.value * 1e-3)
        dev_ts.sort()
        wall_ts.sort()
        cyc.sort()
        clock.append(cyc[len(cyc) // 2] if cyc else (None, None))
        return dev_ts[len(dev_ts) // 2], wall_ts[len(wall_ts) // 2]

    clock = []  # per direction: (shader cycles of the passes, MHz), medians -- s_memtime / s_memrealtime markers around the passes (panda_set_clock_stamps)
    lib.panda_set_clock_stamps(1)
    try:
        fwd, fwd_wall = timed(lib.panda_ntt_execute_bls12_377_v1 if field else lib.panda_ntt_execute_bn254_v1)
        inv, inv_wall = timed(lib.panda_ntt_execute_bls12_377_inverse if field else lib.panda_ntt_execute_bn254_inverse)  # omega^-1 passes + fused n^-1
    finally:
        lib.panda_set_clock_stamps(0)
    gbs = BYTES_PER_NTT_ELEM * n / fwd / 1e9
    passes, bits = C.c_uint(0), (C.c_uint * 4)()
    ffi.check(lib.panda_ntt_pass_plan(log_n, C.byref(passes), bits), "ntt_plan")
    del a, b
    torch.cuda.empty_cache()
    return {"value": n / fwd, "unit": "elements/s", "ms": fwd * 1e3, "inverse_ms": inv * 1e3, "inverse_elements_per_s": n / inv,
            "forward_plus_inverse_ms": (fwd + inv) * 1e3, "forward_plus_inverse_elements_per_s": n / (fwd + inv),
            "wall_ms": fwd_wall * 1e3, "inverse_wall_ms": inv_wall * 1e3, "passes": passes.value, "radix_bits": [int(b) for b in bits if b],
            "mcycles": clock[0][0] / 1e6 if clock[0][0] else None, "sclk_mhz": clock[0][1], "inverse_mcycles": clock[1][0] / 1e6 if clock[1][0] else None,
            "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                         "algorithmic_bytes": BYTES_PER_NTT_ELEM * n}}


def ntt_figure(ctx: Ctx, log_n: int = 24, reps: int = 11, sweep: bool = True, field: int = 0) -> dict:
    """BASELINE config 3: BN254 NTT 2^24 forward and inverse, device-resident -- plus the north_star sweep 2^20 / 2^22 / 2^26; field = 1: the
    same transform over the BLS12-377 scalar field (north_star: "NTT butterfly over BN254/BLS12-377"), 2^24 only."""
    res = ntt_one(ctx, log_n, reps, field)
    n24 = 1 << log_n
    res["metric"] = f"NTT elements/s ({'BLS12-377' if field else 'BN254'} Fr, 2^24, forward)"
    if log_n == 24:
        # what binds the passes: v_mad_u64_u32 per thread (8 elements) and pass in k_ntt_pass8 -- 24.25 butterfly products
        # (7 of wave 0's 12 middle-block products are skipped) of 143 each, + 8 output products of 143 in the first pass and 8 Montgomery
        # products (162 + 9 on the same pipe) from the streamed table in the middle pass -- over the transform's device time
        mads_per_element = ((24.25 * 3 + 8) * 143 + 8 * 171) / 8
        mads = n24 * mads_per_element / (res["ms"] * 1e-3)
        res["roofline_issue"] = {"bound": "valu issue (v_mad_u64_u32)", "achieved": mads / 1e12, "peak": MAD_PEAK_PER_S / 1e12, "unit": "T mad lane-ops/s",
                                 "frac": mads / MAD_PEAK_PER_S, "mads_per_element": mads_per_element}
    res["timing"] = ("ms = device time of the passes (HIP events on the launch stream inside the library: first pass to result ready); "
                     "wall_ms = host time of the synchronous call; untimed warm-up calls for 40 ms (at least 3) carry the chip from idle to its sustained clock, then the median of 11")
    if sweep:
        res["sweep"] = {f"2^{k}": ntt_one(ctx, k, 11, field) for k in (20, 22, 26)}
    return res


def ntt_sharded_figure(ctx: Ctx, reps: int = 5) -> dict:
    """N > 1: the slab-sharded transform, composed on the devices (multi_gpu.ntt_sharded: panda_ntt_slab_step1_bn254 ->
    all_to_all_single over RCCL -> panda_ntt_slab_step2_bn254; intt_sharded: the mirrored steps).  Strong: 2^24 elements in total;
    weak: 2^24 per GPU."""
    torch, lib, ffi = ctx.torch, ctx.lib, ctx.ffi
    g = ctx.world.bit_length() - 1
    res = {"exchange": f"all_to_all_single ({'RCCL, device buffers' if ctx.nccl else ctx.args.dist_backend + ', staged through the host'}), "
                       f"each rank sends (N-1)/N of its slab", "n_gpus": ctx.world, "unit": "elements/s"}
    for label, total_log in (("strong_2_24_total", 24), ("weak_2_24_per_gpu", min(24 + g, 28))):
        m = (1 << total_log) >> g
        slab = torch.empty(m * 32, dtype=torch.uint8, device=ctx.dev)
        scratch = torch.empty_like(slab)
        keep = torch.empty_like(slab)
        ffi.check(lib.panda_gen_scalars(0, 0x4E5455, ctx.rank * m, m, keep.data_ptr(), ctx.pstream), "gen")
        omega = _root_of_unity_host(total_log)

        def step(_timed):
            slab.copy_(keep)  # the transform overwrites both buffers; the refill is a device-to-device copy inside the loop
            ctx.multi_gpu.ntt_sharded(slab, scratch, omega, total_log, stream=ctx.stream)

        def refill_only(_timed):
            slab.copy_(keep)

        def step_inverse(_timed):
            slab.copy_(keep)
            ctx.multi_gpu.intt_sharded(slab, scratch, omega, total_log, stream=ctx.stream)

        dt = ctx.timed(step, 2, reps) / reps
        dt_inv = ctx.timed(step_inverse, 2, reps) / reps
        dt_copy = ctx.timed(refill_only, 1, reps) / reps
        t = max(dt - dt_copy, 1e-9)
        res[label] = {"value": (1 << total_log) / t, "ms": t * 1e3, "inverse_ms": max(dt_inv - dt_copy, 1e-9) * 1e3, "log_n_total": total_log, "elements_per_gpu": m,
                      "exchange_bytes_per_gpu": m * 32 * (ctx.world - 1) // ctx.world,
                      "scaling": "strong" if label.startswith("strong") else "weak"}
        del slab, scratch, keep
        torch.cuda.empty_cache()
    res["value"] = res["strong_2_24_total"]["value"]
    res["metric"] = "NTT elements/s (BN254 Fr, 2^24 in total, slab-sharded forward)"
    return res


def _root_of_unity_host(log_n, field: int = 0):
    """omega of order 2^log_n in wire form: BN254 Fr 7^((r-1)/2^28) squared down (bn254/paramter.cuh:241-258), BLS12-377 Fr
    22^((r-1)/2^47) (two-adicity 47, bls12_377/paramter.cuh:130-181), computed with Python integers -- no oracle involved."""
    import numpy as np

    r, gen, adicity = ((0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001, 7, 28),
                       (0x12AB655E9A2CA55660B44D1E5C37B00159AA76FED00000010A11800000000001, 22, 47))[field]
    w = pow(gen, (r - 1) >> adicity, r)
    w = pow(w, 1 << (adicity - log_n), r)
    assert pow(w, 1 << log_n, r) == 1 and (log_n == 0 or pow(w, 1 << (log_n - 1), r) == r - 1)
    return np.frombuffer((w * (1 << 256) % r).to_bytes(32, "little"), dtype=np.uint32).copy()


if __name__ == "__main__":
    main()
