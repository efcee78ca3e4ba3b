"""One-process-per-GPU sharding of the two hot paths (no reference counterpart: the reference is single-GPU,
msm_cuda.cuh:554-555, wrapper.rs:38).

MSM shards by base-point range: rank r owns points [r n/G, (r+1) n/G) with their scalars, runs the ordinary
single-GPU pipeline on them and produces one Jacobian partial (96 B).  The only exchange is an all-gather of
those partials -- EC addition is not an RCCL reduction operator, so it cannot be an all-reduce -- followed by
G - 1 point additions (panda_msm_combine_bn254).  Message size is 96 B per rank: latency-bound, topology-agnostic.

NTT shards by decimated coefficient slab: rank r owns X_r[j2] = x[r + G j2].  panda_ntt_slab_step1_bn254 does the
local size-m transforms and the twiddle w^(r k2); one all-to-all moves chunk q of every rank to rank q (each pair
of GPUs exchanges m/G elements over its own xGMI link, so the full mesh is used, not a ring);
panda_ntt_slab_step2_bn254 does the size-G transforms.  Rank q ends with y[k1 m + q m/G + k2'] stored at [k1][k2'].

torch.distributed is plumbing here: backend "nccl" is RCCL on ROCm; the CPU tests drive the same functions over "gloo".
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import gpu_ffi as ffi


def shard_range(n_total: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous base-point range of `rank`: (first, count)."""
    per = n_total // world
    assert per * world == n_total, "n must divide evenly over the ranks"
    return rank * per, per


def allgather_partials(partial: np.ndarray, group=None, device=None) -> np.ndarray:
    """All-gather one result-sized byte vector per rank; returns (world, nbytes) uint8 on the host."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    src = torch.from_numpy(np.ascontiguousarray(partial).view(np.uint8).reshape(-1).copy())
    if device is not None:
        src = src.to(device)
    out = torch.empty(world * src.numel(), dtype=torch.uint8, device=src.device)
    dist.all_gather_into_tensor(out, src, group=group)
    return out.cpu().numpy().reshape(world, -1)


def combine_partials(partials: np.ndarray, curve: int = 0, coordinate_type: int = ffi.JACOBIAN) -> np.ndarray:
    """Sum of Jacobian partials (rows of `partials`) through panda_msm_combine_*; 96 / 144 bytes out."""
    lib = ffi.load()
    p = np.ascontiguousarray(partials).view(np.uint8)
    count = p.shape[0]
    out = np.zeros(p.shape[1], dtype=np.uint8)
    fn = (lib.panda_msm_combine_bn254, lib.panda_msm_combine_bls12_377, lib.panda_msm_combine_bls12_381, lib.panda_msm_combine_bn254_g2)[curve]
    ffi.check(fn(C.c_void_p(p.ctypes.data), count, coordinate_type, C.c_void_p(out.ctypes.data)), "SchedulingErr")
    return out


def msm_sharded(local_partial: np.ndarray, curve: int = 0, coordinate_type: int = ffi.JACOBIAN, group=None, device=None) -> np.ndarray:
    """The exchange + combine half of a sharded MSM: every rank passes the Jacobian partial of its base range and
    receives the total."""
    return combine_partials(allgather_partials(local_partial, group, device), curve, coordinate_type)


# ------------------------------------------------------------------------------------------------ NTT slabs

def slab_of(x: np.ndarray, world: int, rank: int) -> np.ndarray:
    """Input layout: rank r holds the decimated sequence x[r + G j2]."""
    return np.ascontiguousarray(x[rank::world])


def natural_from_slab_outputs(outs: list[np.ndarray]) -> np.ndarray:
    """Inverse of the output layout: outs[q][k1][k2'] = y[k1 m + q m/G + k2']."""
    world = len(outs)
    m = outs[0].shape[0]
    per = m // world
    width = outs[0].shape[1]
    y = np.empty((world * m, width), dtype=outs[0].dtype)
    for q, o in enumerate(outs):
        o = o.reshape(world, per, width)
        for k1 in range(world):
            y[k1 * m + q * per:k1 * m + (q + 1) * per] = o[k1]
    return y


def all_to_all_slab(tensor, group=None):
    """The single exchange of the sharded NTT: chunk q of every rank goes to rank q (torch all_to_all_single)."""
    import torch
    import torch.distributed as dist

    out = torch.empty_like(tensor)
    dist.all_to_all_single(out, tensor, group=group)
    return out


def _slab_step(step: int, pstream, d_slab: int, d_scratch: int, omega: np.ndarray, log_n: int, log_ranks: int, rank: int, wait: bool = True) -> int:
    """One device half of the sharded transform through the C ABI; returns the flag (1: the output sits in d_scratch).
    wait=False uses the *_enqueue entry points: nothing is synchronised, the work is ordered on `pstream`."""
    lib = ffi.load()
    flag = C.c_uint(7)
    cfg = ffi.NttSlabConfiguration(pstream, C.c_void_p(d_slab), C.c_void_p(d_scratch), C.c_void_p(omega.ctypes.data), log_n, log_ranks, rank,
                                   C.pointer(flag))
    if step < 0:  # the mirrored steps of the inverse transform (enqueue only)
        fn = lib.panda_ntt_slab_inverse_step1_bn254_enqueue if step == -1 else lib.panda_ntt_slab_inverse_step2_bn254_enqueue
    elif wait:
        fn = lib.panda_ntt_slab_step1_bn254 if step == 1 else lib.panda_ntt_slab_step2_bn254
    else:
        fn = lib.panda_ntt_slab_step1_bn254_enqueue if step == 1 else lib.panda_ntt_slab_step2_bn254_enqueue
    ffi.check(fn(cfg), "SchedulingErr")
    assert flag.value in (0, 1)
    return flag.value


def _log2_exact(v: int) -> int:
    assert v > 0 and v & (v - 1) == 0, "the rank count must be a power of two"
    return v.bit_length() - 1


def ntt_sharded(slab, scratch, omega, log_n: int, group=None, stream=None):
    """The sharded BN254 transform of one rank, composed on the device:
        panda_ntt_slab_step1_bn254  ->  all_to_all_single (RCCL over xGMI)  ->  panda_ntt_slab_step2_bn254.

    `slab` holds this rank's decimated input X_r[j2] = x[r + G j2] (n/G elements of 32 bytes) and `scratch` is a buffer of
    the same size; both are torch uint8 tensors on this rank's device and both are overwritten.  `omega` is the primitive
    n-th root in wire form (host, 8 x u32).  Returns whichever of the two tensors holds this rank's output
    y[k1 m + q m/G + k2'] at [k1][k2'] (q = rank, m = n/G).

    Every device step runs on `stream` (default: torch's current stream); the collective is issued with that stream current,
    so torch orders it after step 1 and step 2 after it.  With the gloo backend (CPU rehearsal of the N > 1 path) the
    exchange is staged through host memory, the two device halves are the same kernels."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    g = _log2_exact(world)
    om = np.ascontiguousarray(np.asarray(omega).view(np.uint32).reshape(-1))
    assert slab.is_cuda and scratch.is_cuda and slab.numel() == scratch.numel() == ((1 << log_n) >> g) * 32
    s = stream if stream is not None else torch.cuda.current_stream(slab.device)
    pstream = ffi.PandaStream(s.cuda_stream)
    with torch.cuda.stream(s):
        # everything is enqueued on `s`: step 1 does not wait (its flag is the parity of the pass count), the collective is ordered
        # after it by torch, step 2 follows on the same stream; one synchronisation at the end
        flag = _slab_step(1, pstream, slab.data_ptr(), scratch.data_ptr(), om, log_n, g, rank, wait=False)
        src, dst = (scratch, slab) if flag else (slab, scratch)
        if dist.get_backend(group) == "nccl":
            dist.all_to_all_single(dst, src, group=group)  # chunk q of every rank -> rank q; each GPU pair has its own xGMI link
        else:
            recv = torch.empty(src.numel(), dtype=torch.uint8)
            dist.all_to_all_single(recv, src.cpu(), group=group)  # .cpu() waits for step 1 on this stream
            dst.copy_(recv)
        flag = _slab_step(2, pstream, dst.data_ptr(), src.data_ptr(), om, log_n, g, rank, wait=False)
        s.synchronize()
    return src if flag else dst


def intt_sharded(slab, scratch, omega, log_n: int, group=None, stream=None):
    """The inverse of ntt_sharded: `slab` holds this rank's part of the forward transform's OUTPUT (rank q: y[k1 m + q m/G + k2'] at
    [k1][k2']); returns whichever of slab / scratch ends up holding this rank's part of the INPUT layout (rank r: x[r + G j2]), n^-1
    included.  The steps run backwards: size-G inverse transforms, the same all-to-all, twiddle and the local inverse transform.
    `omega` is the forward root."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    g = _log2_exact(world)
    om = np.ascontiguousarray(np.asarray(omega).view(np.uint32).reshape(-1))
    assert slab.is_cuda and scratch.is_cuda and slab.numel() == scratch.numel() == ((1 << log_n) >> g) * 32
    s = stream if stream is not None else torch.cuda.current_stream(slab.device)
    pstream = ffi.PandaStream(s.cuda_stream)
    with torch.cuda.stream(s):
        flag = _slab_step(-1, pstream, slab.data_ptr(), scratch.data_ptr(), om, log_n, g, rank, wait=False)
        src, dst = (scratch, slab) if flag else (slab, scratch)
        if dist.get_backend(group) == "nccl":
            dist.all_to_all_single(dst, src, group=group)  # row r of every rank -> rank r: the forward exchange run backwards
        else:
            recv = torch.empty(src.numel(), dtype=torch.uint8)
            dist.all_to_all_single(recv, src.cpu(), group=group)
            dst.copy_(recv)
        flag = _slab_step(-2, pstream, dst.data_ptr(), src.data_ptr(), om, log_n, g, rank, wait=False)
        s.synchronize()
    return src if flag else dst


def intt_sharded_one_process(outs, scratches, omega, log_n: int, stream=None):
    """intt_sharded with all G ranks played by one process on one GPU (device-to-device copies for the exchange); `outs` are the
    forward transform's per-rank outputs.  Returns the per-rank input slabs."""
    import torch

    world = len(outs)
    g = _log2_exact(world)
    om = np.ascontiguousarray(np.asarray(omega).view(np.uint32).reshape(-1))
    s = stream if stream is not None else torch.cuda.current_stream(outs[0].device)
    pstream = ffi.PandaStream(s.cuda_stream)
    res = []
    with torch.cuda.stream(s):
        pairs = []
        for q in range(world):
            flag = _slab_step(-1, pstream, outs[q].data_ptr(), scratches[q].data_ptr(), om, log_n, g, q, wait=False)
            pairs.append((scratches[q], outs[q]) if flag else (outs[q], scratches[q]))
        s.synchronize()
        for r in range(world):
            dst = pairs[r][1].view(world, -1)
            for q in range(world):
                dst[q].copy_(pairs[q][0].view(world, -1)[r])
        s.synchronize()
        for r in range(world):
            src, dst = pairs[r]
            flag = _slab_step(-2, pstream, dst.data_ptr(), src.data_ptr(), om, log_n, g, r, wait=False)
            res.append(src if flag else dst)
        s.synchronize()
    return res


def ntt_sharded_one_process(slabs, scratches, omega, log_n: int, stream=None):
    """The same composition with all G ranks played by one process on one GPU: the slabs go through step 1 one after the
    other, the all-to-all is G x G device-to-device copies, then step 2.  It exists so that the buffer protocol around the
    exchange (which of slab / scratch each step leaves its output in, the chunk layout) can be tested against the real
    kernels on a one-GPU box.  Returns the list of output tensors, rank order."""
    import torch

    world = len(slabs)
    g = _log2_exact(world)
    om = np.ascontiguousarray(np.asarray(omega).view(np.uint32).reshape(-1))
    s = stream if stream is not None else torch.cuda.current_stream(slabs[0].device)
    pstream = ffi.PandaStream(s.cuda_stream)
    outs = []
    with torch.cuda.stream(s):
        pairs = []
        for r in range(world):
            flag = _slab_step(1, pstream, slabs[r].data_ptr(), scratches[r].data_ptr(), om, log_n, g, r)
            pairs.append((scratches[r], slabs[r]) if flag else (slabs[r], scratches[r]))
        for q in range(world):  # what all_to_all_single delivers to rank q: chunk q of every rank, in rank order
            dst = pairs[q][1].view(world, -1)
            for j1 in range(world):
                dst[j1].copy_(pairs[j1][0].view(world, -1)[q])
        s.synchronize()
        for q in range(world):
            src, dst = pairs[q]
            flag = _slab_step(2, pstream, dst.data_ptr(), src.data_ptr(), om, log_n, g, q)
            outs.append(src if flag else dst)
    return outs


# ------------------------------------------------------------------------------------------------ one process, C ABI (csrc/multi_gpu.hip)

class MultiGpu:
    """ctypes face of panda_multi_gpu: ONE process drives all devices through the C entry points (one worker thread, one stream and one
    RCCL communicator per device inside the library) -- what a Rust / C host would call.  torch is not involved.

        mg = MultiGpu([0, 1, 2, 3])                   # transport: ffi.MULTI_RCCL (default) or ffi.MULTI_LOOPBACK
        total = mg.msm(cfgs, result_bytes=96)          # cfgs[d]: ffi.MSMConfiguration on devices[d]
        flags = mg.ntt(slab_ptrs, scratch_ptrs, omega, log_n)
    """

    def __init__(self, devices, transport: int = ffi.MULTI_RCCL):
        self.lib = ffi.load()
        self.devices = list(devices)
        self.n = len(self.devices)
        self.handle = ffi.PandaMultiGpu()
        arr = (C.c_int * self.n)(*self.devices)
        ffi.check(self.lib.panda_multi_gpu_create(C.byref(self.handle), arr, self.n, transport), "CreateContextError")

    def close(self):
        if self.handle.handle:
            self.lib.panda_multi_gpu_destroy(self.handle)
            self.handle = ffi.PandaMultiGpu()

    def msm(self, cfgs, curve: int = 0) -> np.ndarray:
        """panda_msm_execute_*_multi: base ranges on the devices, one all-gather of partials, the total on the host."""
        assert len(cfgs) == self.n and curve in (0, 1, 2, 3)  # BN254, BLS12-377, BLS12-381 (G1), BN254 G2
        arr = (ffi.MSMConfiguration * self.n)(*cfgs)
        out = np.zeros((96, 144, 144, 192)[curve], dtype=np.uint8)
        fn = (self.lib.panda_msm_execute_bn254_multi, self.lib.panda_msm_execute_bls12_377_multi, self.lib.panda_msm_execute_bls12_381_multi,
              self.lib.panda_msm_execute_bn254_g2_multi)[curve]
        ffi.check(fn(self.handle, arr, C.c_void_p(out.ctypes.data)), "SchedulingErr")
        return out

    def msm_from_host(self, cfgs, host_ptrs, ranges: int = 4, curve: int = 0) -> np.ndarray:
        """panda_msm_execute_*_from_host_multi: as msm(), with rank d's scalars starting at host address host_ptrs[d] (pinned or
        pageable) and crossing PCIe inside the call -- every device uploads its own shard, in `ranges` point ranges, beside its kernels."""
        assert len(cfgs) == len(host_ptrs) == self.n and curve in (0, 1, 2, 3)
        arr = (ffi.MSMConfiguration * self.n)(*cfgs)
        hp = (C.c_void_p * self.n)(*[C.c_void_p(int(p)) for p in host_ptrs])
        out = np.zeros((96, 144, 144, 192)[curve], dtype=np.uint8)
        fn = (self.lib.panda_msm_execute_bn254_from_host_multi, self.lib.panda_msm_execute_bls12_377_from_host_multi,
              self.lib.panda_msm_execute_bls12_381_from_host_multi, self.lib.panda_msm_execute_bn254_g2_from_host_multi)[curve]
        ffi.check(fn(self.handle, arr, hp, ranges, C.c_void_p(out.ctypes.data)), "SchedulingErr")
        return out

    def ntt(self, slabs, scratches, omega, log_n: int, inverse: bool = False, streams=None, field: int = 0):
        """panda_ntt_execute_{bn254,bls12_377}[_inverse]_multi (field 0 / 1 / 2) on device pointers slabs[d] / scratches[d]; returns the flags
        (1: rank d's output is in scratches[d])."""
        assert len(slabs) == len(scratches) == self.n
        g = _log2_exact(self.n)
        om = np.ascontiguousarray(np.asarray(omega).view(np.uint32).reshape(-1))
        flags = [C.c_uint(7) for _ in range(self.n)]
        cfgs = (ffi.NttSlabConfiguration * self.n)(*[
            ffi.NttSlabConfiguration(streams[d] if streams else ffi.PandaStream(), C.c_void_p(slabs[d]), C.c_void_p(scratches[d]), C.c_void_p(om.ctypes.data),
                                     log_n, g, d, C.pointer(flags[d])) for d in range(self.n)])
        fn = ((self.lib.panda_ntt_execute_bn254_multi, self.lib.panda_ntt_execute_bn254_inverse_multi),
              (self.lib.panda_ntt_execute_bls12_377_multi, self.lib.panda_ntt_execute_bls12_377_inverse_multi),
              (self.lib.panda_ntt_execute_bls12_381_multi, self.lib.panda_ntt_execute_bls12_381_inverse_multi))[field][1 if inverse else 0]
        ffi.check(fn(self.handle, cfgs), "SchedulingErr")
        return [f.value for f in flags]

    def ntt_batch(self, slabs, scratches, omega, log_n: int, inverse: bool = False, streams=None, field: int = 0):
        """panda_ntt_execute_{bn254,bls12_377}[_inverse]_multi_batch: slabs[t][d] / scratches[t][d] are the device pointers of rank d of transform t; the exchange
        of transform t overlaps the kernels of its neighbours.  Returns flags[t][d] (1: the output is in scratches[t][d])."""
        count = len(slabs)
        assert count >= 1 and len(scratches) == count and all(len(a) == self.n and len(b) == self.n for a, b in zip(slabs, scratches))
        g = _log2_exact(self.n)
        om = np.ascontiguousarray(np.asarray(omega).view(np.uint32).reshape(-1))
        flags = [[C.c_uint(7) for _ in range(self.n)] for _ in range(count)]
        cfgs = (ffi.NttSlabConfiguration * (self.n * count))(*[
            ffi.NttSlabConfiguration(streams[d] if streams else ffi.PandaStream(), C.c_void_p(slabs[t][d]), C.c_void_p(scratches[t][d]), C.c_void_p(om.ctypes.data),
                                     log_n, g, d, C.pointer(flags[t][d])) for t in range(count) for d in range(self.n)])
        fn = ((self.lib.panda_ntt_execute_bn254_multi_batch, self.lib.panda_ntt_execute_bn254_inverse_multi_batch),
              (self.lib.panda_ntt_execute_bls12_377_multi_batch, self.lib.panda_ntt_execute_bls12_377_inverse_multi_batch),
              (self.lib.panda_ntt_execute_bls12_381_multi_batch, self.lib.panda_ntt_execute_bls12_381_inverse_multi_batch))[field][1 if inverse else 0]
        ffi.check(fn(self.handle, cfgs, count), "SchedulingErr")
        return [[f.value for f in row] for row in flags]

    def phases(self, rank: int):
        ph = (C.c_float * 8)()
        ffi.check(self.lib.panda_multi_gpu_last_phase_ms(self.handle, rank, ph), "SchedulingErr")
        return list(ph)
