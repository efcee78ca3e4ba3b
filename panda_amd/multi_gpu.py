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
    fn = (lib.panda_msm_combine_bn254, lib.panda_msm_combine_bls12_377, lib.panda_msm_combine_bls12_381)[curve]
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
