"""Worker for tests/test_multi_gpu_gloo.py: one rank of a world_size-N gloo job on the CPU."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import torch.distributed as dist

    import oracle as po
    import pyref
    from panda_amd import gpu_manager as pgm
    from panda_amd import multi_gpu

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    what = sys.argv[1]
    if what == "msm":
        cid = int(sys.argv[2])
        n = 1 << 11
        first, cnt = multi_gpu.shard_range(n, world, rank)
        # each rank builds only ITS base range and scalars (same seeded stream as the full problem)
        bases = po.gen_bases(cid, 0xBA5E, cnt, first=first)
        scalars = po.gen_scalars(po.FR_OF[cid], 0x5CA1, cnt, first=first)
        partial = pgm.panda_msm_bn254_gpu_host(None, scalars, bases, curve=cid)  # product code path, CPU entry point
        total = multi_gpu.msm_sharded(partial, curve=cid)
        want = po.expected_from_linearity(cid, 0xBA5E, po.gen_scalars(po.FR_OF[cid], 0x5CA1, n))
        ok = (po.to_affine(cid, total.view(np.uint32)) == want).all()
    elif what == "ntt_device":
        # sharded NTT with the product's device halves: every rank drives cuda:0 through multi_gpu.ntt_sharded
        # (step 1 -> all-to-all -> step 2); the exchange is gloo, staged through the host.  Needs a GPU.
        fid, log_n = po.F_BN254_FR, int(sys.argv[2])
        n = 1 << log_n
        m = n // world
        om = po.root_of_unity(fid, log_n)
        x = po.gen_scalars(fid, 0x4E54, n)
        dev = torch.device("cuda", 0)
        slab = torch.from_numpy(multi_gpu.slab_of(x, world, rank).view(np.uint8).reshape(-1).copy()).to(dev)
        scratch = torch.empty_like(slab)
        out = multi_gpu.ntt_sharded(slab, scratch, om, log_n)
        mine = out.cpu()
        gathered = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        y = multi_gpu.natural_from_slab_outputs([t.numpy().view(np.uint32).reshape(m, 8) for t in gathered])
        ok = (y == po.ntt(fid, x, om, log_n)).all()
        back = multi_gpu.intt_sharded(out.clone(), torch.empty_like(out), om, log_n)  # the inverse returns this rank's input slab
        ok = ok and (back.cpu().numpy().view(np.uint32).reshape(m, 8) == multi_gpu.slab_of(x, world, rank)).all()
    else:
        # sharded NTT on a box without a GPU: the exchange and the layouts are the product's (multi_gpu.py); the two local
        # steps are stood in by the oracle because their product implementation is a HIP kernel ("ntt_device" above and
        # the -m gpu tests run the real ones)
        fid, log_n = po.F_BN254_FR, int(sys.argv[2])
        n = 1 << log_n
        g = world.bit_length() - 1
        m = n // world
        c = pyref.CURVES[0]
        om = po.root_of_unity(fid, log_n)
        x = po.gen_scalars(fid, 0x4E54, n)
        slab = multi_gpu.slab_of(x, world, rank)
        w = pyref.decode_scalar(c, om)
        mont = lambda v: pyref.int_to_limbs(v * c.Rr % c.r, 8)
        step1 = po.ntt(fid, slab, mont(pow(w, world, c.r)), log_n - g)
        tw = np.stack([mont(pow(w, rank * k2, c.r)) for k2 in range(m)])
        step1 = po.f_vec(fid, po.OP_MUL, step1, tw)
        recv = multi_gpu.all_to_all_slab(torch.from_numpy(step1.view(np.int32).copy())).numpy().view(np.uint32)
        pieces = recv.reshape(world, m // world, 8)  # [j1][k2']
        out = np.empty_like(pieces)
        wm = mont(pow(w, m, c.r))
        for k2 in range(m // world):
            out[:, k2, :] = po.dft_naive(fid, np.ascontiguousarray(pieces[:, k2, :]), wm, g)
        mine = torch.from_numpy(out.reshape(-1).view(np.int32).copy())
        gathered = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        y = multi_gpu.natural_from_slab_outputs([t.numpy().view(np.uint32).reshape(m, 8) for t in gathered])
        ok = (y == po.ntt(fid, x, om, log_n)).all()
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.destroy_process_group()
    sys.exit(0 if int(flag.item()) == 1 else 1)


if __name__ == "__main__":
    main()
