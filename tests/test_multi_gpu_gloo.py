"""N > 1 paths on the CPU: world_size 2 (and 4) over gloo, rendezvous on 127.0.0.1."""
import os
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(world, *args):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_gloo_worker.py"), *args], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, out.decode()[-2000:]))
    assert all(rc == 0 for rc, _ in outs), outs


@pytest.mark.parametrize("cid", [0, 1])
def test_msm_base_range_sharding_world2(cid):
    _launch(2, "msm", str(cid))


def test_msm_base_range_sharding_world4():
    _launch(4, "msm", "0")


@pytest.mark.parametrize("world,log_n", [(2, 8), (4, 9)])
def test_ntt_slab_exchange_and_layout(world, log_n):
    _launch(world, "ntt", str(log_n))


@pytest.mark.gpu
@pytest.mark.parametrize("world,log_n", [(2, 10), (2, 17), (4, 14)])
def test_ntt_sharded_device_halves_over_gloo(world, log_n):
    """The composed sharded transform (multi_gpu.ntt_sharded: step 1 -> all-to-all -> step 2) with the HIP kernels for both
    local halves; the ranks share cuda:0 and exchange over gloo."""
    _launch(world, "ntt_device", str(log_n))
