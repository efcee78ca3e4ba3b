"""The PANDA_MULTI_RCCL transport of csrc/multi_gpu.hip with MORE THAN ONE RANK, on a one-GPU box.

The RCCL call sites (ncclCommInitAll over several ranks, the in-place grouped ncclAllGather of the MSM partials, the n x n grouped
ncclSend / ncclRecv all-to-all of the sharded NTT, the two-stream batch schedule) are replaced by device copies on the loopback
transport, so until an 8-GPU node runs them they were covered by review only.  tests/fake_rccl/libfake_rccl.so is a test-only
stand-in for those eight RCCL entry points: it validates every call the way single-process multi-communicator RCCL needs it
(communicator / device / stream / buffer ownership, group discipline, in-place rule, send <-> receive matching) and then performs
the transfer with RCCL's stream semantics.  It is preloaded into a FRESH child process (the parent has touched the GPU and is
never re-executed); the product library keeps its -lrccl and does not know the stand-in.  There is no reference counterpart: the
reference is single-GPU (src/gpu_manager/wrapper.rs:38, src/cuda/core/unit/msm/msm_cuda.cuh:554-555).
"""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
FAKE_DIR = os.path.join(HERE, "fake_rccl")
FAKE = os.path.join(FAKE_DIR, "libfake_rccl.so")
RCCL_ENTRY_POINTS = ["ncclCommInitAll", "ncclCommDestroy", "ncclGroupStart", "ncclGroupEnd", "ncclAllGather", "ncclSend", "ncclRecv", "ncclGetErrorString"]


def _child_env(log_path):
    env = dict(os.environ)
    env["LD_PRELOAD"] = FAKE + (":" + env["LD_PRELOAD"] if env.get("LD_PRELOAD") else "")
    env["PANDA_TEST_SHARED_DEVICE_RCCL"] = "1"   # csrc/multi_gpu.hip: several RCCL ranks may name one device
    env["FAKE_RCCL_ALLOW_SHARED_DEVICE"] = "1"   # the stand-in's own duplicate-device check (real RCCL refuses duplicates)
    env["FAKE_RCCL_LOG"] = log_path
    return env


def test_interposer_covers_every_rccl_symbol_the_library_references():
    """CPU check: the stand-in is built, exports exactly the RCCL entry points libpanda-cuda.so leaves undefined, and the product
    library does NOT depend on it (DT_NEEDED names librccl, not the stand-in)."""
    assert os.path.exists(FAKE), "build it with make -C tests/fake_rccl (or __graft_entry__.build())"
    lib_path = os.path.join(ROOT, "panda_amd", "csrc", "libpanda-cuda.so")
    undefined = subprocess.run(["nm", "-D", "--undefined-only", lib_path], capture_output=True, text=True, check=True).stdout
    wanted = sorted(line.split()[-1] for line in undefined.splitlines() if line.split()[-1].startswith("nccl"))
    assert wanted == sorted(RCCL_ENTRY_POINTS), wanted
    defined = subprocess.run(["nm", "-D", "--defined-only", FAKE], capture_output=True, text=True, check=True).stdout
    have = {line.split()[-1] for line in defined.splitlines()}
    assert set(RCCL_ENTRY_POINTS) <= have, set(RCCL_ENTRY_POINTS) - have
    needed = subprocess.run(["readelf", "-d", lib_path], capture_output=True, text=True, check=True).stdout
    assert "librccl" in needed and "fake_rccl" not in needed


@pytest.mark.gpu
def test_rccl_transport_with_2_4_8_ranks_under_the_interposer(tmp_path):
    """MSM, MSM-from-host, NTT, NTT batch, BLS12-377 / BLS12-381 NTT, BLS12-381 and G2 MSM through panda_*_multi with the RCCL transport
    at 2 / 4 / 8 ranks, each against the oracle; per case the interposer must have matched ranks x ranks send/receive pairs per exchange
    and seen every rank in every all-gather, with no validation failure.  Then the interposer's own refusals (tests/fake_rccl/run_cases.py)."""
    log = str(tmp_path / "fake_rccl.log")
    r = subprocess.run([sys.executable, os.path.join(FAKE_DIR, "run_cases.py")], env=_child_env(log), capture_output=True, text=True, timeout=900)
    tail = r.stdout[-6000:] + "\n" + r.stderr[-6000:]
    assert r.returncode == 0, tail
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("FAKE_RCCL_RESULT ")]
    assert line, tail
    res = json.loads(line[-1][len("FAKE_RCCL_RESULT "):])
    ranks_seen = {c["ranks"] for c in res["cases"]}
    assert ranks_seen == {2, 4, 8}
    for c in res["cases"]:
        assert c["matched_pairs"] == c["exchanges"] * c["ranks"] ** 2, c
    kinds = {c["case"].split("[")[0] for c in res["cases"]}
    assert {"msm", "msm_from_host", "ntt", "ntt_batch", "msm_other_curves", "sharded_over", "two_handles_two_threads"} <= kinds
    assert res["product_totals"]["failures"] == 0 and res["product_totals"]["matched_pairs"] > 0 and res["product_totals"]["allgathers"] > 0
    assert len(res["refused"]) >= 14
    text = open(log).read()
    assert "FAIL" in text  # the refusals of the second half are logged ...
    first_fail = text.index("FAIL")
    assert "8 rank(s) on device(s) 0,0,0,0,0,0,0,0" in text[:first_fail]  # ... after the product cases, which logged none
    assert "64 matched send/recv pair(s)" in text[:first_fail]
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):  # keep the validation log of this run next to the other GPU-box outputs
        with open(os.path.join(out_dir, "fake_rccl_validation.log"), "w") as f:
            f.write(text)
        with open(os.path.join(out_dir, "fake_rccl_result.json"), "w") as f:
            json.dump(res, f, indent=1)


@pytest.mark.gpu
def test_cpp_manager_test_over_rccl_under_the_interposer(tmp_path):
    """The C++ gpu_manager mirror's integration binary once with its RCCL pass at four ranks on device 0 and PandaMultiGpuManager on the
    RCCL transport (MANAGER_TEST_RCCL_RANKS_ON_DEVICE0), the stand-in preloaded."""
    exe = os.path.join(ROOT, "panda_amd", "csrc", "tests", "manager_test")
    assert os.path.exists(exe), "build it with make -C panda_amd/csrc"
    log = str(tmp_path / "fake_rccl_manager.log")
    env = _child_env(log)
    env["MANAGER_TEST_RCCL_RANKS_ON_DEVICE0"] = "4"
    r = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "manager_test: all ok" in r.stdout and "4 rank(s), RCCL transport" in r.stdout
    text = open(log).read()
    assert "FAIL" not in text and "4 rank(s) on device(s) 0,0,0,0" in text and "16 matched send/recv pair(s)" in text and "2 rank(s) on device(s) 0,0" in text


@pytest.mark.gpu
def test_bench_single_process_leg_over_rccl_under_the_interposer(tmp_path):
    """`bench.py --gpus 4 --single-process` -- the C-ABI leg the driver's N > 1 runs end with (one process, panda_*_multi over RCCL) -- rehearsed with
    four ranks on device 0 over the RCCL transport under the interposer: one contract line with the sharded MSM, config 4's shape and the sharded NTT legs."""
    log = str(tmp_path / "fake_rccl_bench.log")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--single-process", "--rccl-on-device0", "--steps", "1", "--warmup", "0", "--log-n", "18",
           "--config4-total-log-n", "20", "--detail", str(tmp_path / "detail.json")]
    env = {k: v for k, v in _child_env(log).items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
    line = json.loads(lines[0])
    assert line["n_gpus"] == 4 and line["value"] > 0 and "interposer" in line["config"]["exchange"]
    ms = line["configs_ms"]
    assert ms["c4_msm_2_26_total"] > 0 and ms["ntt_sharded_2_24_total"] > 0 and ms["ntt_sharded_batch4_per_transform"] > 0 and ms["msm_2_24_from_host_one_call"] > 0
    text = open(log).read()
    assert "FAIL" not in text and "4 rank(s) on device(s) 0,0,0,0" in text and "16 matched send/recv pair(s)" in text and "all-gather(s) over 4 rank-call(s)" in text
