"""bench.py's host-side helpers (no GPU): the reference-algorithm model behind `vs_reference_model`, the plain path's window count,
and the argument surface the driver and the single-process mode rely on."""
import importlib.util

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_reference_model_counts():
    b = _bench()
    # n W 11 + B 370 mulmods at c = 16, W = 16 (1.8e8 + 3.7e8 at config 2), priced at 92.58 G mulmod/s
    n = 1 << 20
    mulmods = n * 16 * 11 + 16 * 65535 * 370
    assert abs(mulmods - (1.845e8 + 3.880e8)) < 2e6
    assert abs(b.reference_model_ms(20) - mulmods / 92.58e9 * 1e3) < 1e-9
    assert 35.0 < b.reference_model_ms(24) < 37.0
    assert b.reference_model_ms(26) > 3.9 * b.reference_model_ms(24) - 20


def test_plain_window_counts():
    """the plain path's window plan comes from the library's own policy (a host-side function: no GPU needed)"""
    sys.path.insert(0, ROOT)
    from panda_amd import gpu_ffi as ffi

    b = _bench()
    lib = ffi.load()
    assert b.plain_windows(lib, 0, 22) == 16  # 254 (+1) bits in 16-bit windows at 2^21 and 2^22 points
    assert b.plain_windows(lib, 0, 24) == 13  # 20-bit windows from 2^24 points on (round 4)
    assert b.plain_windows(lib, 1, 24) == 13  # BLS12-377 Fr: 253 bits
    assert 16 <= b.plain_windows(lib, 0, 12) <= 40
    assert b.MADS_PER_ADDITION[0] == 8 * 162 + 2 * 126 - 81
    assert b.MADS_PER_ADDITION[2] == 8 * 2 * 14 * 14 + 2 * (14 * 15 // 2 + 14 * 14) - 14 * 14
    assert b.MADS_PER_ADDITION[1] == b.MADS_PER_ADDITION[2] - 9 * 14  # BLS12-377 Fq: the modulus' top 29-bit limb is zero, 14 products fewer in each of the 9 reductions


def test_argument_surface(monkeypatch):
    b = _bench()
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = b.parse()
    assert (a.gpus, a.steps, a.warmup, a.log_n) == (1, 10, 3, 24) and not a.single_process
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "5", "--warmup", "2", "--single-process"])
    a = b.parse()
    assert a.gpus == 8 and a.single_process and not a.loopback


def test_launch_ranks_starts_the_launcher_as_a_child(monkeypatch):
    """`python bench.py --gpus 8` typed as the driver types the N = 1 command: the launcher runs as a child process of a parent that has
    not touched the GPU; the child's exit code is the parent's; no launcher (or no line) -> the single-process C-ABI leg, labelled"""
    b = _bench()
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5"])
    args = b.parse()
    calls = []

    def ok(cmd):
        calls.append(cmd)
        return 0, True

    assert b.launch_ranks(args, run=ok) == 0
    (cmd,) = calls
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=8" in cmd and "--master-addr" in cmd and "127.0.0.1" in cmd
    assert cmd[cmd.index(os.path.join(ROOT, "bench.py")) + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    calls.clear()

    def launcher_dies(cmd):
        calls.append(cmd)
        return (7, False) if "torch.distributed.run" in cmd else (0, True)

    assert b.launch_ranks(args, run=launcher_dies) == 0
    assert len(calls) == 2 and "--single-process" in calls[1] and "--launcher-note" in calls[1] and "torch.distributed.run" not in calls[1]
    calls.clear()

    def line_then_failure(cmd):  # a leg failed after the line was printed: the launcher's exit code stands, no second run
        calls.append(cmd)
        return 1, True

    assert b.launch_ranks(args, run=line_then_failure) == 1 and len(calls) == 1
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "os.exec" not in src and "execv" not in src


def test_contract_line_fits_the_drivers_record():
    """every BASELINE configuration's milliseconds sit in one flat block of a line that is well under the 8 KB the driver keeps"""
    import json

    b = _bench()
    small = lambda ms: {"workload": "w" * 200, "with_tables": {"ms_per_step": ms, "k_accumulate_ms": ms / 2, "bases": "b" * 300, "roofline_issue": {"frac": 0.7, "x": "y" * 300}},
                        "registered_only": {"ms_per_step": ms * 1.1}, "vs_reference_model": {"value": 2.0}}
    ntt = {"value": 1e10, "ms": 1.45, "inverse_ms": 1.46, "passes": 3, "radix_bits": [8, 8, 8], "roofline": {"achieved": 700.0, "frac": 0.09}, "roofline_issue": {"frac": 0.7},
           "timing": "t" * 500, "sweep": {f"2^{k}": {"ms": 1.0 * k, "roofline": {"note": "n" * 200}} for k in (20, 22, 26)}}
    full = {"metric": "MSM points/s (BN254, 2^24)", "value": 9.5e8, "unit": "points/s", "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 17.3123456, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": "BN254 MSM 2^24 points per GPU, Jacobian output, bases and scalars resident in HBM", "curve": "bn254", "log_points_per_gpu": 24,
                       "bases": "cached: 12 window tables of 22 bits (12.0 GiB, built in 0.51 s before the timed region)", "sharding": "none", "exchange": "none"},
            "roofline": {"bound": "hbm", "kernel": "k_accumulate", "achieved": 112.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.014, "traffic": 1.68e10,
                         "traffic_source": "profiles/x.csv", "algorithmic_bytes_per_launch": 1610612736, "kernel_ms": 14.3},
            "roofline_issue": {"bound": "valu issue (v_mad_u64_u32)", "kernel": "k_accumulate", "achieved": 20.5, "peak": 28.45, "unit": "T mad lane-ops/s", "frac": 0.72,
                               "valu_model": "v" * 400, "peak_source": "p" * 400},
            "phases_ms": {f"phase{i}": 1.23456789 for i in range(8)}, "device_ms_per_step": 17.2,
            "vs_reference_model": {"value": 2.0, "model": "m" * 600},
            "config2_msm_2_20": small(1.45), "msm_2_22": small(5.0), "config5_bls12_377_2_24_projective": small(34.0), "bn254_g2_msm_2_20": small(6.1),
            "ntt": ntt, "ntt_bls12_377": dict(ntt), "config4_msm_2_26": {"ms_per_step": 67.0, "n_gpus": 1, "log_points_per_gpu": 26, "workload": "w" * 300},
            "without_tables": {"ms_per_step": 19.5, "unregistered": {"ms_per_step": 20.0, "note": "n" * 300}},
            "pcie_inclusive": {"ms": 28.0, "note": "n" * 300, "single_call_pipelined": {"ms": 20.1, "note": "n" * 300}},
            "cpu_baseline": {"value": 55000.0, "unit": "points/s", "cores": 1, "kind": "port", "sample": "s" * 150, "multi_thread": {"value": 7e5}},
            "failed_legs": ["x"], "broken_leg": {"error": "e" * 300}}
    line = b.contract_line(full)
    text = json.dumps(line)
    assert len(text) < 7000, len(text)
    assert list(line)[:12] == ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"]
    assert list(line)[12:14] == ["config", "configs_ms"]
    cm = line["configs_ms"]
    for k in ("c2_msm_2_20_tables", "c2_msm_2_20_registered", "c3_ntt_2_24_fwd", "c3_ntt_2_24_inv", "c4_msm_2_26_total", "c5_bls377_2_24_proj_tables",
              "msm_2_24_registered", "msm_2_24_unregistered", "msm_2_22_tables", "ntt_2_26_fwd", "ntt_bls377_2_24_fwd"):
        assert k in cm, k
    assert cm["c2_msm_2_20_tables"] == 1.45 and line["roofline"]["frac"] == 0.014 and line["cpu_baseline"]["kind"] == "port"
    assert line["broken_leg"]["error"].startswith("e") and line["failed_legs"] == ["x"]
    assert not any(isinstance(v, str) and len(v) > 200 for v in json.loads(text)["config"].values())


@pytest.mark.gpu
def test_bench_starts_its_own_ranks_on_the_gpu_box():
    """`python bench.py --gpus 2 ...` typed WITHOUT a launcher in front (the driver's N = 1 command with another N): one contract line under
    7 KB, two ranks (gloo, both on device 0: the rehearsal of the N > 1 path on a one-GPU box), the sharded legs in configs_ms"""
    import json
    import subprocess

    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--all-on-device0", "--steps", "1", "--warmup", "0",
           "--log-n", "18", "--config4-total-log-n", "19", "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
    assert len(lines[0]) < 7000
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["value"] > 0 and "launcher" not in line
    for k in ("c4_msm_2_26_total", "ntt_sharded_2_24_total", "ntt_sharded_2_24_per_gpu", "c_abi_msm_2_24_per_gpu"):
        assert line["configs_ms"].get(k, 0) > 0, (k, line["configs_ms"])


def test_contract_line_of_a_multi_gpu_record():
    """at N > 1 the flat block carries the strong-scaling legs and the single-process C-ABI leg's figures"""
    import json

    b = _bench()
    full = {"metric": "MSM points/s (BN254, 2^24)", "value": 7.4e9, "unit": "points/s", "n_gpus": 8, "steps": 20, "warmup": 5, "ms_per_step": 18.1, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": "w", "curve": "bn254", "log_points_per_gpu": 24, "bases": "cached", "sharding": "base-range x8", "exchange": "all-gather of 96 B partials (RCCL) + host point additions"},
            "roofline": {"bound": "hbm", "kernel": "k_accumulate", "achieved": 112.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.014, "traffic": None, "kernel_ms": 14.3},
            "config4_msm_2_26": {"ms_per_step": 9.9, "n_gpus": 8, "log_points_per_gpu": 23, "scaling": "strong"},
            "ntt_sharded": {"strong_2_24_total": {"ms": 0.6, "inverse_ms": 0.61}, "weak_2_24_per_gpu": {"ms": 2.4}, "value": 2.8e10},
            "c_abi_single_process": {"ms_per_step": 18.3, "configs_ms": {"c4_msm_2_26_total": 10.1, "ntt_sharded_2_24_total": 0.7, "msm_2_24_from_host_one_call": 21.0}},
            "soft_failed_legs": []}
    line = b.contract_line(full)
    cm = line["configs_ms"]
    assert cm == {"c4_msm_2_26_total": 9.9, "ntt_sharded_2_24_total": 0.6, "ntt_sharded_2_24_total_inv": 0.61, "ntt_sharded_2_24_per_gpu": 2.4, "c_abi_msm_2_24_per_gpu": 18.3,
                  "c_abi_c4_msm_2_26_total": 10.1, "c_abi_ntt_sharded_2_24_total": 0.7, "c_abi_msm_2_24_from_host": 21.0}
    assert line["c4_shape"] == {"n_gpus": 8, "log_points_per_gpu": 23, "scaling": "strong"} and line["n_gpus"] == 8
    assert len(json.dumps(line)) < 3000
