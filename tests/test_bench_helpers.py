"""bench.py's host-side helpers (no GPU): the reference-algorithm model behind `vs_reference_model`, the plain path's window count,
and the argument surface the driver and the single-process mode rely on."""
import importlib.util
import os
import sys

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
    assert b.MADS_PER_ADDITION[1] == 8 * 2 * 14 * 14 + 2 * (14 * 15 // 2 + 14 * 14) - 14 * 14


def test_argument_surface(monkeypatch):
    b = _bench()
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = b.parse()
    assert (a.gpus, a.steps, a.warmup, a.log_n) == (1, 10, 3, 24) and not a.single_process
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "5", "--warmup", "2", "--single-process"])
    a = b.parse()
    assert a.gpus == 8 and a.single_process and not a.loopback
