"""CPU-side checks of bench.py's driver-facing contract: flags, defaults, the algorithmic byte model."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("m2d_bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_algorithmic_bytes_follow_survey_8d():
    b = _bench()
    # (C+2)*E*4 + C*4 + 12: SURVEY.md section 8d
    assert [b.algorithmic_bytes_per_pair(4, E) for E in (32, 64, 128, 200)] == [796, 1564, 3100, 4828]
    assert b.HBM_PEAK_GBS == 8000.0


def test_flags_and_defaults(monkeypatch):
    b = _bench()
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = b.parse()
    assert (a.gpus, a.workload, a.users, a.dishes, a.embed, a.pairs) == (1, "pairs", 1_000_000, 100_000, 64, 1 << 22)
    assert a.steps > 0 and a.warmup > 0
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "7", "--warmup", "3"])
    a = b.parse()
    assert (a.gpus, a.steps, a.warmup) == (8, 7, 3)


def test_usable_cores_is_bounded_by_the_affinity_mask():
    b = _bench()
    n = b.usable_cores()
    assert 1 <= n <= len(os.sched_getaffinity(0))
