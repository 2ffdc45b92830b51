"""Twenty fixed-seed cases of each randomized soak (tests/soak_*.py) under `pytest -m gpu`: the soaks draw their shapes, tables
and option forms from a seed -- the clock's when run by hand, these when collected -- so the round-end run exercises the same
generators that found the round-3 bugs (the LDS ring race, the 73-range merge, the batch-dependent repair tier)."""
import os
import runpy
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("script,cases,seed", [
    ("soak_topk_plan.py", 20, 20261004),
    ("soak_random_cases.py", 24, 20261005),             # four of each of its six kinds
    ("soak_mlp_random_cases.py", 20, 20261006),
    ("soak_train_contention.py", 20, 20261007),
])
def test_fixed_seed_cases_of_the_soaks(script, cases, seed, monkeypatch):
    monkeypatch.setattr(sys, "argv", [script, str(cases), str(seed)])
    runpy.run_path(os.path.join(HERE, script), run_name="__main__")
