"""foodrec_amd -- MI355X (gfx950) scoring engine for the Market2Dish recommender forward pass.

Scope: ``Code/Recommender/Model_Recommender.py:56-97`` (``Model.inference``) and its caller
``Code/Recommender/evaluate.py`` of WenjieWWJ/FoodRec, behind the reference's own model-build /
predict / evaluate surface.  The arithmetic lives in ``libm2d.so`` (hand-written HIP, C ABI in
``include/m2d.h``); importing this package without that library raises.
"""
from . import _native

_native.lib()          # fail loudly, at import, if the HIP library has not been built

from .ops import ScoringEngine                                  # noqa: E402
from .recommender import Model, Session                         # noqa: E402
from .evaluator import clear_eval_plans, evaluate_model, eval_one_rating, getHitRatio, getNDCG   # noqa: E402
from .formats import Dataset                                    # noqa: E402

__all__ = ["ScoringEngine", "Model", "Session", "evaluate_model", "eval_one_rating", "getHitRatio", "getNDCG",
           "Dataset", "clear_eval_plans"]
