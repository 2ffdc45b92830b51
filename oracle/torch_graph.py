"""torch-CPU restatement of the reference graph, op for op.  TEST INFRASTRUCTURE ONLY.

This is the `cpu_baseline` leg of bench.py: the reference's TF-CPU path cannot run (TensorFlow is not
in the image), so the baseline is the same sequence of ops on torch's CPU kernels -- row gather x2,
broadcast multiplies that materialise every [B, C, E] temporary, sum over axes (1, 2), true division,
float32 blend (Model_Recommender.py:57-96) -- threaded over all host cores like TF's Eigen pool.
Label it "CPU restatement of reference graph (TF unavailable)", never "TF-CPU".
PARITY UNPINNED for the arithmetic (see m2d_oracle.py); checked against m2d_oracle.inference_f32 in tests.
"""
import torch


def inference(PM: torch.Tensor, RE: torch.Tensor, CE: torch.Tensor, users: torch.Tensor, items: torch.Tensor,
              cats: torch.Tensor, coef: float = 0.99) -> torch.Tensor:
    a = torch.tensor(coef, dtype=torch.float32)                 # :17
    b = torch.tensor(1.0, dtype=torch.float32) - a              # :96, float32
    if cats.dim() == 2:
        cats = cats.unsqueeze(2)                                # [B, C, 1] :32
    User_Memory = PM.index_select(0, users.long())              # :57
    U_high, U_low = User_Memory[:, :1, :], User_Memory[:, 1:, :]  # :59
    Item = RE.index_select(0, items.long()).unsqueeze(1)        # :63-65
    Dish_Category = cats * CE                                   # :67
    category_score = U_high * Dish_Category                     # :71
    s1 = category_score.sum(dim=(1, 2))                         # :75
    n = cats.sum(dim=(1, 2))                                    # :77
    high = s1 / n                                               # :79
    Dish_Memory = cats * U_low                                  # :82
    dish_score = Item * Dish_Memory                             # :86
    s2 = dish_score.sum(dim=(1, 2))                             # :90
    low = s2 / n                                                # :92
    return a * high + b * low                                   # :95-96
