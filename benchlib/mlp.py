"""The build-defined 3-layer head (BASELINE configs[2]; no reference counterpart): synthetic head, MFMA roofline."""
from __future__ import annotations

from .common import (BF16_MFMA_PEAK_TFLOPS, F32_MFMA_PEAK_TFLOPS, HBM_PEAK_GBS, bare_loop_fields)

H1, H2 = 256, 64                                        # widths of the head: (C + 1) E -> 256 -> 64 -> 1


def synthetic_head(torch, K, dev, gen):
    def rn(*shape):
        return torch.randn(shape, generator=gen, device=dev)
    return (rn(K, H1) / K ** 0.5, rn(H1) * 0.1, rn(H1, H2) / 16.0, rn(H2) * 0.1, rn(H2) / 8.0, 0.0)


def mlp_roofline(torch, eng, kernel_used, dish_cats, items, C, E, B, avg_ms):
    """Split-bf16 form: layers 1-2 run as 3 bf16 MFMAs per product (executed flops = 3 x algorithmic) against the
    dense bf16 peak; the exact form runs everything on the f32 MFMA against its peak.  The producer / consumer kernel
    groups the pairs by dish mask pattern and runs only the k-blocks a pattern keeps (the E k-values of a category
    of weight 0 are zeros in z): executed flops and fetched bytes count those blocks only."""
    K = (C + 1) * E
    fl = 2.0 * (K * H1 + H1 * H2 + H2)
    sec = avg_ms * 1e-3
    tf = fl * B / sec / 1e12
    x3 = kernel_used.endswith("bf16x3")
    pc = kernel_used.startswith("m2d_mlp_pc")
    grouped = pc and eng.get_option("skip_masked") != 0 and E >= 64
    act = float((dish_cats[items.long()] != 0).sum(1).float().mean().item()) if grouped else float(C)
    Ka = (1.0 + act) * E                                          # k-values of layer 1 actually multiplied, per pair
    ex = 3.0 * 2.0 * (Ka * H1 + H1 * H2) * B / sec / 1e12 if x3 else tf
    dense_ex = 3.0 * 2.0 * (K * H1 + H1 * H2) * B / sec / 1e12 if x3 else tf
    peak = BF16_MFMA_PEAK_TFLOPS if x3 else F32_MFMA_PEAK_TFLOPS
    hbm = (2 * Ka * 4 + 12) * B / sec / 1e9
    shape = "v_mfma_f32_16x16x32_bf16" if pc else "v_mfma_f32_32x32x16_bf16"
    roof = {"bound": "mfma", "achieved": ex, "peak": peak, "unit": "TFLOP/s", "frac": ex / peak,
            "mean_active_categories": act, "k_values_multiplied_per_pair": Ka,
            "dense_equivalent_frac": dense_ex / peak, "traffic": None, "kernel_avg_ms": avg_ms,
            "flop_per_pair": fl, "pairs_per_launch": B, "algorithmic_tflops": tf,
            "f32_mfma_equivalent_frac": tf / F32_MFMA_PEAK_TFLOPS,
            "dtype": ("split bf16 for layers 1-2 (3 x %s per product, fp32 accumulate)" % shape if x3
                      else "f32 (v_mfma_f32_32x32x2_f32, exact)"),
            "hbm_algorithmic_GBps": hbm, "hbm_frac": hbm / HBM_PEAK_GBS}
    if x3:
        roof.update(bare_loop_fields(ex))
    return roof, ("bf16x3" if x3 else "f32")
