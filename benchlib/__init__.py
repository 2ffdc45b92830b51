"""The legs of bench.py, one module per leg (bench.py itself: argument parsing, the headline leg, line assembly).

    common      constants, SURVEY.md 8d's byte model, synthetic inputs, HIP-event timing
    cli         flags, the self-launch of N ranks, the CPU dry run
    baselines   cpu_baseline / mlp_baseline: the oracle timed on the host cores (the ONLY importers of oracle/)
    pairs       side legs of the pair path: no-reuse, stream probe, HBM-only estimate, user_high, ingredient table
    topk        full-catalogue top-k side leg and its MFMA roofline
    mlp         the build-defined MLP head: roofline of the timed step, BASELINE configs[2] as a side leg
    sharded     user-sharded legs: top-k + all-gather, routed pairs, scaling_path, world identity
    evaluator   evaluate.py's loop beside the one-launch device evaluator
    train       --workload train
    line        flattening of nested legs into the top-level scalars a key-only record keeps
"""
