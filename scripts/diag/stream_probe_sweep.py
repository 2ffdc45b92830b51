"""which plain streaming read of Personal_Memory (1.28 GB) gets closest to the guide's 6.0-6.3 TB/s.
Run ON THE GPU BOX; one line per (mode, blocks per CU).  mode = form * 2 + nt (form 0: grid-strided, 1: block-contiguous
4 x 16 B per lane, 2: block-contiguous 8 x 16 B per lane)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import foodrec_amd  # noqa: E402

dev = torch.device("cuda", 0)
U, C, E = 1_000_000, 4, 64
PM = torch.randn((U, C + 1, E), device=dev) / 8
eng = foodrec_amd.ScoringEngine(PM, torch.randn((1000, E), device=dev), torch.randn((C, E), device=dev), device=dev)
sink = torch.zeros(4, device=dev)
nbytes = PM.numel() * 4
for mode in range(6):
    for bpc in (2, 4, 8, 16, 32):
        os.environ["M2D_PROBE_MODE"], os.environ["M2D_PROBE_GRID"] = str(mode), str(bpc)
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(13)]
        for i in range(12):
            evs[i].record()
            eng.stream_read_probe(PM, sink)
        evs[12].record()
        torch.cuda.synchronize()
        ms = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(2, 12))
        print("mode %d (form %d, nt %d) blocks/CU %2d: median %.4f ms  %.0f GB/s  (best %.0f)"
              % (mode, mode // 2, mode & 1, bpc, ms[5], nbytes / ms[5] / 1e6, nbytes / ms[0] / 1e6), flush=True)
