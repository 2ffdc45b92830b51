// The split-bf16 scan kernels of width 64 (m2d_catalogue_scan_bf16.hip holds the code; two units so that the build's longest
// compile runs as two), and the launcher the other units call.
#define M2D_SCAN_E 64
#include "m2d_catalogue_scan_bf16.hip"

M2D_INTERNAL int m2d_topk_scan_bf16_launch_e128(m2d_engine *h, const GroupedArgs &a, const ScanShape &s, dim3 grid, size_t lds, hipStream_t st);

int m2d_topk_scan_bf16_launch(m2d_engine *h, const GroupedArgs &a, const ScanShape &s, dim3 grid, size_t lds, hipStream_t st)
{
    return s.E == 64 ? m2d_topk_scan_bf16_launch_e64(h, a, s, grid, lds, st) : m2d_topk_scan_bf16_launch_e128(h, a, s, grid, lds, st);
}
