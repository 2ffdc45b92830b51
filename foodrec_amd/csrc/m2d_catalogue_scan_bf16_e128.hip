// The split-bf16 scan kernels of width 128 (see m2d_catalogue_scan_bf16_e64.hip).
#define M2D_SCAN_E 128
#include "m2d_catalogue_scan_bf16.hip"
