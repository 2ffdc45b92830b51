mkdir -p gpurun_out/ab
export PROBE_VARIANTS=""
(bash scripts/diag/ab_probe.sh foodrec_amd/libm2d_old.so 100000 64 65536; bash scripts/diag/ab_probe.sh foodrec_amd/libm2d_old.so 100000 64 262144;  bash scripts/diag/ab_probe.sh foodrec_amd/libm2d_old.so 100000 64 16384) > gpurun_out/ab/cmp3.txt 2>&1
grep -A2 "==" gpurun_out/ab/cmp3.txt | grep "==\|prune=1"
