set -x
mkdir -p gpurun_out/r04
cp foodrec_amd/libm2d.so /tmp/new.so
cp build/r3bounds/libm2d.so foodrec_amd/libm2d.so
timeout -k 10 500 python -m pytest tests/test_gpu_prune_adversarial.py -q -m gpu -p no:cacheprovider > gpurun_out/r04/adv_oldlib.log 2>&1; echo "old rc $?" >> gpurun_out/r04/adv_oldlib.log
cp /tmp/new.so foodrec_amd/libm2d.so
timeout -k 10 500 python -m pytest tests/test_gpu_prune_adversarial.py -q -m gpu -p no:cacheprovider > gpurun_out/r04/adv_newlib.log 2>&1; echo "new rc $?" >> gpurun_out/r04/adv_newlib.log
tail -5 gpurun_out/r04/adv_oldlib.log; tail -5 gpurun_out/r04/adv_newlib.log
