# Run ON THE GPU BOX: the bench lines kept under profiles/ (one JSON line each, gpurun_out/${ROUND:-r04}/lines/).
set -o pipefail
D=gpurun_out/${ROUND:-r05}/lines; mkdir -p $D
python bench.py > $D/bench_default.json 2> $D/bench_default.err; echo default rc=$?
python bench.py --embed 128 --no-cpu-baseline --scaling-users 0 > $D/bench_e128.json 2>/dev/null; echo e128 rc=$?
python bench.py --users 10000000 --dishes 1000000 --no-cpu-baseline --scaling-users 0 > $D/bench_10Musers_1Mdishes.json 2>/dev/null; echo 10M rc=$?
python bench.py --workload ingredients --no-cpu-baseline --scaling-users 0 > $D/bench_ingredients.json 2>/dev/null; echo ing rc=$?
python bench.py --workload mlp --embed 128 --scaling-users 0 > $D/bench_mlp_e128.json 2>/dev/null; echo mlp rc=$?
python bench.py --config 3 --steps 2 --warmup 1 --no-cpu-baseline > $D/bench_config3.json 2>/dev/null; echo config3 rc=$?
python bench.py --config 4 --steps 2 --warmup 1 --no-cpu-baseline > $D/bench_config4.json 2>/dev/null; echo config4 rc=$?
python bench.py --workload topk --no-cpu-baseline > $D/bench_topk_100kdishes_e64.json 2>/dev/null; echo topk rc=$?
python bench.py --workload topk --topk-with-ingredients --no-cpu-baseline > $D/bench_topk_ingredients_e64.json 2>/dev/null; echo topk_ing rc=$?
python bench.py --workload topk --users 64657 --dishes 4548 --embed 200 --no-cpu-baseline > $D/bench_topk_refshape_e200.json 2>/dev/null; echo topk_e200 rc=$?
for K in 10 16; do python bench.py --workload topk --topk-weighted-masks --topk-k $K --no-cpu-baseline > $D/bench_topk_weighted_masks_e64_k$K.json 2>/dev/null; echo weighted k$K rc=$?; done
python bench.py --workload mlp --embed 64 --no-cpu-baseline --scaling-users 0 > $D/bench_mlp_e64.json 2>/dev/null; echo mlp_e64 rc=$?
python bench.py --workload topk --dishes 1000000 --no-cpu-baseline > $D/bench_topk_1Mdishes_e64.json 2>/dev/null; echo topk_1M rc=$?
python bench.py --workload topk --dishes 1000000 --embed 128 --no-cpu-baseline > $D/bench_topk_1Mdishes_e128.json 2>/dev/null; echo topk_1M_e128 rc=$?
python bench.py --workload train --learner sgd --steps 300 > $D/bench_train_sgd_refdefault.json 2>/dev/null; echo sgd rc=$?
python bench.py --workload train --learner adam --steps 300 > $D/bench_train_adam_refdefault.json 2>/dev/null; echo adam rc=$?
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --no-cpu-baseline > $D/bench_dist1.json 2>/dev/null; echo dist1 rc=$?
for f in $D/*.json; do tail -1 $f | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$f'.split('/')[-1], round(d['value']/1e9,3), 'G/s', round(d['ms_per_step'],4), 'ms', r.get('bound'), r.get('frac'))"; done
