# Run ON THE GPU BOX: the bench lines kept under profiles/ (one JSON line each, gpurun_out/${ROUND:-r06}/lines/).
# Every line is written by bench.py itself (--out): a file holds the line alone, never a banner a library printed first.
set -o pipefail
D=gpurun_out/${ROUND:-r06}/lines; mkdir -p $D
run() { name=$1; shift; python bench.py "$@" --out $D/bench_$name.json > /dev/null 2> $D/bench_$name.err; echo $name rc=$?; }
run default
run e128 --embed 128 --no-cpu-baseline --scaling-users 0 --no-config-legs
run 10Musers_1Mdishes --users 10000000 --dishes 1000000 --no-cpu-baseline --scaling-users 0 --no-config-legs
run ingredients --workload ingredients --no-cpu-baseline --scaling-users 0
run mlp_e128 --workload mlp --embed 128 --scaling-users 0
run mlp_e64 --workload mlp --embed 64 --no-cpu-baseline --scaling-users 0
run config3 --config 3 --steps 2 --warmup 1 --no-cpu-baseline
run config4 --config 4 --steps 2 --warmup 1 --no-cpu-baseline
run topk_100kdishes_e64 --workload topk --no-cpu-baseline
run topk_ingredients_e64 --workload topk --topk-with-ingredients --no-cpu-baseline
run topk_refshape_e200 --workload topk --users 64657 --dishes 4548 --embed 200 --no-cpu-baseline
for K in 10 16; do run topk_weighted_masks_e64_k$K --workload topk --topk-weighted-masks --topk-k $K --no-cpu-baseline; done
run topk_1Mdishes_e64 --workload topk --dishes 1000000 --no-cpu-baseline
run topk_1Mdishes_e128 --workload topk --dishes 1000000 --embed 128 --no-cpu-baseline
run train_sgd_refdefault --workload train --learner sgd --steps 300
run train_adam_refdefault --workload train --learner adam --steps 300
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 \
       --no-cpu-baseline --out $D/bench_dist1.json > /dev/null 2> $D/bench_dist1.err; echo dist1 rc=$?
for f in $D/*.json; do python -c "
import json,sys; d=json.load(open('$f')); r=d['roofline']
print('$f'.split('/')[-1], round(d['value']/1e9,3), 'G/s', round(d['ms_per_step'],4), 'ms', r.get('bound'), r.get('frac'))"; done
