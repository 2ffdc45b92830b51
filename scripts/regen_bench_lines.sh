set -o pipefail
python bench.py > gpurun_out/b_default.json 2> gpurun_out/b_default.err; echo default rc=$?
python bench.py --embed 128 --no-cpu-baseline > gpurun_out/b_e128.json 2>/dev/null; echo e128 rc=$?
python bench.py --users 10000000 --dishes 1000000 --no-cpu-baseline > gpurun_out/b_10M.json 2>/dev/null; echo 10M rc=$?
python bench.py --users 64657 --dishes 4548 --embed 200 --no-cpu-baseline > gpurun_out/b_ref200.json 2>/dev/null; echo ref200 rc=$?
python bench.py --workload ingredients --no-cpu-baseline > gpurun_out/b_ing.json 2>/dev/null; echo ing rc=$?
python bench.py --opt skip_masked=0 --no-cpu-baseline --no-side > gpurun_out/b_noskip.json 2>/dev/null; echo noskip rc=$?
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --no-cpu-baseline > gpurun_out/b_dist1.json 2>/dev/null; echo dist1 rc=$?
for f in default e128 10M ref200 ing noskip dist1; do tail -1 gpurun_out/b_$f.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$f', round(d['value']/1e9,3), round(d['ms_per_step'],4), r.get('bound'), r.get('frac'), r.get('algorithmic_bytes_per_pair'), (r.get('hbm_only') or {}).get('frac_of_spec_peak'), (r.get('no_reuse') or {}).get('frac'))"; done
