mkdir -p gpurun_out/r4
python -m pytest tests -x -q -m gpu > gpurun_out/r4/gputest3.txt 2>&1; tail -6 gpurun_out/r4/gputest3.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4/bench_a.json 2> gpurun_out/r4/bench_a.err
python -c "
import json; d=json.loads(open('gpurun_out/r4/bench_a.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('steady_state_200'), d['parity_check']['max_rel_elbo'], d['parity_check']['label_flips'], d['dtype'])"
