O=gpurun_out/s23; mkdir -p $O
for i in 1 2 3; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config5 > $O/b$i.json 2>/dev/null; python -c "
import json;b=json.load(open('$O/b$i.json'));print(b['value'],b['ms_per_step'],b['roofline']['avg_launch_us'],b['decode_128']['tokens_per_s_mean'],b['long_context']['tokens_per_s'],b['parity']['match'])"; done
(timeout 900 python -m pytest tests -m gpu -q > $O/gputests.log 2>&1); grep "passed\|failed" $O/gputests.log | tail -1
python bench.py --steps 20 --warmup 5 > $O/r06_bench_n1_steps20.json 2>/dev/null; cut -c1-100 $O/r06_bench_n1_steps20.json
