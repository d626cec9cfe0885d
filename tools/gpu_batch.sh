O=gpurun_out/s33; mkdir -p $O
for i in 1 2 3; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config5 > $O/b$i.json 2>/dev/null; python -c "
import json;b=json.load(open('$O/b$i.json'));print(b['value'],b['ms_per_step'],b['roofline']['avg_launch_us'],b['decode_128']['tokens_per_s_mean'],b['long_context']['tokens_per_s'],b['parity']['match'])"; done
