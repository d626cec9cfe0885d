O=gpurun_out/s41; mkdir -p $O
(timeout 1500 python -m pytest tests -m gpu -q -x > $O/gputests.log 2>&1); grep "passed\|failed" $O/gputests.log | tail -1
for pos in 14 100 300 516 900; do echo "== pos $pos" >> $O/chains.log; timeout 300 python tools/back_bench.py 32 $pos int8 "tuning=1;tuning=1" 2>&1 | tail -2 >> $O/chains.log; done; cat $O/chains.log
for i in 1 2; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config5 > $O/b$i.json 2>/dev/null; python -c "
import json;b=json.load(open('$O/b$i.json'));print(b['value'],b['ms_per_step'],b['roofline']['avg_launch_us'],b['decode_128']['tokens_per_s_mean'],b['long_context']['tokens_per_s'],b['parity']['match'])"; done
