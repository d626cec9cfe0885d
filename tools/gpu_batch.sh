mkdir -p gpurun_out/s10
(timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/s10/gputests.log 2>&1; echo rc=$? >> gpurun_out/s10/gputests.log); grep -n "passed\|failed\|^FAILED\|^E  " gpurun_out/s10/gputests.log | head -20
python bench.py --steps 20 --warmup 5 > gpurun_out/s10/bench20.json 2> gpurun_out/s10/bench20.err; cut -c1-260 gpurun_out/s10/bench20.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/s10/bench20.json').read())
print('decode_128', d.get('decode_128',{}).get('tokens_per_s_mean'), d.get('decode_128',{}).get('token_roofline_frac')); print('roofline', d['roofline']['frac'], d['roofline']['avg_launch_us'], d['roofline'].get('us_per_layer')); print('long', d.get('long_context',{}).get('tokens_per_s'), d.get('long_context',{}).get('token_roofline_frac')); print('prefill', d.get('prefill',{}).get('ms'), d.get('config5_prefill512_int16',{}).get('ms')); print('parity', d['parity'].get('match'), d.get('token_path'))
PY
