O=gpurun_out/s39; mkdir -p $O
for pos in 40 70 100 130; do echo "== pos $pos" >> $O/split_from.log; timeout 300 python tools/back_bench.py 32 $pos int8 "tuning=1;attn_split=4;attn_split=1;attn_split=4;attn_split=1" 2>&1 | tail -5 >> $O/split_from.log; done; cat $O/split_from.log
