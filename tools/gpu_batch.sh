O=gpurun_out/s19; mkdir -p $O
(timeout 900 python tools/fuzz_shapes.py 60 21 0; timeout 900 python tools/fuzz_shapes.py 40 22 1; timeout 900 python tools/fuzz_shapes.py 40 23 1) > $O/fuzz_shapes.txt 2>&1; grep -c ": ok" $O/fuzz_shapes.txt; grep "fuzz:\|MISMATCH\|Error" $O/fuzz_shapes.txt | head
timeout 900 python tools/stress.py 6 1000 3 > $O/stress.txt 2>&1; tail -6 $O/stress.txt | cut -c1-200
for i in 1 2; do (timeout 900 python -m pytest tests -m gpu -q > $O/gputests_$i.log 2>&1); grep "passed\|failed" $O/gputests_$i.log | tail -1; done
