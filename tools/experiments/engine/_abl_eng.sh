timeout 100 python tools/eng_check.py 1 1 2>&1 | tail -3
FLM_ABL=17 timeout 200 python tools/trace_eng.py 2 1 6 2>&1 | tail -5 | cut -c1-300
FLM_ABL=0 timeout 200 python tools/trace_eng.py 2 1 6 2>&1 | tail -14 | cut -c1-250
for a in 17 0; do echo "== ablate $a"; timeout 200 python tools/kbench.py 4 64 0 $a 2>&1 | grep -E "eng_ffn|graph|  ffn "; done
