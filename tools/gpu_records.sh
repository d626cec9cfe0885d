#!/bin/bash
# round 6's records in one gpurun call: everything lands in gpurun_out/r06/ and is copied into profiles/ by hand
O=gpurun_out/r06; mkdir -p $O
(timeout 1500 python -m pytest tests -m gpu -q --durations=10 > $O/r06_gputests.log 2>&1; echo rc=$? >> $O/r06_gputests.log); grep -n "passed\|failed" $O/r06_gputests.log | tail -2
python bench.py --steps 20 --warmup 5 > $O/r06_bench_n1_steps20.json 2> $O/bench20.err; cut -c1-120 $O/r06_bench_n1_steps20.json
python bench.py > $O/r06_bench_n1.json 2> $O/bench128.err; cut -c1-120 $O/r06_bench_n1.json
bash tools/prof_bench.sh --steps 20 --warmup 5 > $O/prof_bench.txt 2>&1; cp gpurun_out/prof_bench/run_kernel_stats.csv $O/r06_bench_kernel_stats.csv; cp gpurun_out/prof_bench/bench.json $O/r06_bench_under_rocprof.json; head -8 $O/prof_bench.txt
bash tools/pmc_bench.sh > $O/pmc_bench.txt 2>&1; cp gpurun_out/pmc/pmc_fetch_write_raw.json $O/r06_pmc_fetch_write_raw.json; tail -6 $O/pmc_bench.txt
for n in 2 4; do FLM_BENCH_FORCE_DEVICE=0 GPU_MAX_HW_QUEUES=16 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2960$n bench.py --gpus $n --steps 10 --warmup 3 --no-cpu-baseline 2> $O/tp$n.err | grep '^{"metric' > $O/r06_bench_tp${n}_one_gpu.json; cut -c1-120 $O/r06_bench_tp${n}_one_gpu.json; done
for w in 2 4 8; do GPU_MAX_HW_QUEUES=16 timeout 400 python tools/tp_onegpu.py $w 4 64 2>&1 | grep -v Warning; done > $O/r06_tp_onegpu.txt; cut -c1-170 $O/r06_tp_onegpu.txt
(FLM_GPU_LIB=fast-llama_amd/lib/var/libflm_abl.so python tools/trace_back.py 4 14 "" 103; FLM_GPU_LIB=fast-llama_amd/lib/var/libflm_abl.so python tools/trace_back.py 4 14 "tuning=1,gr_edges=0" 103; FLM_GPU_LIB=fast-llama_amd/lib/var/libflm_abl.so python tools/trace_back.py 4 516 "" 103) > $O/r06_layer_timelines.txt 2>&1; grep -c "" $O/r06_layer_timelines.txt
python bench.py --steps 20 --warmup 5 --quant int16 --no-cpu-baseline > $O/r06_bench_int16.json 2>/dev/null; cut -c1-120 $O/r06_bench_int16.json
python bench.py --steps 20 --warmup 5 --shape 1.3B --no-cpu-baseline > $O/r06_bench_1p3B.json 2>/dev/null; cut -c1-120 $O/r06_bench_1p3B.json
python bench.py --steps 20 --warmup 5 --pos 512 --no-cpu-baseline > $O/r06_bench_pos512.json 2>/dev/null; cut -c1-120 $O/r06_bench_pos512.json
(timeout 900 python tools/soak32.py 1000 2; timeout 600 python tools/stress2.py 200; timeout 600 python tools/fuzz_shapes.py 30 7 0; timeout 600 python tools/fuzz_shapes.py 12 9 1) > $O/r06_fuzz_stress.txt 2>&1; tail -12 $O/r06_fuzz_stress.txt
bash tools/pmc_prefill.sh > $O/pmc_prefill.txt 2>&1; tail -4 $O/pmc_prefill.txt
python3 tools/pmc_agg.py gpurun_out/pmc/p1 gpurun_out/pmc/p2 gpurun_out/pmc/p3 > gpurun_out/pmc/prefill_counters.json 2>/dev/null
python3 - > $O/r06_prefill_gemm_pmc.txt <<'PY'
import json
d = json.load(open("gpurun_out/pmc/prefill_counters.json"))
print("== prefill GEMM tiles (k_gemm_q8_mfma), round 6: SQ counters per launch, rocprofv3 --pmc (three separate passes, --kernel-trace only beside them) over tools/prefill_bench.py 4 512 ==")
print("(4 layers of 7B width, 512-token prompt; mean per launch over the run's launches of each instantiation)")
keys = ["SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU",
        "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_ADDR_CONFLICT", "SQ_LDS_UNALIGNED_STALL", "SQ_INSTS_VMEM_RD", "SQ_ACTIVE_INST_VMEM", "SQ_INST_CYCLES_VMEM", "SQ_INSTS_SALU"]
for k, v in sorted(d.items()):
    if "k_gemm_q8_mfma" not in k and "k_gemm_q16_mfma" not in k: continue
    print("\n" + k[:110] + f"   (n = {max(e['n'] for e in v.values())})")
    for c in keys:
        if c in v: print(f"  {c:28s} {v[c]['mean']:.4g}")
    g = lambda c: v.get(c, {}).get("mean", 0.0)
    if g("SQ_INSTS_MFMA") and g("SQ_BUSY_CYCLES"):
        print(f"  -> matrix pipe busy {g('SQ_VALU_MFMA_BUSY_CYCLES') / max(g('SQ_BUSY_CYCLES'), 1) * 100:.1f} % of SQ busy cycles (x 4 SIMDs per SQ counted together); MFMA busy cycles per MFMA {g('SQ_VALU_MFMA_BUSY_CYCLES') / g('SQ_INSTS_MFMA'):.1f}")
        print(f"  -> VALU instructions per MFMA {g('SQ_INSTS_VALU') / g('SQ_INSTS_MFMA'):.1f}, LDS instructions per MFMA {g('SQ_INSTS_LDS') / g('SQ_INSTS_MFMA'):.2f}, LDS_IDX_ACTIVE per MFMA {g('SQ_LDS_IDX_ACTIVE') / g('SQ_INSTS_MFMA'):.1f}, bank-conflict cycles per LDS instruction {g('SQ_LDS_BANK_CONFLICT') / max(g('SQ_INSTS_LDS'), 1):.3f}")
        if g("SQ_WAVE_CYCLES"): print(f"  -> of a wave's cycles: issuing {g('SQ_ACTIVE_INST_ANY') / g('SQ_WAVE_CYCLES') * 100:.0f} %, waiting on an instruction {g('SQ_WAIT_INST_ANY') / g('SQ_WAVE_CYCLES') * 100:.0f} %, waiting otherwise {g('SQ_WAIT_ANY') / g('SQ_WAVE_CYCLES') * 100:.0f} %; LDS waits {g('SQ_WAIT_INST_LDS') / g('SQ_WAVE_CYCLES') * 100:.0f} %")
PY
head -30 $O/r06_prefill_gemm_pmc.txt
python __graft_entry__.py --smoke > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
