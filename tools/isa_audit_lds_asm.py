"""ISA audit for the inline-assembly LDS reads (flm_attn.h: FLM_RD4 -- ds_read_b128 groups the compiler does not track): between a group of four reads and the next
s_waitcnt lgkmcnt no instruction may touch the destination registers (a copy or a spill of a register whose data has not landed would carry stale bits).
llvm-objcopy --dump-section .hip_fatbin=fat.bin <unit>.o; clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=fat.bin --output=u.co;
llvm-objdump -d u.co > u.s; python tools/isa_audit_lds_asm.py u.s      (round 6: 3842 groups over the six units, none touched)"""
import re, sys
lines = [l.rstrip() for l in open(sys.argv[1])]
ins = []
for i, l in enumerate(lines):
    m = re.match(r"\s+(\S+)\s+(.*?)\s*//", l)
    if m: ins.append((i, m.group(1), m.group(2)))
def regs(tok):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(1): out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
        else: out.add(int(m.group(3)))
    return out
groups = bad = 0
k = 0
while k + 3 < len(ins):
    if all(ins[k + j][1] == "ds_read_b128" for j in range(4)):
        ops = [ins[k + j][2] for j in range(4)]
        addr = [o.split(",")[1].strip().split()[0] for o in ops]
        offs = [re.search(r"offset:(\d+)", o) for o in ops]
        offs = [int(m.group(1)) if m else 0 for m in offs]
        if len(set(addr)) == 1 and offs == [0, 16, 32, 48]:
            groups += 1
            dst = set()
            for o in ops: dst |= regs(o.split(",")[0])
            j = k + 4
            while j < len(ins) and not (ins[j][1] == "s_waitcnt" and "lgkmcnt" in ins[j][2]):
                if ins[j][1] in ("s_cbranch_scc0", "s_cbranch_scc1", "s_branch", "s_endpgm"): break
                r = regs(ins[j][2])
                if r & dst and not (ins[j][1] == "ds_read_b128"):
                    bad += 1; print("TOUCHED before wait:", lines[ins[j][0]].strip()[:100], "| dst", sorted(dst)[:4], "...")
                j += 1
            k += 4; continue
    k += 1
print("asm read groups", groups, "touched", bad)
