#!/usr/bin/env python3
"""Instruction mix of a kernel's main loop from hipcc's -save-temps assembly.

    tools/isa_loop_stats.py <file.s> <kernel-name-substring> [--reserved N]

Prints, for the outermost loop that contains the most instructions (the demodulator's main loop), the number of
VALU / SALU / LDS / VMEM / branch / waitcnt instructions, scratch accesses and v_readlane/v_writelane (SGPR spills),
split into compiler-generated code and inline-asm blocks.  With --reserved N it also fails (exit 1) when compiler-
generated code touches a VGPR >= N: the register partition of the rotating-window kernel (demod_kernel_rot.hip)."""
import re, sys

def classify(op):
    if op.startswith("v_readlane") or op.startswith("v_writelane"): return "lane"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_cbranch") or op.startswith("s_branch") or op.startswith("s_setpc"): return "branch"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith("scratch_"): return "scratch"
    if op.startswith("global_") or op.startswith("flat_") or op.startswith("buffer_"): return "vmem"
    return "other"

def main():
    path, name = sys.argv[1], sys.argv[2]
    reserved = int(sys.argv[sys.argv.index("--reserved") + 1]) if "--reserved" in sys.argv else None
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and name in l and l.rstrip().endswith(":") or (name in l and "; @" in l and l.startswith("_Z")))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start:end]
    inasm = False
    bad = []
    tot = {}
    for i, l in enumerate(body):
        if "#ASMSTART" in l: inasm = True; continue
        if "#ASMEND" in l: inasm = False; continue
        t = l.strip()
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"): continue
        op = t.split()[0]
        k = ("asm:" if inasm else "") + classify(op)
        tot[k] = tot.get(k, 0) + 1
        if reserved is not None and not inasm:
            regs = [int(m.group(1)) for m in re.finditer(r"\bv(\d+)\b", t)] + [int(m.group(2)) for m in re.finditer(r"\bv\[(\d+):(\d+)\]", t)]
            if any(r >= reserved for r in regs): bad.append((start + i + 1, t))
    print("whole kernel:", dict(sorted(tot.items())))
    # main loop = the depth-1 loop with the most lines between its header and its last back edge
    best = None
    for i, l in enumerate(body):
        if "Loop Header: Depth=1" in l:
            j = i
            while not re.match(r"^\.LBB\d+_\d+:", body[j]): j -= 1
            lab = body[j].split(":")[0]
            last = max((k for k, t in enumerate(body) if re.search(r"s_c?branch\S*\s+" + re.escape(lab) + r"\b", t)), default=j)
            if best is None or last - j > best[1] - best[0]: best = (j, last, lab)
    if best:
        j, last, lab = best
        inasm = False; lt = {}
        for t in body[j:last + 1]:
            if "#ASMSTART" in t: inasm = True; continue
            if "#ASMEND" in t: inasm = False; continue
            t = t.strip()
            if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"): continue
            k = ("asm:" if inasm else "") + classify(t.split()[0])
            lt[k] = lt.get(k, 0) + 1
        print("main loop %s (static, all paths, all 10 copies of the asm):" % lab, dict(sorted(lt.items())))
    for m in ("NumVgprs", "ScratchSize", "Occupancy"):
        for l in lines[end:end + 60]:
            if m in l: print(l.strip("; ")); break
    if bad:
        print("compiler-generated code touches reserved VGPRs:")
        for b in bad[:10]: print("  line %d: %s" % b)
        sys.exit(1)

if __name__ == "__main__":
    main()
