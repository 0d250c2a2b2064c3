#!/usr/bin/env python3
"""what the NON-FIR VALU instructions of a firing of the configs[1] kernel are, stage by stage, priced with the
measured per-instruction SIMD costs (tools/valu_cost.py, profiles/r04_valu_mix_ubench.jsonl), and which of them are not the
reference's own arithmetic.  Works on the compiler-generated instructions of the firing path of the built kernel
(meteor_demod_amd/lib/demod_kernel_rot.gfx950.s, kept by build.py): from the label behind the slide to the end of the firing.
The stage map below is BY POSITION in that path for the round-6 build; the script checks the instruction count and the first
opcode of every stage and refuses to print a table for a build it was not written against.
usage: price_scalar_stage.py [pipe_cycles_per_wave_firing=1559]  ->  markdown"""
import re, sys, collections
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))
import valu_cost as V

PIPE = float(sys.argv[1]) if len(sys.argv) > 1 else 1559.0
KERNEL = r"^_ZN\S*demod_kernel_rotILi16ELi0ELi14ELi0ELi1E\S*:"
R, B = "reference arithmetic", "bookkeeping"
# (first index, last index, what, class, reference lines)
STAGES = [
    ((0, 3), (5, 5), (7, 7), "FIR padding votes: smallest / largest alignment of the wave (4 compares, result to a scalar)", B, "- (buys 6.7 %: profiles/r06_fir_padding.json)"),
    ((4, 4), "coefficient row address (alignment, bank)", B, "filter.c:48 `interp - 1 - i`"),
    ((6, 6), (8, 9), "accumulator := 0, window rotation to a scalar", B, "filter.c:51"),
    ((10, 19), "AGC bias and gain", R, "agc.c:13-20"),
    ((20, 34), "cabsf: f64 squares, v_rsq_f64 + one exact Newton step, boundary test", R, "agc.c:21 (libm hypot)"),
    ((35, 44), "turn codes of fast_sin / fast_cos (f64)", R, "sincos.c:13-26, 37-40"),
    ((45, 47), "sine-table addresses", R, "sincos.c:27-31 (18 integer instructions each, as a table)"),
    ((48, 49), "only the last symbol fired inside one input sample survives", B, "demod.c:33-47"),
    ((50, 55), "sign and scale of the table values", R, "sincos.c:29-33"),
    ((56, 56), (59, 59), "end-of-block test of the lane", B, "main.c:303 `for`"),
    ((57, 58), (60, 60), (120, 122), "mixer, both rails", R, "pll.c:56-57"),
    ((61, 77), "timing error, loop filter, clamp of the clock word", R, "timing.c:60-87"),
    ((78, 93), (95, 95), (105, 105), "symbol clock: 19 adds (16 blind + 3 checked)", R, "timing.c:32-38, 16 times per symbol"),
    ((94, 94), (96, 104), (106, 119), "which of the checked steps fired; sample position of the next firing (one 32-bit division by -O)", R, "timing.c:36 (16 compares per symbol) + main.c's sample counter"),
    ((123, 126), (129, 130), (133, 134), (136, 138), "Costas error: tanh table look-ups, products, alpha e", R, "pll.c:100-111, 143-159"),
    ((127, 128), (131, 132), (135, 135), "NCO advance and its wrap", R, "pll.c:59-61"),
    ((139, 147), "phase update wrap (fmod within one period, branch-free)", R, "pll.c:113 (libm fmod)"),
    ((148, 164), "lock metric (f64), sweep, frequency integrator", R, "pll.c:114-116, 125-127"),
    ((165, 176), "lock flags as bit arithmetic (locked / locked_once / updown packed in the state word)", R, "pll.c:117-124, 126-127 (nine compare + select pairs as ints)"),
    ((177, 181), (203, 203), "carrier word: clamp, sweep direction", R, "pll.c:126-128"),
    ((182, 187), (196, 196), "quantiser, both rails, packed into 16 bits", R, "main.c:305-306"),
    ((188, 195), "output ring address, 32-symbol flush test", B, "main.c:307-309 `ring[i++]`"),
    ((197, 202), (204, 204), "AGC gain update", R, "agc.c:21-24"),
]
FIRST_OPS = {0: "v_cmp_gt_i32", 10: "v_mul_f32", 20: "v_cvt_f64_f32", 35: "v_cvt_f64_f32", 78: "v_add_f32", 123: "v_med3_f32", 165: "v_bitop3_b32", 182: "v_mul_f32", 197: "v_cvt_f32_f64"}


def firing_path():
    s = (ROOT / "meteor_demod_amd" / "lib" / "demod_kernel_rot.gfx950.s").read_text().split("\n")
    st = next(i for i, l in enumerate(s) if re.match(KERNEL, l))
    en = next(i for i in range(st, len(s)) if s[i].startswith(".Lfunc_end"))
    body = s[st:en]
    # the main loop = the depth-1 loop with the most instructions; the firing path = from the block that holds the first FIR vote
    # (v_cmp_gt_i32 vcc, 8, ...) behind the slide to the s_or_b64 exec that closes the firing
    heads = [i for i, l in enumerate(body) if "Loop Header: Depth=1" in l]
    best = None
    for h in heads:
        k = h
        while k > 0 and not re.match(r"^\.LBB\d+_\d+:", body[k]):
            k -= 1
        lab = body[k].split(":")[0]
        last = max((q for q, t in enumerate(body) if re.search(r"s_c?branch\S*\s+" + re.escape(lab) + r"\b", t)), default=k)
        if best is None or last - k > best[1] - best[0]:
            best = (k, last)
    k, last = best
    out, inasm, asm_blocks = [], False, 0
    for l in body[k:last + 1]:
        if "#ASMSTART" in l:
            inasm = True; asm_blocks += 1; out.append("<<ASM>>"); continue
        if "#ASMEND" in l:
            inasm = False; continue
        if not inasm:
            t = l.split(";")[0].strip()
            if t:
                out.append(t)
    # the firing starts at the label in front of the FIR asm block (2nd asm block of the loop) and ends where the timing-critical section is left (s_setprio 0)
    asm_idx = [i for i, t in enumerate(out) if t == "<<ASM>>"]
    fir = asm_idx[1]
    a = max(i for i in range(fir) if out[i].endswith(":"))
    b = next(i for i in range(fir, len(out)) if out[i].startswith("s_setprio 0"))
    return [t for t in out[a:b] if t.startswith("v_")]


def main():
    table = V.load_costs()
    path = firing_path()
    assert len(path) == 205, f"{len(path)} VALU instructions in the firing path: not the build this stage map was written against"
    for i, op in FIRST_OPS.items():
        assert path[i].split()[0].startswith(op), (i, path[i])
    costs = [V.cost_of(t.split()[0], t, table)[0] for t in path]
    used = set()
    rows, tot = [], collections.Counter()
    n_by = collections.Counter()
    for st in STAGES:
        ranges, (what, cls, ref) = [x for x in st if isinstance(x, tuple)], st[-3:]
        idx = [i for lo, hi in ranges for i in range(lo, hi + 1)]
        assert not (set(idx) & used), what
        used |= set(idx)
        c = sum(costs[i] for i in idx)
        rows.append((what, ref, cls, len(idx), c))
        tot[cls] += c; n_by[cls] += len(idx)
    assert used == set(range(205)), sorted(set(range(205)) - used)
    print("| stage of a firing (compiler-generated VALU instructions, configs[1] instance) | reference | class | instructions | SIMD cycles | % of the VALU pipe time of a wave-firing |")
    print("|---|---|---|---|---|---|")
    for what, ref, cls, n, c in rows:
        print(f"| {what} | `{ref}` | {cls} | {n} | {c:.0f} | {100 * c / PIPE:.1f} |")
    print(f"| **all of the firing path** | | | **{len(path)}** | **{sum(costs):.0f}** | {100 * sum(costs) / PIPE:.1f} |")
    for cls in (R, B):
        print(f"| of which {cls} | | | {n_by[cls]} | {tot[cls]:.0f} | {100 * tot[cls] / PIPE:.1f} |")
    big = max((r for r in rows if r[2] == B), key=lambda r: r[4])
    print(f"\nLargest single bookkeeping item: \"{big[0]}\", {big[4]:.0f} cycles = {100 * big[4] / PIPE:.1f} % of the pipe time.")


main()
