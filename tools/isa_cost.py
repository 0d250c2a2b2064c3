#!/usr/bin/env python3
"""Cost-weighted view of a kernel's hot path: SIMD cycles per wave-instruction (tools/ubench/valu_mix at 2 waves per SIMD, MI355X)
applied to the instructions between two labels / line numbers of a hipcc -S listing (compiler code) plus N copies of an asm block.
    tools/isa_cost.py loop.s [first_line last_line]"""
import re, sys, collections
CHEAP = {"v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_fma_f32", "v_fmac_f32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32",
         "v_mov_b32", "v_accvgpr_read_b32", "v_accvgpr_write_b32", "v_not_b32", "v_addc_co_u32", "v_subb_co_u32", "v_subbrev_co_u32", "v_add_co_u32", "v_sub_co_u32"}
COST = {"v_rsq_f64": 16.4, "v_rcp_f32": 8.4, "v_rsq_f32": 8.4, "v_sqrt_f32": 8.4, "v_mul_lo_u32": 5.5, "v_mul_hi_u32": 4.9, "v_mul_i32_i24": 5.0, "v_mul_u32_u24": 5.0,
        "v_pk_mul_f32": 4.9, "v_pk_add_f32": 4.9, "v_mov_b64": 4.8}
def cost(op, line):
    base = re.sub(r"_e32$|_e64$|_sdwa$|_dpp$", "", op)
    if base in COST: return COST[base]
    if base in CHEAP:
        # an SGPR source operand takes the op off the fast path (v_mul_f32 with an SGPR: 4.25 against 2.7)
        ops = line.split(None, 1)[1] if len(line.split(None, 1)) > 1 else ""
        srcs = ops.split(",")[1:]
        if op.endswith("_e64") or any(re.match(r"\s*-?\|?s\d|\s*-?s\[", s) for s in srcs): return 4.3
        return 2.6
    if base.startswith("v_"): return 4.4
    return 0.0
def main():
    lines = open(sys.argv[1]).read().split("\n")
    lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1, len(lines))
    tot = collections.Counter(); cnt = collections.Counter()
    for l in lines[lo - 1:hi]:
        t = l.strip()
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":") or t.startswith("<<"): continue
        op = t.split()[0]
        if not op.startswith("v_"): continue
        c = cost(op, t); tot[op] += c; cnt[op] += 1
    s = sum(tot.values()); n = sum(cnt.values())
    print(f"{n} VALU instructions, {s:.0f} SIMD cycles")
    for op, c in tot.most_common(40): print(f"  {op:28s} x{cnt[op]:3d}  {c:7.1f}")
main()
