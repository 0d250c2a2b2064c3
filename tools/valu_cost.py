#!/usr/bin/env python3
"""SIMD pipe time of a kernel's VALU instruction mix, from MEASURED per-instruction costs.

    tools/valu_cost.py <kernel .s from hipcc -save-temps> <kernel-name substring> [ubench.jsonl]

`tools/ubench/valu_mix.hip` measures, on an MI355X at the occupancy the demodulator kernels run at (two waves per SIMD), how many
cycles of a SIMD one wave-instruction of each kind takes (profiles/r04_valu_mix_ubench.jsonl).  gfx950 has two classes: plain VOP2
float / integer arithmetic on VGPR (or literal) operands - v_add/sub/mul/fma_f32, v_and/or/xor, v_add/sub_u32, v_mov - issues every
~2.6 cycles; everything else - VOP3 forms, conversions, comparisons, selects, shifts, min/max/med3, every f64 and every packed
(v_pk_*) instruction, and the plain ops as soon as they take an SGPR operand - every 4.2..5.5; v_rcp/rsq_f32 8.4, v_rsq_f64 16.4.
This script classifies the instructions of the kernel's MAIN LOOP (all paths, ONE rotation copy of each assembly block) and prints
the mix-weighted mean cost per VALU instruction; make_profile_md.py multiplies it by the measured SQ_INSTS_VALU per wave-firing to
get the pipe time a wave-firing needs at least, i.e. the `peak_measured` of bench.py's roofline.valu."""
import json, re, sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CHEAP = {"v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_fma_f32", "v_fmac_f32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_add_u32", "v_sub_u32",
         "v_subrev_u32", "v_mov_b32", "v_accvgpr_read_b32", "v_accvgpr_write_b32", "v_not_b32", "v_addc_co_u32", "v_subb_co_u32", "v_subbrev_co_u32",
         "v_add_co_u32", "v_sub_co_u32"}
NAMES = {"v_mul_f32": "v_mul_f32", "v_pk_mul_f32": "v_pk_mul_f32 op_sel bcast", "v_pk_add_f32": "v_pk_add_f32", "v_cvt_f32_i32": "v_cvt_f32_i32",
         "v_mul_f64": "v_mul_f64", "v_fma_f64": "v_fma_f64", "v_fmac_f64": "v_fma_f64", "v_add_f64": "v_add_f64", "v_cvt_f64_f32": "v_cvt_f64_f32",
         "v_cvt_f32_f64": "v_cvt_f32_f64", "v_cvt_i32_f64": "v_cvt_i32_f64", "v_rsq_f64": "v_rsq_f64", "v_cndmask_b32": "v_cndmask_b32_e64 sgpr mask",
         "v_mul_lo_u32": "v_mul_lo_u32", "v_mul_i32_i24": "v_mul_i32_i24", "v_mul_u32_u24": "v_mul_i32_i24", "v_mul_hi_u32": "v_mul_hi_u32",
         "v_med3_f32": "v_med3_f32", "v_cvt_i32_f32": "v_cvt_i32_f32", "v_lshl_add_u32": "v_lshl_add_u32", "v_perm_b32": "v_perm_b32",
         "v_and_b32": "v_and_b32", "v_add_u32": "v_add_u32", "v_sub_u32": "v_sub_u32", "v_lshlrev_b32": "v_lshlrev_b32", "v_lshrrev_b32": "v_lshlrev_b32",
         "v_ashrrev_i32": "v_lshlrev_b32", "v_mov_b32": "v_mov_b32", "v_xor_b32": "v_xor_b32", "v_or_b32": "v_xor_b32", "v_max_f32": "v_max_f32",
         "v_min_f32": "v_max_f32", "v_sub_f32": "v_sub_f32", "v_subrev_f32": "v_sub_f32", "v_add_f32": "v_add_f32 chain", "v_fma_f32": "v_fma_f32",
         "v_bfe_u32": "v_bfe_u32", "v_bfe_i32": "v_bfe_u32", "v_and_or_b32": "v_and_or_b32", "v_sad_u32": "v_sad_u32", "v_add3_u32": "v_add3_u32",
         "v_mad_u32_u24": "v_mad_u32_u24", "v_rcp_f32": "v_rcp_f32", "v_rsq_f32": "v_rcp_f32", "v_floor_f32": "v_floor_f32",
         "v_accvgpr_read_b32": "v_accvgpr_read_b32"}


def load_costs(path=None, waves=2):
    path = Path(path) if path else ROOT / "profiles" / "r04_valu_mix_ubench.jsonl"
    table = {}
    for line in path.read_text().splitlines():
        if line.startswith("{") and '"op"' in line:
            d = json.loads(line)
            if d["waves_per_simd"] == waves:
                table[d["op"]] = d["cycles_at_2p4GHz"]
    return table


def cost_of(op, line, table):
    base = re.sub(r"_e32$|_e64$|_sdwa$|_dpp$", "", op)
    if op.endswith("_sdwa"):
        return table.get("v_cvt_f32_i32_sdwa", 4.3), "conversion (SDWA)"
    sgpr_src = False
    parts = line.split(None, 1)
    if len(parts) > 1:
        srcs = parts[1].split(",")[1:]
        sgpr_src = any(re.match(r"\s*-?\|?(s\d|s\[|vcc|exec)", x) for x in srcs)
    if base in CHEAP:
        if op.endswith("_e64") or sgpr_src:
            return table.get("v_mul_f32 sgpr", 4.25), "plain op with an SGPR operand / VOP3 form"
        return table.get(NAMES.get(base, ""), 2.6), "plain VOP2 f32 / integer (fast path)"
    if base.startswith("v_cmp"):
        return table.get("v_cmp_lt_f32 -> sgpr", 4.5), "comparison"
    if base in ("v_pk_mul_f32", "v_pk_add_f32"):
        return table.get(NAMES[base], 4.8), "packed f32 (FIR taps)"
    if base.endswith("_f64") or "f64" in base:
        return table.get(NAMES.get(base, "v_fma_f64"), 4.4), "f64 / f64 conversion"
    if base == "v_cndmask_b32":
        return table.get("v_cndmask_b32_e64 sgpr mask", 4.35), "select"
    if base.startswith("v_cvt"):
        return table.get(NAMES.get(base, "v_cvt_i32_f32"), 4.25), "conversion"
    if base in NAMES and NAMES[base] in table:
        return table[NAMES[base]], "other VOP3 / shift / min-max / integer multiply"
    return 4.4, "other VOP3 / shift / min-max / integer multiply"


def main_loop(lines, kernel):
    st = next(i for i, l in enumerate(lines) if l.startswith("_Z") and kernel in l and l.rstrip().endswith(":") or (l.startswith("_Z") and kernel in l and "; @" in l))
    en = next(i for i in range(st, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[st:en]
    best = None
    for i, l in enumerate(body):
        if "Loop Header: Depth=1" in l:
            j = i
            while not re.match(r"^\.LBB\d+_\d+:", body[j]):
                j -= 1
            lab = body[j].split(":")[0]
            last = max((k for k, t in enumerate(body) if re.search(r"s_c?branch\S*\s+" + re.escape(lab) + r"\b", t)), default=j)
            if best is None or last - j > best[1] - best[0]:
                best = (j, last)
    return body[best[0]:best[1] + 1]


def mix(asm_path, kernel, table=None, ring_chunks=None):
    """(mean SIMD cycles per VALU instruction, {class: [count, cycles]}) of the kernel's main loop; the copies of an assembly block
    (one per window rotation) count once; the chunks of a FIR RING (packed windows: the code exists once per physical chunk and a
    firing runs through `ring_chunks` of them - (taps + 3) / 16 on average) count `ring_chunks` times."""
    table = table or load_costs()
    loop = main_loop(Path(asm_path).read_text().split("\n"), kernel)
    classes, n_tot, c_tot = {}, 0.0, 0.0
    inasm, block = False, []
    def account(instrs, weight):
        nonlocal n_tot, c_tot
        for t in instrs:
            op = t.split()[0]
            if not op.startswith("v_") or op.startswith("v_readlane") or op.startswith("v_writelane") or op.startswith("v_readfirstlane"):
                continue
            c, cls = cost_of(op, t, table)
            e = classes.setdefault(cls, [0.0, 0.0])
            e[0] += weight; e[1] += weight * c
            n_tot += weight; c_tot += weight * c
    for l in loop:
        t = l.strip()
        if "#ASMSTART" in l:
            inasm, block = True, []
            continue
        if "#ASMEND" in l:
            inasm = False
            # rotation copies: labels .Lxxx_<r>_ mark them; count one copy's worth
            copies = max(1, len({m.group(1) for b in block for m in [re.match(r"^\.L(?:fir|put|fira\d?|mig\d?|puta\d?)_(\d+)_\d+:", b)] if m}))
            ring = ring_chunks if (ring_chunks and any(re.match(r"s_branch\s+\.Lfir_0_", b) for b in block)) else 1.0
            account([b for b in block if b and not b.startswith(".") and not b.startswith(";") and not b.endswith(":")], ring / copies)
            continue
        if inasm:
            block.append(t)
            continue
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        account([t.split(";")[0].strip()], 1.0)
    return c_tot / n_tot, {k: [round(v[0], 1), round(v[1], 1)] for k, v in sorted(classes.items(), key=lambda kv: -kv[1][1])}


if __name__ == "__main__":
    rc = next((float(a.split("=")[1]) for a in sys.argv if a.startswith("ring_chunks=")), None)
    mean, classes = mix(sys.argv[1], sys.argv[2], None, rc)
    print(f"mean SIMD cycles per VALU instruction (static main-loop mix, two waves per SIMD): {mean:.3f}")
    for k, (n, c) in classes.items():
        print(f"  {k:60s} {n:7.1f} instructions {c:8.1f} cycles")
