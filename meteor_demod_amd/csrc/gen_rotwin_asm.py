#!/usr/bin/env python3
"""Generates rotwin_asm.h: the gfx950 assembly of the ROTATING register window (demod_kernel_rot.hip).

Why assembly, and why generated.  The v2 kernel kept its FIR window (80 samples x (re, im) = 160 VGPRs per lane) in a
C++ array and slid it with register moves: 144 v_mov_b32 per slide plus ~30 PHI copies per iteration that hipcc adds
around them - 13 % of the kernel's VALU instructions, none of them the reference's arithmetic (profiles/r02_kernels.md).
Renaming instead of moving needs one copy of the FIR code per window rotation, each with its own physical registers;
hipcc merges such copies behind PHI moves and spills (tried in r01).  So the window registers are taken away from the
compiler (`amdgpu_num_vgpr` keeps it below v80; v80..v95 are the FIR's coefficient buffers, v96..v255 the window) and
the three pieces of code that touch them are written here, one copy per rotation, entered through a computed jump:

  FIR     filter.c:46-65 - the sequential, oldest-first, unfused sum over the 80 window slots (twenty half-chunks of 4
          taps; coefficients three half-chunks ahead in four rotating buffers; an edge chunk of 8 slots that is all
          padding for the whole wave is skipped).  Per tap: v_pk_mul_f32 + v_pk_add_f32 on the (re, im) register pair - the two
          rounded products and two rounded sums of the reference's `acc += mem * coeff` (no fused multiply-add).
  PUT     converts two granules (8 samples) of raw input into one physical chunk: the slide.  Nothing moves.
  PUTF    the same for 16 ready floats (history at kernel start, float input).

Physical chunk q (8 slots) = v[WB + 16 q .. WB + 16 q + 15], slot s of it = (re, im) at +2s, +2s+1.  Logical chunk c of
rotation r is physical chunk (c + r) mod 10.

Run:  python3 gen_rotwin_asm.py > rotwin_asm.h      (build.py does it when the header is older than this script)
"""
import os
import sys

WB = 96          # first window register
CB = None        # coefficient buffers: NBUF x 4 registers below the window (set below)
NCH = 10         # chunks of 8 slots
# coefficient prefetch of the FIR: 0 = one load and one s_waitcnt per half-chunk, three half-chunks ahead; 1 = two loads and one
# s_waitcnt per chunk, one chunk ahead at the wait (half the s_waitcnt instructions)
PAIRWAIT = int(os.environ.get("ROTWIN_PAIRWAIT", "1"))
DC = int(os.environ.get("ROTWIN_DC", "1"))       # with PAIRWAIT: chunks of coefficients in flight behind the one being used (2 * (DC + 1) buffers)
NBUF = 2 * (DC + 1) if PAIRWAIT else 4
CB = 96 - 4 * NBUF
# the FIR's tap: 0 = v_mul x2 + v_add x2; 1 = v_pk_mul_f32 + v_pk_add_f32 on (re, im) pairs (the default); 2 = v_pk_mul_f32 +
# v_add x2; 3 = v_mul x2 + v_pk_add_f32.  2 and 3 keep their products in v[TB:TB+3] (taken from the compiler like the coefficient
# buffers).  All four are the reference's two rounded products and two rounded sums per tap (filter.c:58-59); measured on
# MI355X (configs[1] / configs[2], GS/s, one box): 0: 240.9 / 109.3, 1: 242.7 / 111.0, 2: 230.3 / 104.8, 3: 232.2 / 103.2 - a wave
# issues one instruction per ~4.7 cycles whatever it is, and the packed form halves the FIR's instruction count while its
# longer pipe time is paid by the other wave of the SIMD.
PK = int(os.environ.get("ROTWIN_PK", "1"))
TB = CB - 4      # product temporaries of the mixed forms
CHB = 32         # bytes of coefficients per chunk


def q(lines):
    """C string literal lines for an asm statement."""
    return "\n".join('\t"%s\\n\\t"' % l for l in lines)


def jump(tag, n):
    """Computed jump to copy %[rot] of `tag` (all copies have the same size)."""
    return [
        "s_getpc_b64 vcc",
        ".L%s_pc_%%=:" % tag,
        "s_mul_i32 %%[tmp], %%[rot], (.L%s_1_%%= - .L%s_0_%%=)" % (tag, tag),
        "s_add_u32 %%[tmp], %%[tmp], (.L%s_0_%%= - .L%s_pc_%%=)" % (tag, tag),
        "s_add_u32 vcc_lo, vcc_lo, %[tmp]",
        "s_addc_u32 vcc_hi, vcc_hi, 0",
        "s_setpc_b64 vcc",
    ]


def fir():
    """Half-chunks of 4 taps: one ds_read_b128 each, four rotating buffers of 4 registers, three half-chunks ahead."""
    NH, D = 2 * NCH, 3
    def load(h):
        b = CB + 4 * (h % NBUF)
        return "ds_read_b128 v[%d:%d], %%[addr] offset:%d" % (b, b + 3, 16 * h)
    L = []
    if PK in (0, 2):
        L += ["v_mov_b32 %[ar], 0", "v_mov_b32 %[ai], 0"]
    L += [load(h) for h in range(2 * DC if PAIRWAIT else D)]  # on their way before the jump (the same for every rotation)
    L += jump("fir", NCH)
    for r in range(NCH):
        L += [".Lfir_%d_%%=:" % r]
        for h in range(NH):
            c = h // 2
            if PAIRWAIT:
                if h % 2 == 0:
                    L += [load(x) for x in (h + 2 * DC, h + 2 * DC + 1) if x < NH]
                    L += ["s_waitcnt lgkmcnt(%d)" % min(2 * DC, max(0, NH - 2 - h))]
            else:
                if h + D < NH:
                    L += [load(h + D)]
                L += ["s_waitcnt lgkmcnt(%d)" % min(D, NH - 1 - h)]
            # padding the whole wave agrees on is not run, by half-chunks of 4 slots: %[flags] bit h (h = 0, 1, 2) - half-chunk h is
            # in front of every lane's first tap (each is jumped over on its own, after its prefetch and its wait: the next one
            # may run); bits 3, 4, 5 - half-chunks 19, 18.., 17.. are behind every lane's last tap (the sum leaves before them)
            if h < 3:
                L += ["s_bitcmp1_b32 %%[flags], %d" % h, "s_cbranch_scc1 .Lfir_%d_s%d_%%=" % (r, h)]
            if h >= NH - 3:
                L += ["s_bitcmp1_b32 %%[flags], %d" % (3 + NH - 1 - h), "s_cbranch_scc1 .Lfir_%d_s9_%%=" % r]
            hb = CB + 4 * (h % NBUF)
            wq = WB + 16 * ((c + r) % NCH) + 8 * (h & 1)
            if PK == 1:
                # (re, im) of a slot are an even-aligned register pair, the coefficient is broadcast with op_sel
                def mul(j):
                    return ["v_pk_mul_f32 %%[p%d], v[%d:%d], v[%d:%d] op_sel:[0,%d] op_sel_hi:[1,%d]"
                            % (j & 1, wq + 2 * j, wq + 2 * j + 1, hb + 2 * (j // 2), hb + 2 * (j // 2) + 1, j & 1, j & 1)]
                def add(j):
                    return ["v_pk_add_f32 %%[acc], %%[acc], %%[p%d]" % (j & 1)]
            elif PK == 2:
                def mul(j):
                    t = TB + 2 * (j & 1)
                    return ["v_pk_mul_f32 v[%d:%d], v[%d:%d], v[%d:%d] op_sel:[0,%d] op_sel_hi:[1,%d]"
                            % (t, t + 1, wq + 2 * j, wq + 2 * j + 1, hb + 2 * (j // 2), hb + 2 * (j // 2) + 1, j & 1, j & 1)]
                def add(j):
                    t = TB + 2 * (j & 1)
                    return ["v_add_f32 %%[ar], %%[ar], v%d" % t, "v_add_f32 %%[ai], %%[ai], v%d" % (t + 1)]
            elif PK == 3:
                def mul(j):
                    t = TB + 2 * (j & 1)
                    return ["v_mul_f32 v%d, v%d, v%d" % (t, hb + j, wq + 2 * j), "v_mul_f32 v%d, v%d, v%d" % (t + 1, hb + j, wq + 2 * j + 1)]
                def add(j):
                    t = TB + 2 * (j & 1)
                    return ["v_pk_add_f32 %%[acc], %%[acc], v[%d:%d]" % (t, t + 1)]
            else:
                # products one tap ahead of the sums, two temporaries per tap
                def mul(j):
                    t = "%%[t%d]" % (2 * (j & 1)), "%%[t%d]" % (2 * (j & 1) + 1)
                    return ["v_mul_f32 %s, v%d, v%d" % (t[0], hb + j, wq + 2 * j),
                            "v_mul_f32 %s, v%d, v%d" % (t[1], hb + j, wq + 2 * j + 1)]
                def add(j):
                    t = "%%[t%d]" % (2 * (j & 1)), "%%[t%d]" % (2 * (j & 1) + 1)
                    return ["v_add_f32 %%[ar], %%[ar], %s" % t[0], "v_add_f32 %%[ai], %%[ai], %s" % t[1]]
            L += mul(0)
            for j in range(4):
                if j + 1 < 4:
                    L += mul(j + 1)
                L += add(j)
            if h < 3:
                L += [".Lfir_%d_s%d_%%=:" % (r, h)]
        L += [".Lfir_%d_s9_%%=:" % r, "s_branch .Lfir_end_%="]
    L += [".Lfir_end_%=:", "s_waitcnt lgkmcnt(0)"]
    return L


def put(kind):
    """16 conversions of 8 raw samples (s16: 8 dwords g0..g7; u8: 4 dwords g0..g3, already xor-ed with 0x80808080) or 16
    moves of ready floats (f0..f15) into physical chunk %[rot]."""
    L = ["s_nop 1"] if kind == "u8" else []          # the xor that feeds the SDWA selects may be the instruction before
    L += jump("put", NCH)
    for r in range(NCH):
        L += [".Lput_%d_%%=:" % r]
        wq = WB + 16 * r
        for s in range(8):
            for comp in range(2):
                dst = wq + 2 * s + comp
                if kind == "s16":
                    L += ["v_cvt_f32_i32_sdwa v%d, sext(%%[g%d]) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_%d" % (dst, s, comp)]
                elif kind == "u8":
                    L += ["v_cvt_f32_i32_sdwa v%d, sext(%%[g%d]) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_%d" % (dst, s // 2, 2 * (s & 1) + comp)]
                else:
                    L += ["v_mov_b32 v%d, %%[f%d]" % (dst, 2 * s + comp)]
        L += ["s_branch .Lput_end_%="]
    L += [".Lput_end_%=:"]
    return L


# ---- the AccVGPR half of the hybrid window (float input, 66..129 taps: demod_kernel_rot.hip, WinA) ------------------------
# A wave that has its SIMD to itself (one wave per SIMD: 512 registers per lane) owns 256 AccVGPRs next to its 256 VGPRs.
# VALU instructions cannot read them, v_accvgpr_read_b32 / v_accvgpr_write_b32 move one register each way.  The NEWER 80 slots of
# the 160-slot window live in a[96:255], ten physical chunks of 8 slots like the VGPR half, rotating with it.
AB = 96          # first AccVGPR of the window: a[96:255] (hipcc hands out AccVGPRs from a0 up when it spills into them; build.py checks
                 # that compiler-generated code stays below this, as it does for the VGPRs)


def fir_acc(NA, tag):
    """filter.c:46-65 continued over the AccVGPR part (NA chunks): logical chunk c of rotation r is physical chunk (c + r) mod NA.  Per tap two
    v_accvgpr_read_b32 into a VGPR pair, then the same v_pk_mul_f32 + v_pk_add_f32 as the VGPR half.  %[addr] points at the
    coefficient of the part's first slot; the sum leaves after logical chunk %[last] (the chunk of the wave's last tap; the first
    `lo` chunks are inside every lane's taps and carry no test)."""
    NH = 2 * NA
    lo = 6 if NA == NCH else 0
    # v80..v91: three coefficient buffers of four (a half-chunk each, loaded two half-chunks ahead); v[92:93], v[94:95]: the two
    # pairs a tap's samples are read into
    def load(h):
        b = CB + 4 * (h % 3)
        return "ds_read_b128 v[%d:%d], %%[addr] offset:%d" % (b, b + 3, 16 * h)
    L = [load(h) for h in range(2)]
    L += jump(tag, NA)
    for r in range(NA):
        L += [".L%s_%d_%%=:" % (tag, r)]
        for h in range(NH):
            c = h // 2
            if h + 2 < NH:
                L += [load(h + 2)]
            L += ["s_waitcnt lgkmcnt(%d)" % min(2, NH - 1 - h)]
            hb = CB + 4 * (h % 3)
            aq = AB + 16 * ((c + r) % NA) + 8 * (h & 1)
            def rd(j):
                t = CB + 12 + 2 * (j & 1)
                return ["v_accvgpr_read_b32 v%d, a%d" % (t, aq + 2 * j), "v_accvgpr_read_b32 v%d, a%d" % (t + 1, aq + 2 * j + 1)]
            def mul(j):
                t = CB + 12 + 2 * (j & 1)
                return ["v_pk_mul_f32 v[%d:%d], v[%d:%d], v[%d:%d] op_sel:[0,%d] op_sel_hi:[1,%d]" % (t, t + 1, t, t + 1, hb + 2 * (j // 2), hb + 2 * (j // 2) + 1, j & 1, j & 1)]
            def add(j):
                t = CB + 12 + 2 * (j & 1)
                return ["v_pk_add_f32 %%[acc], %%[acc], v[%d:%d]" % (t, t + 1)]
            # oldest tap first; a tap's product is not the instruction before its sum, a pair is read while the other one is summed
            L += rd(0) + rd(1) + mul(0) + mul(1) + add(0) + rd(2) + add(1) + rd(3) + mul(2) + mul(3) + add(2) + add(3)
            if h % 2 == 1 and lo <= c < NA - 1:
                L += ["s_cmp_eq_u32 %%[last], %d" % c, "s_cbranch_scc1 .L%s_end_%%=" % tag]
        L += ["s_branch .L%s_end_%%=" % tag]
    L += [".L%s_end_%%=:" % tag, "s_waitcnt lgkmcnt(0)"]
    return L


def migrate(NA, tag):
    """The slide of the hybrid window at rotation %[rot] of the VGPR ring: AccVGPR chunk q mod NA (the oldest of the newer part; NA
    divides the ten VGPR chunks, so the two rings stay in step) moves into VGPR chunk q (whose samples leave the window), the 8 new
    samples (16 floats f0..f15) take its place."""
    assert NCH % NA == 0
    L = jump(tag, NCH)
    for r in range(NCH):
        L += [".L%s_%d_%%=:" % (tag, r)]
        L += ["v_accvgpr_read_b32 v%d, a%d" % (WB + 16 * r + k, AB + 16 * (r % NA) + k) for k in range(16)]
        L += ["v_accvgpr_write_b32 a%d, %%[f%d]" % (AB + 16 * (r % NA) + k, k) for k in range(16)]
        L += ["s_branch .L%s_end_%%=" % tag]
    L += [".L%s_end_%%=:" % tag]
    return L


def put_acc(NA, tag):
    """16 ready floats into AccVGPR chunk %[rot] (history and the first samples at kernel start)."""
    L = jump(tag, NA)
    for r in range(NA):
        L += [".L%s_%d_%%=:" % (tag, r)]
        L += ["v_accvgpr_write_b32 a%d, %%[f%d]" % (AB + 16 * r + k, k) for k in range(16)]
        L += ["s_branch .L%s_end_%%=" % tag]
    L += [".L%s_end_%%=:" % tag]
    return L


def main():
    out = []
    out.append("/* GENERATED by gen_rotwin_asm.py - do not edit.  gfx950 assembly of the rotating register window. */")
    out.append("#ifndef MDEMOD_ROTWIN_ASM_H")
    out.append("#define MDEMOD_ROTWIN_ASM_H")
    out.append("#define ROTWIN_WB %d" % WB)
    out.append("#define ROTWIN_CB %d" % CB)
    out.append("#define ROTWIN_LIMIT %d   /* first register the compiler may not use */" % (TB if PK in (2, 3) else CB))
    out.append("#define ROTWIN_NCH %d" % NCH)
    out.append("#define ROTWIN_AB %d   /* first AccVGPR of the hybrid window's newer half */" % AB)
    out.append("#define ROTWIN_PK %d" % PK)
    out.append("#define ROTWIN_FIR_ASM \\\n" + q(fir()).replace("\n", " \\\n"))
    for kind in ("s16", "u8", "f32"):
        out.append("#define ROTWIN_PUT_%s_ASM \\\n" % kind.upper() + q(put(kind)).replace("\n", " \\\n"))
    # ten AccVGPR chunks: 160-slot window (129 taps); two: 96-slot window (65 taps at up to 30 samples per firing); five: 120 slots
    # (65 taps at up to 54)
    for NA, sfx in ((NCH, ""), (2, "2"), (5, "5")):
        out.append("#define ROTWIN_FIR_ACC%s_ASM \\\n" % sfx + q(fir_acc(NA, "fira" + sfx)).replace("\n", " \\\n"))
        out.append("#define ROTWIN_MIGRATE%s_ASM \\\n" % sfx + q(migrate(NA, "mig" + sfx)).replace("\n", " \\\n"))
        out.append("#define ROTWIN_PUT_ACC%s_ASM \\\n" % sfx + q(put_acc(NA, "puta" + sfx)).replace("\n", " \\\n"))
    out.append("#define ROTWIN_ACC_CLOBBERS " + ", ".join('"a%d"' % i for i in range(AB, AB + 16 * NCH)))
    # clobber lists
    out.append("#define ROTWIN_COEF_CLOBBERS " + ", ".join('"v%d"' % i for i in range(TB if PK in (2, 3) else CB, WB)))
    out.append("#endif")
    sys.stdout.write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
