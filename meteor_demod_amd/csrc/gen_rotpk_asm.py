#!/usr/bin/env python3
"""Generates rotpk_asm.h: the gfx950 assembly of the rotating PACKED register window (demod_kernel_rotp.hip) - the wide, mid
and far geometries: filters up to 129 taps and / or up to 30 input samples per firing, s16 and u8 input.

Same idea as gen_rotwin_asm.py (the window registers belong to this assembly, a slide renames instead of moving), other
trade-offs, because a 96..160-slot window only fits as RAW samples (one VGPR per s16 sample pair, one per two u8 pairs),
converted at every use:

  tap     v_cvt_f32_i32_sdwa x2 (re, im out of the raw word)  +  v_pk_mul_f32 by the broadcast coefficient  +  v_pk_add_f32
          into the accumulator pair: 4 instructions where the v2 kernel's C++ needed 6 - the reference's two rounded products
          and two rounded sums (filter.c:58-59), oldest tap first.
  ring    one copy of the FIR code per PHYSICAL chunk of 16 slots (not per rotation: ten rotations x 160 taps x 32 bytes would
          be 51 KB of a 64 KB instruction cache): P_0 .. P_{n-1} laid out in a ring, entered at the chunk that holds the first
          tap any lane of the wave needs, left after the chunk that holds the last one (chunks that are padding for every lane
          are never entered).  Two scalar instructions of loop control per 16 taps.
  coefs   four shifted copies of every polyphase bank (the "compact4" table of demod_host.cpp) make every (alignment, 4 taps)
          an aligned ds_read_b128; four buffers of four registers rotate with the four groups of a chunk, so the buffer of a
          group does not depend on where the ring was entered; loads run three groups ahead of the arithmetic.
  put     a slide overwrites the oldest physical chunk: 16 v_mov_b32 (s16) or 8 v_xor_b32 with 0x80808080 (u8: the byte with
          its top bit flipped, read as signed, is wavfile.c:61's `byte - 128`).

Registers: window v[256 - NWR .. 255], then 4 temporaries and 16 coefficient registers below it; the compiler stays below those
(`amdgpu_num_vgpr`; build.py: check_rot_partition scans the emitted assembly of every kernel of this file against its ROTPK_<GEO>_<FMT>_LIMIT, inside build() and in the CPU tests).
"""
import os
import sys

# the tap's arithmetic after the two converts: 1 = v_pk_mul_f32 + v_pk_add_f32 (4 instructions per tap), 0 = v_mul_f32 x2 +
# v_add_f32 x2 (6 per tap, less VALU pipe time: packed f32 ops take 6.3 cycles of a SIMD's pipe, plain ones 2.3)
FORM = int(os.environ.get("ROTPK_FORM", "1"))
# coefficient prefetch: 0 = one load and one s_waitcnt per group of 4 taps, three groups ahead; 1 = two loads and one s_waitcnt
# per PAIR of groups, two groups ahead at the wait (half the s_waitcnt instructions)
PAIRWAIT = int(os.environ.get("ROTPK_PAIRWAIT", "1"))
# slots between two points at which the FIR ring can be left: 4 (after every group), 8 (every half-chunk) or 16 (every chunk)
EXIT = int(os.environ.get("ROTPK_EXIT", "4"))      # (configs[3], GS/s on one box: 16: 423, 8: 429, 4: 432; whole-chunk entry as well: 421)

GEOS = {            # name: (embedded taps, window slots)
    "WIDE": (129, 160),
    "MID": (65, 96),
    "FAR": (65, 112),
}


def q(lines):
    return "\n".join('\t"%s\\n\\t"' % l for l in lines)


def layout(nw, fmt):
    nwr = nw if fmt == 16 else nw // 2          # window registers
    wb = 256 - nwr
    tb = wb - 4                                 # 2 temporary pairs
    cb = tb - 16                                # 4 coefficient buffers of 4
    return nwr, wb, tb, cb


def jump(tag, idx):
    return [
        "s_getpc_b64 vcc",
        ".L%s_pc_%%=:" % tag,
        "s_mul_i32 %%[tmp], %%[%s], (.L%s_1_%%= - .L%s_0_%%=)" % (idx, tag, tag),
        "s_add_u32 %%[tmp], %%[tmp], (.L%s_0_%%= - .L%s_pc_%%=)" % (tag, tag),
        "s_add_u32 vcc_lo, vcc_lo, %[tmp]",
        "s_addc_u32 vcc_hi, vcc_hi, 0",
        "s_setpc_b64 vcc",
    ]


def fir(nw, fmt):
    """The ring of chunks (16 slots = four groups of 4 taps), ENTERED at group granularity: %[entry] = physical chunk of the first
    group any lane of the wave needs, %[sub] = that group's place in the chunk, %[cnt] = exit points (every EXIT slots) to pass
    before leaving, %[addr] = coefficient of slot 0 of the entry chunk.  (Until round 4 the ring was also entered by whole
    chunks: 7.5 slots of leading padding on average where this has 1.5.)"""
    nwr, wb, tb, cb = layout(nw, fmt)
    nch = nw // 16
    assert PAIRWAIT, "the group-granular ring is built on the pairwise prefetch"

    def load(g):                                # group g of the chunk at %[addr] (g may reach into the next chunk)
        b = cb + 4 * (g % 4)
        return "ds_read_b128 v[%d:%d], %%[addr] offset:%d" % (b, b + 3, 16 * g)

    def group(p, g):
        hb = cb + 4 * g

        def src(j):                         # the raw word of slot 4g + j of physical chunk p, and the SDWA selects of (re, im)
            s = 4 * g + j
            if fmt == 16:
                return wb + 16 * p + s, "WORD_0", "WORD_1"
            return wb + 8 * p + s // 2, "BYTE_%d" % (2 * (s & 1)), "BYTE_%d" % (2 * (s & 1) + 1)

        def cvt(j):
            r, a, b = src(j)
            t = tb + 2 * (j & 1)
            return ["v_cvt_f32_i32_sdwa v%d, sext(v%d) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:%s" % (t, r, a),
                    "v_cvt_f32_i32_sdwa v%d, sext(v%d) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:%s" % (t + 1, r, b)]

        def mul(j):
            t = tb + 2 * (j & 1)
            h = hb + 2 * (j // 2)
            if FORM == 0:
                return ["v_mul_f32 v%d, v%d, v%d" % (t, hb + j, t), "v_mul_f32 v%d, v%d, v%d" % (t + 1, hb + j, t + 1)]
            return ["v_pk_mul_f32 v[%d:%d], v[%d:%d], v[%d:%d] op_sel:[0,%d] op_sel_hi:[1,%d]" % (t, t + 1, t, t + 1, h, h + 1, j & 1, j & 1)]

        def add(j):
            t = tb + 2 * (j & 1)
            if FORM == 0:
                return ["v_add_f32 %%[ar], %%[ar], v%d" % t, "v_add_f32 %%[ai], %%[ai], v%d" % (t + 1)]
            return ["v_pk_add_f32 %%[acc], %%[acc], v[%d:%d]" % (t, t + 1)]

        return cvt(0) + cvt(1) + mul(0) + mul(1) + add(0) + cvt(2) + add(1) + cvt(3) + mul(2) + mul(3) + add(2) + add(3)

    # the coefficients of the first groups, on their way before the jump: one case per group of the chunk the ring is entered at
    # (%[sub] = 0..3).  Entering at a second group of a half-chunk skips that half-chunk's own prefetch of the NEXT pair: issue it here.
    # %[tmp] = offset of the group's code within its chunk's (the four groups differ in size: two carry the prefetch, the last the
    # loop control).
    L = ["s_cmp_lg_u32 %[sub], 0", "s_cbranch_scc1 .Lpro_n0_%=",
         load(0), load(1), "s_mov_b32 %[tmp], 0", "s_branch .Lpro_done_%=",
         ".Lpro_n0_%=:", "s_cmp_lg_u32 %[sub], 1", "s_cbranch_scc1 .Lpro_n1_%=",
         load(1), load(2), load(3), "s_mov_b32 %[tmp], (.Lfir_0g1_%= - .Lfir_0_%=)", "s_waitcnt lgkmcnt(2)", "s_branch .Lpro_done_%=",
         ".Lpro_n1_%=:", "s_cmp_lg_u32 %[sub], 2", "s_cbranch_scc1 .Lpro_n2_%=",
         load(2), load(3), "s_mov_b32 %[tmp], (.Lfir_0g2_%= - .Lfir_0_%=)", "s_branch .Lpro_done_%=",
         ".Lpro_n2_%=:", load(3), load(4), load(5), "s_mov_b32 %[tmp], (.Lfir_0g3_%= - .Lfir_0_%=)", "s_waitcnt lgkmcnt(2)",
         ".Lpro_done_%=:"]
    # computed jump to chunk %[entry] (all of one size), plus the offset of group %[sub] within a chunk
    L += ["s_getpc_b64 vcc",
          ".Lfir_pc_%=:",
          "s_add_u32 %[tmp], %[tmp], (.Lfir_0_%= - .Lfir_pc_%=)",
          "s_add_u32 vcc_lo, vcc_lo, %[tmp]",
          "s_addc_u32 vcc_hi, vcc_hi, 0",
          "s_mul_i32 %[tmp], %[entry], (.Lfir_1_%= - .Lfir_0_%=)",
          "s_add_u32 vcc_lo, vcc_lo, %[tmp]",
          "s_addc_u32 vcc_hi, vcc_hi, 0",
          "s_setpc_b64 vcc"]
    test = ["s_sub_u32 %[cnt], %[cnt], 1", "s_cbranch_scc1 .Lfir_end_%="]
    for p in range(nch):
        L += [".Lfir_%d_%%=:" % p]
        for g in range(4):
            if g:
                L += [".Lfir_%dg%d_%%=:" % (p, g)]
            if g % 2 == 0:
                L += [load(g + 2), load(g + 3), "s_waitcnt lgkmcnt(2)"]
            L += group(p, g)
            # the ring is LEFT at the granularity EXIT (slots): every instruction of loop control costs an issue slot of a wave that
            # has only one other wave to hide behind, so the finest exit is not the fastest (measured on configs[3]: NOTEBOOK.md 5.0)
            if g == 3:
                L += ["v_add_u32 %[addr], 64, %[addr]"] + test        # the chunk's base moves on
            elif EXIT == 4 or (EXIT == 8 and g == 1):
                L += test
    L += ["s_branch .Lfir_0_%=", ".Lfir_end_%=:", "s_waitcnt lgkmcnt(0)"]
    return L


def put(nw, fmt):
    nwr, wb, tb, cb = layout(nw, fmt)
    nch = nw // 16
    L = jump("put", "rot")
    for p in range(nch):
        L += [".Lput_%d_%%=:" % p]
        if fmt == 16:
            L += ["v_mov_b32 v%d, %%[g%d]" % (wb + 16 * p + s, s) for s in range(16)]
        else:
            L += ["v_xor_b32 v%d, 0x80808080, %%[g%d]" % (wb + 8 * p + s, s) for s in range(8)]
        L += ["s_branch .Lput_end_%="]
    L += [".Lput_end_%=:"]
    return L


def main():
    out = ["/* GENERATED by gen_rotpk_asm.py - do not edit.  gfx950 assembly of the rotating packed register window. */",
           "#ifndef MDEMOD_ROTPK_ASM_H", "#define MDEMOD_ROTPK_ASM_H", "#define ROTPK_FORM %d" % FORM]
    for name, (kt, nw) in GEOS.items():
        for fmt in (16, 8):
            nwr, wb, tb, cb = layout(nw, fmt)
            tag = "%s_%d" % (name, fmt)
            out.append("#define ROTPK_EXIT %d" % EXIT) if (name, fmt) == ("WIDE", 16) else None
            out.append("#define ROTPK_%s_LIMIT %d   /* first register the compiler may not use */" % (tag, cb))
            out.append("#define ROTPK_%s_FIR_ASM \\\n" % tag + q(fir(nw, fmt)).replace("\n", " \\\n"))
            out.append("#define ROTPK_%s_PUT_ASM \\\n" % tag + q(put(nw, fmt)).replace("\n", " \\\n"))
            out.append("#define ROTPK_%s_CLOBBERS " % tag + ", ".join('"v%d"' % i for i in range(cb, wb)) + ', "v255"')
    out.append("#endif")
    sys.stdout.write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
