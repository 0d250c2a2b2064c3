#!/usr/bin/env python3
"""the DYNAMIC count of packed FIR instructions per wave-firing of the std-window kernels, from two --pmc passes
(tools/price_fir_padding.sh): the product, and the same source with the wave-agreed padding-skip flags zeroed behind hipcc's back
(-DROT_EXP_NOSKIP: every one of the 20 half-chunks runs = 160 packed instructions; the votes that work the flags out stay).
    SQ_INSTS_VALU(noskip) - SQ_INSTS_VALU(product) = packed instructions the skips save;  dynamic = 160 - that, per wave-firing.
usage: fir_dynamic.py gpurun_out/<dir> [tiles=393216] [tile_samples=16448]  ->  markdown on stdout, profiles/r06_fir_padding.json"""
import json, re, sys
from pathlib import Path
src = Path(sys.argv[1])
T = int(sys.argv[2]) if len(sys.argv) > 2 else 393216
L = int(sys.argv[3]) if len(sys.argv) > 3 else 16448
ROOT = Path(__file__).resolve().parent.parent
CFG = {"c1": ("configs[1] QPSK 72k", 72000, 230000, 1, 65), "c3": ("configs[2] OQPSK 80k", 80000, 230000, 2, 65)}
PK_COST = 4.8          # SIMD cycles per v_pk_mul_f32 / v_pk_add_f32 at two waves per SIMD (profiles/r04_valu_mix_ubench.jsonl)


def counter(text, name):
    m = re.search(rf"demod_kernel[^`]*` \| {name} \| \d+ \| ([\d.e+]+) \|", text)
    return float(m.group(1)) if m else float("nan")


def avg_ms(text):
    m = re.search(r"\| `[^`]*demod_kernel[^`]*` \| \d+ \| [\d.]+ \| ([\d.]+) \|", text)
    return float(m.group(1)) if m else float("nan")


ab = {}
f = src / "ab.jsonl"
if f.exists():
    for line in f.read_text().splitlines():
        d = json.loads(line)
        ab.setdefault((d["config"], "product" if d["lib"] == "product" else "noskip"), []).append(d["kernel_ms"])
out = {}
print("| config | SQ_INSTS_VALU / wave-firing, product | with every half-chunk run | packed FIR instructions per wave-firing: static | **dynamic (measured)** | floor (2 per tap) | slots evaluated | "
      "excess over the floor, SIMD cycles (x 4.8) | kernel ms product / all half-chunks (3 A/B rounds, one box) | what the skips are worth |")
print("|---|---|---|---|---|---|---|---|---|---|")
for c, (name, symrate, fs, fires, taps) in CFG.items():
    ship, nosk = (src / f"{c}_ship.md").read_text(), (src / f"{c}_noskip.md").read_text()
    wave_fir = T * (L * symrate / fs) * fires / 64
    v0, v1 = counter(ship, "SQ_INSTS_VALU") / wave_fir, counter(nosk, "SQ_INSTS_VALU") / wave_fir
    dyn = 160 - (v1 - v0)
    a, b = ab.get((c, "product"), [avg_ms(ship)]), ab.get((c, "noskip"), [avg_ms(nosk)])
    ma, mb = sum(a) / len(a), sum(b) / len(b)
    out[c] = {"config": name, "valu_per_wave_firing_product": round(v0, 1), "valu_per_wave_firing_all_half_chunks": round(v1, 1),
              "fir_packed_static": 160, "fir_packed_dynamic": round(dyn, 1), "fir_packed_floor": 2 * taps, "slots_evaluated": round(dyn / 2, 1),
              "kernel_ms_product": round(ma, 3), "kernel_ms_all_half_chunks": round(mb, 3), "skips_worth_pct": round((mb / ma - 1) * 100, 1),
              "source": f"{src}: rocprofv3 --pmc SQ_INSTS_VALU, product vs -DROT_EXP_NOSKIP (tools/price_fir_padding.sh)", "tiles": T, "tile_samples": L}
    print(f"| {name} | {v0:.1f} | {v1:.1f} | 160 | **{dyn:.1f}** | {2 * taps} | {dyn / 2:.1f} of 80 | {(dyn - 2 * taps) * PK_COST:.0f} | {ma:.2f} / {mb:.2f} | {(mb / ma - 1) * 100:.1f} % |")
(ROOT / "profiles" / "r06_fir_padding.json").write_text(json.dumps(out, indent=1) + "\n")
