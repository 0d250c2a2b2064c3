#!/usr/bin/env python3
"""Condense one tools/profile_round.sh output directory (gpurun_out/<dir>/cX_{stats,FETCH,WRITE,sq}.md) into a profiles/*.md
summary and refresh profiles/hbm_traffic.json (what bench.py reports as roofline.traffic).
usage: make_profile_md.py gpurun_out/<dir> profiles/<name>.md <round tag> [tiles=393216] [tile_samples=16448]"""
import json, re, sys
from pathlib import Path

src, out, tag = Path(sys.argv[1]), Path(sys.argv[2]), sys.argv[3]
T = int(sys.argv[4]) if len(sys.argv) > 4 else 393216
L = int(sys.argv[5]) if len(sys.argv) > 5 else 16448
CFG = {"c1": ("configs[1] QPSK 72k, 230 kS/s, -f 32 -O 5", 72000, 230000, 1), "c3": ("configs[2] OQPSK 80k, 230 kS/s", 80000, 230000, 2),
       "c4": ("configs[3] QPSK 72k, 1 MS/s, -f 64 -O 8", 72000, 1000000, 1)}
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))
import valu_cost
# the built kernel of each configuration: (hipcc -save-temps assembly kept by build.py, mangled-name piece, chunks of the FIR ring a firing runs through)
KASM = {"c1": ("demod_kernel_rot.gfx950.s", "ILi16ELi0ELi14ELi0ELi1E", None), "c3": ("demod_kernel_rot.gfx950.s", "ILi16ELi1ELi6ELi0ELi1E", None),
        "c4": ("demod_kernel_rotp.gfx950.s", "WIDE_16_0_ks109", (129 + 3) / 16)}
tf = ROOT / "profiles" / "hbm_traffic.json"
traffic = json.loads(tf.read_text()) if tf.exists() else {}
lines = [f"# {tag}: demodulator kernels under rocprofv3 (bench.py --config X --steps 10 --warmup 2 --no-cpu-baseline --no-check, {T} tiles x {L} samples)\n",
         "Separate passes per counter group (`tools/profile_round.sh`): `--kernel-trace --stats`, `--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, `--pmc SQ_*`.",
         "FETCH_SIZE is doubled (gfx950 tallies 128-byte requests at 64 B: MI355X_MICROARCH.md, HBM section); WRITE_SIZE as counted.\n",
         "| config | kernel | avg ms (stats pass) | min ms | GS/s at avg | algorithmic GB | % of 8 TB/s | FETCH x2 GB | WRITE GB | traffic / algorithmic | VALU / wave-firing | SALU | LDS | branch | wave-cycles / firing | WAIT_ANY / WAVE_CYCLES | WAIT_INST_ANY / WAVE_CYCLES | ACTIVE_INST_VALU / WAVE_CYCLES |",
         "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
mixes = []
for c, (name, symrate, fs, fires) in CFG.items():
    def grab(kind):
        p = src / f"{c}_{kind}.md"
        return p.read_text() if p.exists() else ""
    st = grab("stats")
    m = re.search(r"\| `([^`]*demod_kernel[^`]*)` \| (\d+) \| ([\d.]+) \| ([\d.]+) \| ([\d.]+) \| ([\d.]+)", st)
    if not m:
        continue
    kname, avg, mn = m.group(1), float(m.group(4)), float(m.group(5))
    def counter(kind, cname):
        mm = re.search(rf"demod_kernel[^`]*` \| {cname} \| \d+ \| ([\d.e+]+) \|", grab(kind))
        return float(mm.group(1)) if mm else float("nan")
    fetch_kb, write_kb = counter("FETCH", "FETCH_SIZE"), counter("WRITE", "WRITE_SIZE")
    algo = T * L * (4 + 2 * symrate / fs)
    hbm = fetch_kb * 1024 * 2 + write_kb * 1024
    # firings per launch: symbols * fires; one wave covers 64 streams
    sym_per_tile = L * symrate / fs
    wave_fir = T * sym_per_tile * fires / 64
    valu, salu, lds, br = (counter("sq", k) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_BRANCH"))
    wc, wa = counter("sq", "SQ_WAVE_CYCLES"), counter("sq", "SQ_WAIT_ANY")
    wi, av = counter("sq2", "SQ_WAIT_INST_ANY"), counter("sq2", "SQ_ACTIVE_INST_VALU")
    lines.append(f"| {name} | `{kname[:70]}` | {avg:.3f} | {mn:.3f} | {T*L/avg/1e6:.1f} | {algo/1e9:.2f} | {algo/avg/1e6/8000*100:.2f} | {fetch_kb*2048/1e9:.2f} | {write_kb*1024/1e9:.2f} | "
                 f"{hbm/algo:.3f} | {valu/wave_fir:.0f} | {salu/wave_fir:.0f} | {lds/wave_fir:.0f} | {br/wave_fir:.0f} | {wc*4/wave_fir:.0f} | {wa/wc:.2f} | {wi/wc:.2f} | {av/wc:.2f} |")
    traffic[f"{c}:{T}x{L}"] = {"hbm_bytes_per_launch": int(hbm), "fetch_size_kb_raw": fetch_kb, "write_size_kb_raw": write_kb,
                               "correction": "reads x2 (gfx950 FETCH_SIZE tallies 128-B requests at 64 B: MI355X_MICROARCH.md HBM section); writes as counted; L2-miss (fabric) bytes, Infinity-Cache hits included",
                               "algorithmic_bytes_per_launch": int(algo), "kernel": kname[:90], "round": tag, "kernel_ms_under_profiler": avg,
                               "valu_per_wave_firing": round(valu / wave_fir, 1), "salu_per_wave_firing": round(salu / wave_fir, 1),
                               "lds_per_wave_firing": round(lds / wave_fir, 1), "wave_cycles_per_firing": round(wc * 4 / wave_fir, 0),
                               "wait_any_over_wave_cycles": round(wa / wc, 3),
                               # two waves share a SIMD: the share of the SIMD's 4-cycle issue quanta that carry a VALU instruction
                               "simd_valu_busy_frac": round(2 * av / wc, 3)}
    try:
        fn, piece, ring = KASM[c]
        mean, classes = valu_cost.mix(ROOT / "meteor_demod_amd" / "lib" / fn, piece, None, ring)
        traffic[f"{c}:{T}x{L}"].update({
            # tools/valu_cost.py: the kernel's VALU mix priced with the per-instruction SIMD costs measured by tools/ubench/valu_mix.hip
            "valu_mean_simd_cycles_per_instruction": round(mean, 3),
            "valu_pipe_cycles_per_wave_firing": round(mean * valu / wave_fir, 0),
            "valu_mix_static_main_loop": classes,
            "samples_per_wave_firing": round(64 * fs / symrate / fires, 2)})
        mixes.append((name, mean, mean * valu / wave_fir, wc * 4 / wave_fir / 2, classes))
    except Exception as e:                                   # (no built assembly next to the library: the counters alone)
        print("valu_cost:", e)
if mixes:
    lines += ["\n## VALU pipe time of the instruction mix (measured per-instruction costs)\n",
              "`tools/ubench/valu_mix.hip` (profiles/r04_valu_mix_ubench.jsonl): SIMD cycles per wave-instruction at two waves per SIMD - plain VOP2 f32 / integer ops on VGPRs 2.4-2.7,",
              "everything else (VOP3, conversions, comparisons, selects, shifts, f64, `v_pk_*`) 4.2-5.5, `v_rsq_f64` 16.4.  `tools/valu_cost.py` prices the kernel's main loop with them.\n",
              "| config | mean SIMD cycles per VALU instruction | VALU pipe cycles per wave-firing (x measured SQ_INSTS_VALU) | SIMD cycles per wave-firing available (wave-cycles / 2 waves) | pipe busy |",
              "|---|---|---|---|---|"]
    for name, mean, pipe, avail, classes in mixes:
        lines.append(f"| {name} | {mean:.2f} | {pipe:.0f} | {avail:.0f} | {pipe / avail:.2f} |")
    for name, mean, pipe, avail, classes in mixes:
        lines.append(f"\n{name}: static main-loop mix (one window rotation; the packed ring weighted by the chunks a firing runs through)\n")
        lines.append("| class | instructions | SIMD cycles |\n|---|---|---|")
        for k, (n_i, c_i) in classes.items():
            lines.append(f"| {k} | {n_i} | {c_i} |")
lines.append("\n(wave-firing = one firing of the symbol clock for each of the 64 streams of a wave; QPSK: one per symbol, OQPSK: two.  SQ_WAVE_CYCLES counts 4-cycle quanta.)\n")
for c in CFG:
    for kind in ("stats", "FETCH", "WRITE", "sq", "sq2"):
        p = src / f"{c}_{kind}.md"
        if p.exists():
            body = "\n".join(l for l in p.read_text().splitlines() if l.startswith("|") and ("demod_kernel" in l or l.startswith("| kernel") or l.startswith("|---")))
            lines.append(f"\n## {c} {kind}\n\n{body}\n")
out.write_text("\n".join(lines) + "\n")
tf.write_text(json.dumps(traffic, indent=1))
print(out.read_text()[:3000])
