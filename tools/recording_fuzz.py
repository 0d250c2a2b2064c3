"""Soak test of the overlapped-tile stitcher (mdemod_demodulate_recording, csrc/recording.hip): random sample rates,
QPSK / OQPSK, input formats, carrier offsets, Doppler ramps, clock errors, tile sizes.  Every recording is demodulated
natively with spectral carrier seeds and compared with the serial oracle.  A case counts only if the serial run's lock
is genuine (its PLL frequency is on the synthetic carrier when the pilot hands over, NOTEBOOK.md 3.1), otherwise there is
no serial stream to compare with.  Usage: recording_fuzz.py [n_cases] [seed] [only_case]
(FUZZ_SYMBOLS=lo,hi / FUZZ_RAMPS=a,b,.. draw the length in symbols / the Doppler ramps from there; FUZZ_TILE / FUZZ_SETTLE / FUZZ_SEEDMODE in the environment override a case's tile size / settling length / carrier_seed when one case is replayed)"""
import os
import dataclasses
import sys
import time

sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import oracle_py as O
from meteor_demod_amd import DemodConfig, synth
from meteor_demod_amd.recording import agreement, demodulate_recording_native

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
only = int(sys.argv[3]) if len(sys.argv) > 3 else None
bad, jumps, skipped, t0 = [], [], 0, time.time()
within = []
for ci in range(n_cases):
    oqpsk = bool(rng.random() < (1.0 if os.environ.get("FUZZ_ONLY_OQPSK") else 0.35))
    symrate = 80000 if oqpsk else 72000
    osf = float(rng.choice([2.5, 2.875, 3.1944, 3.6, 4.0, 6.0, 14.2]))
    samplerate = int(symrate * osf)
    bps = int(rng.choice([8, 16, 16, 32]))
    cfg = DemodConfig(samplerate=samplerate, symrate=symrate, oqpsk=oqpsk, bps=bps)
    ramp = float(rng.choice([0.0, 10.0, -25.0, 40.0, -40.0]))
    if os.environ.get("FUZZ_RAMPS"):                       # long recordings: ramps that keep the carrier inside the loop's range
        ramp = float(rng.choice([float(v) for v in os.environ["FUZZ_RAMPS"].split(",")]))
    # start on the side the sweep meets first (the reference sweeps up from 0): keeps its lock genuine in most cases
    f0 = float(rng.uniform(100.0, 900.0)) if ramp <= 0 else float(rng.uniform(-200.0, 600.0))
    ppm = float(rng.uniform(-30.0, 30.0))
    n = int(rng.integers(3_000_000, 9_000_000))
    if os.environ.get("FUZZ_SYMBOLS"):                     # e.g. 42e6,83e6: the 2 048-wave-tile grid of round 5
        lo_s, hi_s = (float(v) for v in os.environ["FUZZ_SYMBOLS"].split(","))
        n = int(rng.uniform(lo_s, hi_s) * osf)
    # a ramp that walks the carrier out of the reference's own sweep range (+-fmax: 3.4 kHz QPSK 72k, 1.9 kHz OQPSK 80k) over the
    # recording leaves nothing to compare with - the serial run loses lock for good (r06: FUZZ_SYMBOLS without FUZZ_RAMPS gave
    # 40 Hz/s x 900 s and six "failures" that were the oracle's): the drift over the recording is kept within 2.4 kHz
    drift_max = 2400.0                                   # (the short default cases reach 2 kHz and keep their ramps)
    dur = n / samplerate
    if abs(ramp) * dur > drift_max:
        ramp = float(np.sign(ramp) * drift_max / dur)
    # s16 amplitude scaled down with the oversampling: at 1 MS/s an RMS of 6000 LSB drives the reference's AGC into a
    # 0 <-> 0.019 limit cycle (its step is absolute, agc.c:13-25), where no two runs agree on anything
    amp = {8: dict(rms=40.0, dc=(1.5, -1.0)), 16: dict(rms=float(rng.choice([1500.0, 6000.0])) * min(1.0, 3.2 / osf)),
           32: dict(rms=0.25, dc=(0.001, -0.002))}[bps]
    esn0 = float(rng.choice([10.0, 12.0, 15.0]))
    if os.environ.get("FUZZ_ESN0"):
        esn0 = float(os.environ["FUZZ_ESN0"])           # e.g. 7: where the reference's own loops start to struggle
    kw = {}
    if rng.random() < 0.5:
        kw["tile_samples"] = int(rng.choice([32768 + 64, 50000 // 64 * 64, 131072 + 64]) * max(1.0, osf / 3.2)) // 64 * 64
    if only is not None and ci != only:
        continue
    mode = "spectrum"
    if only is not None:
        if os.environ.get("FUZZ_TILE"):
            kw["tile_samples"] = int(os.environ["FUZZ_TILE"])
        mode = os.environ.get("FUZZ_SEEDMODE", mode)
        if os.environ.get("FUZZ_CLOCKSEED"):
            kw["clock_seed"] = os.environ["FUZZ_CLOCKSEED"]
        if os.environ.get("FUZZ_MARGIN"):
            kw["pilot_margin_symbols"] = int(os.environ["FUZZ_MARGIN"])
        if os.environ.get("FUZZ_SETTLE"):
            kw["settle_samples"] = int(os.environ["FUZZ_SETTLE"])
        print("replay: esn0", esn0, "amp", amp)
    st = synth.make_stream(1000 + ci, samplerate, symrate, f0_hz=f0, clock_ppm=ppm, esn0_db=esn0,
                           doppler_hz_per_s=ramp, clock_ppm_per_s=0.0 if os.environ.get("FUZZ_CLOCK_RAMP", "1") == "0" else ramp / 137.1,   # a pass moves the clock with the carrier
                           oqpsk=oqpsk, fmt=bps, **amp)
    iq = synth.generate_device([st], n)[0]
    serial, tr, ev = O.oracle_demod(cfg, iq.cpu().numpy(), True)
    soft, rep = demodulate_recording_native(cfg, iq, carrier_seed=mode, **kw)
    steps = 2 if oqpsk else 1
    k = min(int(rep.pilot_symbols), len(tr) - 1)
    t_k = tr["sample_index"][k] / samplerate
    f_true = 2 * np.pi * (f0 + ramp * t_k) / (symrate * steps)
    genuine = rep.pilot_locked and abs(float(tr["pll_freq"][k]) - f_true) < 2 * np.pi * 60.0 / (symrate * steps) and len(ev) == 1
    tag = f"case {ci}: {'oqpsk' if oqpsk else 'qpsk'} fs={samplerate} bps={bps} f0={f0:.0f} ramp={ramp} ppm={ppm:.1f} n={n} {kw}"
    if not genuine or rep.n_tiles < 3:
        skipped += 1
        print(tag, "-> skipped (serial lock not genuine / too few tiles)", flush=True)
        continue
    a = agreement(soft.cpu().numpy(), serial)
    a.pop("windows", None)
    within.append(a["within_1lsb"])
    # the very last symbol of a recording may fire in one run and not in the other (clock phases differ by a fraction of a sample)
    ok = abs(a["len_stitched"] - a["len_serial"]) <= 1 and a["hard_decisions_equal"] > 0.9995 and rep.weak_seams == 0
    if rep.rotation_jumps:
        jumps.append(tag)          # reported by the stitcher itself (round 1 tolerated these; since the predecessor hand-over none is left)
        ok = False
    print(tag, "->", "ok" if ok else "FAIL", {k_: (round(v, 5) if isinstance(v, float) else v) for k_, v in a.items()},
          "tiles", rep.n_tiles, "weak", rep.weak_seams, "weak_carrier", rep.weak_carrier_tiles, "frame_misses", rep.frame_misses,
          "repaired", rep.repaired_tiles, "rotation_jumps", rep.rotation_jumps, f"dr_rms {rep.frame_residual_rms:.2f}", flush=True)
    if not ok:
        bad.append(tag)
print(f"{n_cases} cases in {time.time() - t0:.0f} s, skipped {skipped}, with a reported rotation jump {len(jumps)}, failures {len(bad)}")
if within:
    print("within +-1 LSB of the serial run: min %.4f median %.4f" % (min(within), float(np.median(within))))
for b in bad:
    print("  ", b)
sys.exit(1 if bad else 0)
