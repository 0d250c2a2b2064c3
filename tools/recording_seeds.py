"""The +-1 LSB agreement of the stitcher against the serial oracle over many noise / data seeds of the bench signals
(recording_check.py uses one): recording_seeds.py [c1|c3|c4] [n_seeds=12] [log2=24] [key=value: fmax=Hz esn0=dB rms= bps= fs= doppler=Hz/s tile=samples]"""
import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
import oracle_py as O
from meteor_demod_amd import DemodConfig, synth
from meteor_demod_amd.recording import agreement, demodulate_recording_native
kv = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
sys.argv = [a for a in sys.argv if "=" not in a]
tag = sys.argv[1] if len(sys.argv) > 1 else "c1"
n_seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 12
log2 = int(sys.argv[3]) if len(sys.argv) > 3 else 24
cfg = {"c1": DemodConfig(samplerate=230000), "c3": DemodConfig(samplerate=230000, symrate=80000, oqpsk=True),
       "c4": DemodConfig(samplerate=1000000, rrc_order=64, interp_factor=8)}[tag]
import dataclasses
bps = int(kv.get("bps", 16))
cfg = dataclasses.replace(cfg, bps=bps, **({"samplerate": int(kv["fs"])} if "fs" in kv else {}))
fmax_hz, esn0, dop = float(kv.get("fmax", 1500.0)), float(kv.get("esn0", 12.0)), float(kv.get("doppler", 0.0))
amp = {8: dict(rms=40.0, dc=(1.5, -1.0)), 16: dict(rms=2000.0 if tag == "c4" else 6000.0), 32: dict(rms=0.25, dc=(0.001, -0.002))}[bps]
if "rms" in kv: amp["rms"] = float(kv["rms"])
n = 1 << (log2 + (1 if tag == "c4" else 0))
rows = []
genuine_all = []
for k in range(n_seeds):
    rng = np.random.default_rng(77 + k)
    st = synth.make_stream(3000 + k, cfg.samplerate, cfg.symrate, oqpsk=cfg.oqpsk, f0_hz=float(rng.uniform(-fmax_hz, fmax_hz)), clock_ppm=float(rng.uniform(-30, 30)),
                           esn0_db=esn0, fmt=bps, doppler_hz_per_s=dop, clock_ppm_per_s=dop / 137.1, **amp)
    iq = synth.generate_device([st], n)[0]
    serial, tr, ev = O.oracle_demod(cfg, iq.cpu().numpy(), True)
    soft, rep = demodulate_recording_native(cfg, iq, **({"tile_samples": int(kv["tile"])} if "tile" in kv else {}))
    a = agreement(soft.cpu().numpy(), serial); a.pop("windows")
    steps = 2 if cfg.oqpsk else 1
    kk = min(int(rep.pilot_symbols), len(tr) - 1)
    f_hz = st.car_step / 2**32 * cfg.samplerate if st.car_step < 2**31 else (st.car_step - 2**32) / 2**32 * cfg.samplerate
    err_hz = float(tr["pll_freq"][kk]) * cfg.symrate * steps / (2 * np.pi) - f_hz
    genuine = bool(rep.pilot_locked) and abs(err_hz) < 60.0 and len(ev) == 1
    f0_clk = (genuine, round(err_hz, 1), len(ev), round(st.car_step / 2**32 * cfg.samplerate if st.car_step < 2**31 else (st.car_step - 2**32) / 2**32 * cfg.samplerate), int(rep.pilot_locked), int(rep.pilot_samples))
    rows.append((a["within_1lsb"], a["hard_decisions_equal"], a["len_stitched"] - a["len_serial"], rep.frame_misses, rep.repaired_tiles, rep.rotation_jumps, rep.weak_seams))
    genuine_all.append(genuine)
    print(k, rows[-1], "genuine serial lock, its carrier error Hz at the hand-over, lock events, f0 Hz, pilot locked, pilot samples:", f0_clk, flush=True)
gen = [r for r, g in zip(rows, genuine_all) if g]
print(f"{len(gen)} of {n_seeds} recordings with a genuine serial lock (carrier within 60 Hz at the hand-over, one lock event)")
rows = gen or rows
w = np.array([r[0] for r in rows])
print(f"{tag}: {n_seeds} seeds of 2^{int(np.log2(n))} samples: within +-1 LSB min {w.min():.5f} median {np.median(w):.5f} max {w.max():.5f}; "
      f"hard decisions min {min(r[1] for r in rows):.6f}; length differences {sorted(set(r[2] for r in rows))}; wrong frames {sum(r[3] for r in rows)}, repaired {sum(r[4] for r in rows)}, jumps {sum(r[5] for r in rows)}, weak seams {sum(r[6] for r in rows)}")
