"""Which tiles of a stitched recording start badly?  Per tile: +-1 LSB agreement with the serial oracle over the first 4096
symbols of its body and over the rest.  Usage: tile_start_quality.py c1|c3|c4 [log2=25] [seed=3001] [f0=-800] [ppm=12] [settle=24000]"""
import sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np, torch
import oracle_py as O
from meteor_demod_amd import DemodConfig, synth
from meteor_demod_amd.recording import agreement, demodulate_recording_native

tag = sys.argv[1] if len(sys.argv) > 1 else "c3"
kv = dict(a.split("=") for a in sys.argv[2:] if "=" in a)
CFG = {"c1": DemodConfig(samplerate=230000), "c3": DemodConfig(samplerate=230000, symrate=80000, oqpsk=True),
       "c4": DemodConfig(samplerate=1000000, rrc_order=64, interp_factor=8)}
cfg = CFG[tag]
n = 1 << int(kv.get("log2", 25))
osf = cfg.samplerate / cfg.symrate
st = synth.make_stream(int(kv.get("seed", 3001)), cfg.samplerate, cfg.symrate, oqpsk=cfg.oqpsk, f0_hz=float(kv.get("f0", -800.0)),
                       clock_ppm=float(kv.get("ppm", 12.0)), rms=2000.0 if tag == "c4" else 6000.0)
iq = synth.generate_device([st], n)[0]
serial, trace, ev = O.oracle_demod(cfg, iq.cpu().numpy(), True)
print("serial lock events", ev[:6], "final freq", float(trace[-1]["pll_freq"]))
soft, rep = demodulate_recording_native(cfg, iq, settle_samples=int(float(kv.get("settle", 24000)) * osf))
got = soft.cpu().numpy()
a = agreement(got, serial); a.pop("windows")
print(a, "pilot_locked", rep.pilot_locked, "pilot_samples", rep.pilot_samples, "tiles", rep.n_tiles, "tile_samples", rep.tile_samples,
      "repaired", rep.repaired_tiles, "jumps", rep.rotation_jumps, "frame_misses", rep.frame_misses, "weak", rep.weak_seams, "odd kept", getattr(rep, "odd_tiles_kept", None))
m = min(len(got), len(serial))
ok = (np.abs(got[:m].astype(np.int16) - serial[:m].astype(np.int16)).max(axis=1) <= 1)
sps = m / n
idx = np.arange(m)
smp = idx / sps
tile = np.floor((smp - rep.pilot_samples) / rep.tile_samples).astype(int)
pos = ((smp - rep.pilot_samples) % rep.tile_samples) * sps
heads, rests = [], []
for t in range(1, rep.n_tiles):
    sel = tile == t
    h = ok[sel & (pos < 4096)]; r = ok[sel & (pos >= 4096)]
    heads.append(h.mean() if len(h) else 1.0); rests.append(r.mean() if len(r) else 1.0)
heads, rests = np.array(heads), np.array(rests)
print("first-4096 agreement per tile: mean %.4f  p1 %.4f  p10 %.4f  median %.4f  min %.4f" % (heads.mean(), np.percentile(heads, 1), np.percentile(heads, 10), np.median(heads), heads.min()))
print("rest-of-body agreement per tile: mean %.4f  p1 %.4f  p10 %.4f  median %.4f  min %.4f" % (rests.mean(), np.percentile(rests, 1), np.percentile(rests, 10), np.median(rests), rests.min()))
worst = np.argsort(heads)[:12]
print("worst tiles (index, head, rest):", [(int(t) + 1, round(float(heads[t]), 3), round(float(rests[t]), 3)) for t in worst])
print("tiles with head < 0.97:", int((heads < 0.97).sum()), "of", len(heads))
