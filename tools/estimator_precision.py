import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np, torch
import oracle_py as O
from meteor_demod_amd import DemodConfig, synth
from meteor_demod_amd.recording import estimate_carrier_native
for tag, dop in (("c1", 0.0), ("c1", 40.0), ("c3", 0.0), ("c4", 0.0)):
    cfg = {"c1": DemodConfig(samplerate=230000), "c3": DemodConfig(samplerate=230000, symrate=80000, oqpsk=True),
           "c4": DemodConfig(samplerate=1000000, rrc_order=64, interp_factor=8)}[tag]
    n = 1 << 24
    st = synth.make_stream(2000, cfg.samplerate, cfg.symrate, oqpsk=cfg.oqpsk, f0_hz=1200.0, rms=2000.0 if tag == "c4" else 6000.0, doppler_hz_per_s=dop)
    iq = synth.generate_device([st], n)[0]
    nco = 2 if cfg.oqpsk else 1
    for win in (65536, 262144):
        starts = np.arange(n // 4, n - win - 1, win // 2)
        slope = 2 * np.pi * dop / (cfg.symrate * nco) / cfg.samplerate
        f, q, used = estimate_carrier_native(cfg, iq, starts, win, chirp=np.full(len(starts), slope, dtype=np.float32) if dop else None)
        true = 2 * np.pi * (1200.0 + dop * (starts + used / 2) / cfg.samplerate) / (cfg.symrate * nco)
        err = f.cpu().numpy() - true
        print(tag, "doppler", dop, "window", used, "err mean %.2e std %.2e max %.2e  (rad/NCO step; x20536 symbols x%d = %.3f rad rms)" % (err.mean(), err.std(), np.abs(err).max(), nco, np.sqrt((err**2).mean()) * 20536 * nco), "quality %.0f" % float(q.mean()))
