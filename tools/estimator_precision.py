import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np, torch
import oracle_py as O
from meteor_demod_amd import DemodConfig, synth
from meteor_demod_amd.recording import estimate_carrier_native
for tag, dop in (("c1", 0.0), ("c1", 20.0), ("c3", 0.0), ("c4", 0.0)):      # (20 Hz/s over 2^24 samples: +1460 Hz, inside the searched band)
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

# ---- symbol clock (mdemod_estimate_clock) against the generator's exact rate -----------------------------------------------
from meteor_demod_amd.recording import estimate_clock_native
for tag, dop, ppm, bps in (("c1", 0.0, 7.3, 16), ("c1", 40.0, -31.0, 16), ("c1", 0.0, 150.0, 8), ("c3", 0.0, 7.3, 16), ("c3", 40.0, -12.0, 32), ("c4", 0.0, 7.3, 16)):
    cfg = {"c1": DemodConfig(samplerate=230000, bps=bps), "c3": DemodConfig(samplerate=230000, symrate=80000, oqpsk=True, bps=bps),
           "c4": DemodConfig(samplerate=1000000, rrc_order=64, interp_factor=8, bps=bps)}[tag]
    n = 1 << 23
    amp = {8: dict(rms=40.0, dc=(1.5, -1.0)), 16: dict(rms=2000.0 if tag == "c4" else 6000.0), 32: dict(rms=0.25, dc=(0.001, -0.002))}[bps]
    st = synth.make_stream(2000, cfg.samplerate, cfg.symrate, oqpsk=cfg.oqpsk, f0_hz=1200.0, clock_ppm=ppm, doppler_hz_per_s=dop, fmt=bps, **amp)
    iq = synth.generate_device([st], n)[0]
    nco = 2 if cfg.oqpsk else 1
    true = 2 * np.pi * (st.sym_step / 2.0 ** 32) / cfg.interp_factor
    for win in (16384, 65536, 262144):
        starts = np.arange(n // 8, n - win - 1, win // 2)
        slope = 2 * np.pi * dop / (cfg.symrate * nco) / cfg.samplerate
        fc = 2 * np.pi * (1200.0 + dop * (starts + win / 2) / cfg.samplerate) / (cfg.symrate * nco)
        tf, q = estimate_clock_native(cfg, iq, starts, win, carrier=fc.astype(np.float32) if cfg.oqpsk else None,
                                      chirp=np.full(len(starts), slope, dtype=np.float32) if (dop and cfg.oqpsk) else None)
        err = (tf.cpu().numpy().astype(np.float64) - true) / true
        print(tag, f"bps {bps} clock {ppm:+.1f} ppm doppler {dop}", "window", win, "relative err mean %.2e std %.2e max %.2e" % (err.mean(), err.std(), np.abs(err).max()),
              "quality %.1f (min %.1f)" % (float(q.mean()), float(q.min())))
# noise alone: the quality that a line must beat
g = torch.Generator(device="cuda").manual_seed(1)
noise = (torch.randn((1 << 22, 2), device="cuda", generator=g) * 800).to(torch.int16)
for cfg in (DemodConfig(samplerate=230000), DemodConfig(samplerate=230000, symrate=80000, oqpsk=True)):
    tf, q = estimate_clock_native(cfg, noise, np.arange(0, (1 << 22) - 65536, 65536), 65536)
    print("noise only:", "oqpsk" if cfg.oqpsk else "qpsk", "quality mean %.1f max %.1f" % (float(q.mean()), float(q.max())))
