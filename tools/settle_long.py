"""Settle lengths BEYOND the default (24 000 symbols) against the serial oracle, configs[2] (OQPSK) and configs[1], two seeds: does more
settling close the gap to the perturbation floor?  (It does not: NOTEBOOK.md 3.1.)"""
import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np, torch
import oracle_py as O
from meteor_demod_amd import DemodConfig, synth
from meteor_demod_amd.recording import agreement, demodulate_recording_native
for name, cfg, rms in (("oqpsk", DemodConfig(samplerate=230000, symrate=80000, oqpsk=True), 6000.0), ("qpsk", DemodConfig(samplerate=230000), 6000.0)):
    for seed in (2000, 3001):
        st = synth.make_stream(seed, cfg.samplerate, cfg.symrate, oqpsk=cfg.oqpsk, f0_hz=1200.0, clock_ppm=-3.5, rms=rms)
        iq = synth.generate_device([st], 1 << 25)[0]
        serial = O.oracle_demod(cfg, iq.cpu().numpy())[0]
        for settle in (24000, 36000, 48000, 72000):
            soft, rep = demodulate_recording_native(cfg, iq, settle_samples=int(settle * cfg.samplerate / cfg.symrate))
            a = agreement(soft.cpu().numpy(), serial); a.pop("windows")
            print(name, seed, settle, rep.n_tiles, round(a["within_1lsb"], 5), round(a["worst_window"], 4), round(rep.tiles_seconds * 1e3, 1), "ms", flush=True)
