"""Settle-length sweep of mdemod_demodulate_recording against the serial oracle.
Usage: settle_sweep.py [log2=25] [out.md] : configs[1]/[2]/[3], settle in {4k, 8k, 12k, 16k, 24k} symbols, 2 seeds each;
per run: +-1 LSB agreement overall, over the first 4096 symbols of the tile bodies and over the rest, worst 4096-symbol window,
work per sample, seconds.  The serial oracle runs once per (config, seed)."""
import sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np, torch
import oracle_py as O
from meteor_demod_amd import DemodConfig, synth
from meteor_demod_amd.recording import agreement, demodulate_recording_native

log2 = int(sys.argv[1]) if len(sys.argv) > 1 else 25
out_path = sys.argv[2] if len(sys.argv) > 2 else None
CFG = {"configs[1] QPSK 72k": DemodConfig(samplerate=230000), "configs[2] OQPSK 80k": DemodConfig(samplerate=230000, symrate=80000, oqpsk=True),
       "configs[3] 1 MS/s -f 64 -O 8": DemodConfig(samplerate=1000000, rrc_order=64, interp_factor=8)}
rows = ["| config | seed | settle (symbols) | tiles | work / sample | seconds | within +-1 LSB | first 4096 of a body | rest of the bodies | worst 4096 window | decisions | floor (1-LSB perturbation) |",
        "|---|---|---|---|---|---|---|---|---|---|---|---|"]


def by_position(got, serial, rep, n, body_head=4096):
    m = min(len(got), len(serial))
    ok = (np.abs(got[:m].astype(np.int16) - serial[:m].astype(np.int16)).max(axis=1) <= 1)
    sym_per_sample = m / n
    idx = np.arange(m)
    smp = idx / sym_per_sample                                  # approximate input sample of each symbol
    first_tile = rep.exact_symbols                              # pilot + tile 0 are the serial run's own bytes
    in_tiles = idx >= first_tile
    pos = ((smp - rep.pilot_samples) % rep.tile_samples) * sym_per_sample
    head = in_tiles & (pos < body_head)
    rest = in_tiles & ~head
    return float(ok[head].mean()) if head.any() else 1.0, float(ok[rest].mean()) if rest.any() else 1.0


for name, cfg in CFG.items():
    n = 1 << (log2 + (1 if cfg.samplerate > 500000 else 0))
    osf = cfg.samplerate / cfg.symrate
    for seed in (2000, 3001):
        st = synth.make_stream(seed, cfg.samplerate, cfg.symrate, oqpsk=cfg.oqpsk, f0_hz=1200.0 if seed == 2000 else -800.0,
                               clock_ppm=0.0 if seed == 2000 else 12.0, rms=2000.0 if cfg.samplerate > 500000 else 6000.0)
        iq = synth.generate_device([st], n)[0]
        x = iq.cpu().numpy()
        serial = O.oracle_demod(cfg, x)[0]
        x2 = x.copy(); x2[len(x2) // 8, 0] += 1
        pert = O.oracle_demod(cfg, x2)[0]
        mm = min(len(serial), len(pert))
        dd = np.abs(serial[:mm].astype(np.int16) - pert[:mm].astype(np.int16)).max(axis=1)
        f0 = int(np.argmax(dd > 0))
        floor = float((dd[f0:] <= 1).mean())
        demodulate_recording_native(cfg, iq[: 1 << 21])
        for settle in (4000, 8000, 12000, 16000, 24000):
            torch.cuda.synchronize(); t0 = time.time()
            soft, rep = demodulate_recording_native(cfg, iq, settle_samples=int(settle * osf))
            torch.cuda.synchronize(); dt = time.time() - t0
            got = soft.cpu().numpy()
            a = agreement(got, serial)
            h, r = by_position(got, serial, rep, n)
            rows.append(f"| {name} | {seed} | {settle} | {rep.n_tiles} | {rep.samples_demodulated / n:.2f} | {dt:.3f} | {a['within_1lsb']:.5f} | {h:.5f} | {r:.5f} | "
                        f"{a['worst_window']:.4f} | {a['hard_decisions_equal']:.6f} | {floor:.5f} |")
            print(rows[-1], flush=True)
text = "\n".join(rows) + "\n"
if out_path:
    open(out_path, "w").write(text)
