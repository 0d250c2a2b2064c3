"""Where the stitched output of ONE recording differs from the serial run: per configuration
    - truth check of the whole output against the transmitted symbols (synth.truth_check: no serial run needed),
    - agreement with the serial oracle per 4096-symbol window, the worst windows located relative to the tile seams,
    - the reference's own 1-LSB-perturbation floor computed the same way.
    python tools/tiled_evidence.py [c1 c3 c4] [log2=26] [rms=..] [settle=..]"""
import json, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
import oracle_py as O
from bench import demod_config
from meteor_demod_amd import synth
from meteor_demod_amd.recording import demodulate_recording_native

args = [a for a in sys.argv[1:] if "=" not in a] or ["c1", "c3", "c4"]
kw = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
n = 1 << int(kw.get("log2", 26))
W = 4096
for tag in args:
    cfg, name = demod_config(tag)
    rms = float(kw.get("rms", 2000.0 if tag == "c4" else 6000.0))
    st = synth.make_stream(int(kw.get("seed", 2000)), cfg.samplerate, cfg.symrate, oqpsk=cfg.oqpsk, f0_hz=1200.0, rms=rms)
    iq = synth.generate_device([st], n)[0].contiguous()
    opts = {}
    if "settle" in kw: opts["settle_samples"] = int(float(kw["settle"]) * cfg.samplerate / cfg.symrate)
    if "margin" in kw: opts["pilot_margin_symbols"] = int(kw["margin"])
    if "tile" in kw: opts["tile_samples"] = int(float(kw["tile"]) * cfg.samplerate / cfg.symrate) // 64 * 64
    demodulate_recording_native(cfg, iq[: 1 << 21])
    torch.cuda.synchronize(); t0 = time.time()
    soft, rep = demodulate_recording_native(cfg, iq, **opts)
    torch.cuda.synchronize(); dt = time.time() - t0
    truth = synth.truth_check(st, soft.contiguous(), first_symbol=int(max(rep.first_lock_symbol, 0)) + 20000)
    serial = O.oracle_demod(cfg, iq.cpu().numpy())[0]
    got = soft.cpu().numpy()
    m = min(len(got), len(serial))
    ok = np.abs(got[:m].astype(np.int16) - serial[:m].astype(np.int16)).max(axis=1) <= 1
    sps = rep.n_symbols / n
    body0 = rep.pilot_samples * sps                      # symbol at which tile 0's body starts (the exact continuation of the head)
    tile_sym = rep.tile_samples * sps
    wins = np.array([ok[i:i + W].mean() for i in range(0, m - W + 1, W)])
    order = np.argsort(wins)[:12]
    rows = []
    for w in order:
        s0 = w * W
        t = (s0 + W / 2 - body0) / tile_sym
        seam = round(t) * tile_sym + body0                # nearest seam (start of a tile's body)
        rows.append({"window_at_symbol": int(s0), "within_1lsb": round(float(wins[w]), 4), "tile": int(np.floor(t)) if t >= 0 else "head",
                     "position_in_tile_body": round(float(t - np.floor(t)), 3) if t >= 0 else None, "symbols_from_nearest_seam": int(s0 + W / 2 - seam)})
    # by position within a tile's body: is the damage near the seams?
    idx = np.arange(m); tiled = idx >= int(rep.exact_symbols)
    pos = ((idx - body0) % tile_sym) / tile_sym
    bins = [float(ok[tiled & (pos >= a) & (pos < a + 0.1)].mean()) for a in np.arange(0, 1, 0.1)]
    x = iq[: 1 << 25].cpu().numpy(); a = O.oracle_demod(cfg, x)[0]; x[len(x) // 8, 0] += 1; b = O.oracle_demod(cfg, x)[0]
    mm = min(len(a), len(b)); d = np.abs(a[:mm].astype(np.int16) - b[:mm].astype(np.int16)).max(axis=1); first = int(np.argmax(d > 0))
    okf = d[first:] <= 1; fw = [float(okf[i:i + W].mean()) for i in range(0, len(okf) - W + 1, W)]
    print(json.dumps({"config": name.split(":")[0], "opts": opts, "pilot_samples": int(rep.pilot_samples), "pilot_seconds": round(rep.pilot_seconds, 4), "samples": n, "rms": rms, "seconds": round(dt, 3), "symbols": [int(rep.n_symbols), len(serial)],
                      "tiles": int(rep.n_tiles), "tile_symbols": round(tile_sym, 1), "seam_fixes": int(rep.seam_fixes), "weak_seams": int(rep.weak_seams),
                      "repaired": int(rep.repaired_tiles), "rotation_jumps": int(rep.rotation_jumps), "weak_clock_tiles": int(rep.weak_clock_tiles),
                      "within_1lsb": round(float(ok.mean()), 5), "worst_window": round(float(wins.min()), 4), "windows_below_0.99": int((wins < 0.99).sum()), "windows": len(wins),
                      "floor": {"within_1lsb": round(float(okf.mean()), 5), "worst_window": round(min(fw), 4), "windows_below_0.99": int((np.array(fw) < 0.99).sum()), "windows": len(fw)},
                      "within_1lsb_by_tenth_of_tile_body": [round(v, 5) for v in bins], "worst_windows": rows, "truth": truth}), flush=True)
    del iq, soft
    torch.cuda.empty_cache()
