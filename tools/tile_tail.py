"""What the bad windows of the tiled mode ARE.

Per tile of ONE recording: what it was given (clock seed, carrier seed, lead lengths: the stitcher's `[tile]` lines at debug 3), where
it ended up (clock word, carrier word at the end of its body), the serial run's own words at the same symbols (the oracle's trace),
and how the tile's body agrees with the serial run (+-1 LSB overall and in its worst 4096-symbol window).  Then: which of those
predicts the windows below 0.99 - rank correlations, and the worst tiles next to the medians.

    python tools/tile_tail.py [c1|c3|c4] [log2=26] [seed=2000] [clock_ppm=0] [settle=<symbols>] [tile=<symbols>] [out=<json>]
"""
import json, os, sys, tempfile
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
import oracle_py as O
from bench import demod_config
from meteor_demod_amd import synth
from meteor_demod_amd.recording import demodulate_recording_native

args = [a for a in sys.argv[1:] if "=" not in a] or ["c1"]
kw = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
n = 1 << int(kw.get("log2", 26))
W = 4096


def run_with_tile_lines(cfg, iq, **opts):
    """The library writes its trace to the C stderr: fd 2 goes to a file for the duration of the call."""
    os.environ["MDEMOD_RECORDING_DEBUG"] = "3"
    sys.stderr.flush()
    with tempfile.TemporaryFile(mode="w+b") as tmp:
        saved = os.dup(2)
        os.dup2(tmp.fileno(), 2)
        try:
            soft, rep = demodulate_recording_native(cfg, iq, **opts)
            torch.cuda.synchronize()
        finally:
            os.dup2(saved, 2); os.close(saved)
            os.environ["MDEMOD_RECORDING_DEBUG"] = "0"
        tmp.seek(0)
        text = tmp.read().decode(errors="replace")
    tiles = []
    for line in text.splitlines():
        if not line.startswith("[tile] "):
            continue
        f = line.split()
        d = {"i": int(f[1])}
        for k, v in zip(f[2::2], f[3::2]):
            d[k] = float(v) if ("." in v or "e" in v or "n" in v) else int(v)
        tiles.append(d)
    return soft, rep, tiles


def spearman(a, b):
    ra, rb = np.argsort(np.argsort(a)), np.argsort(np.argsort(b))
    return float(np.corrcoef(ra, rb)[0, 1])


for tag in args:
    cfg, name = demod_config(tag)
    rms = float(kw.get("rms", 2000.0 if tag == "c4" else 6000.0))
    st = synth.make_stream(int(kw.get("seed", 2000)), cfg.samplerate, cfg.symrate, oqpsk=cfg.oqpsk, f0_hz=1200.0, rms=rms, clock_ppm=float(kw.get("clock_ppm", 0.0)))
    iq = synth.generate_device([st], n)[0].contiguous()
    opts = {}
    if "settle" in kw: opts["settle_samples"] = int(float(kw["settle"]) * cfg.samplerate / cfg.symrate)
    if "tile" in kw: opts["tile_samples"] = int(float(kw["tile"]) * cfg.samplerate / cfg.symrate) // 64 * 64
    demodulate_recording_native(cfg, iq[: 1 << 21])
    soft, rep, tiles = run_with_tile_lines(cfg, iq, **opts)
    serial, trace, _ = O.oracle_demod(cfg, iq.cpu().numpy(), want_trace=True)
    got = soft.cpu().numpy()
    m = min(len(got), len(serial))
    ok = np.abs(got[:m].astype(np.int16) - serial[:m].astype(np.int16)).max(axis=1) <= 1
    sidx = trace["sample_index"].astype(np.int64)            # input sample at which symbol j fired in the serial run
    om, pf = trace["omega"].astype(np.float64), trace["pll_freq"].astype(np.float64)
    centre = float(np.float32(2 * np.pi * cfg.symrate / (cfg.samplerate * cfg.interp_factor)))
    rows = []
    for t in tiles:
        if t["i"] == 0 or t["len"] == 0:
            continue
        j0, j1 = int(np.searchsorted(sidx, t["E"])), int(np.searchsorted(sidx, t["E"] + t["len"]))
        js = int(np.searchsorted(sidx, t["s0"]))
        j1 = min(j1, m)
        if j1 - j0 < W:
            continue
        body = ok[j0:j1]
        wins = [float(body[a:a + W].mean()) for a in range(0, len(body) - W + 1, W // 2)]
        rows.append({
            "tile": t["i"], "symbols": j1 - j0, "within_1lsb": float(body.mean()), "worst_window": min(wins), "windows_below_0.99": int(sum(w < 0.99 for w in wins)),
            # the clock: seed against the serial run's word where the tile's stream starts; the tile's word at the end of its body against the serial run's there (ppm of the rate)
            "clock_seed_minus_serial_ppm": (t["seed_tf"] - om[min(js, m - 1)]) / centre * 1e6,
            "clock_end_minus_serial_ppm": (t["end_tf"] - om[j1 - 1]) / centre * 1e6,
            "clock_start_minus_serial_ppm": (t["start_tf"] - om[max(j0 - 1, 0)]) / centre * 1e6,       # the tile's word when its body starts (after the settle)
            "first_quarter_within_1lsb": float(body[: len(body) // 4].mean()), "last_quarter_within_1lsb": float(body[-(len(body) // 4):].mean()),
            "serial_clock_moved_over_tile_ppm": (om[j1 - 1] - om[min(js, m - 1)]) / centre * 1e6,
            # the carrier, rad per NCO step
            "carrier_seed_minus_serial": t["seed_f0"] - pf[min(js, m - 1)], "carrier_end_minus_serial": t["end_f0"] - pf[j1 - 1],
            "lead_symbols": int((t["acq"] + t["frm"] + t["stl"]) * cfg.symrate / cfg.samplerate), "out_rot": t["out_rot"],
        })
    if not rows:
        print(json.dumps({"config": name.split(":")[0], "error": "no [tile] lines: library without debug 3?"})); continue
    A = {k: np.array([r[k] for r in rows], dtype=np.float64) for k in rows[0]}
    ulp_ppm = float(np.spacing(np.float32(centre))) / centre * 1e6
    bad = 1.0 - A["within_1lsb"]
    preds = {k: spearman(np.abs(A[k]), bad) for k in ("clock_seed_minus_serial_ppm", "clock_start_minus_serial_ppm", "clock_end_minus_serial_ppm", "serial_clock_moved_over_tile_ppm",
                                                       "carrier_seed_minus_serial", "carrier_end_minus_serial")}
    worst = sorted(rows, key=lambda r: r["worst_window"])[:16]
    has_low = A["windows_below_0.99"] > 0
    def med(k, sel): return float(np.median(np.abs(A[k][sel]))) if sel.any() else None
    e0 = int(rep.exact_symbols)
    allw = np.array([float(ok[a:a + W].mean()) for a in range(e0, m - W + 1, W)])
    out = {"config": name.split(":")[0], "samples": n, "opts": opts, "tiles": len(rows), "within_1lsb": round(float(ok.mean()), 5),
           "work_per_sample": round(rep.samples_demodulated / n, 2), "seconds": [round(rep.pilot_seconds, 4), round(rep.tiles_seconds, 4)],
           "windows_after_the_exact_prefix": {"n": int(len(allw)), "below_0.99": int((allw < 0.99).sum()), "share_below_0.99": round(float((allw < 0.99).mean()), 5),
                                              "worst": round(float(allw.min()), 4), "p01": round(float(np.quantile(allw, 0.01)), 4)},
           "tiles_by_ulps_of_the_clock_word_from_the_serial_run_at_body_end": {str(u): [int(((np.abs(A["clock_end_minus_serial_ppm"]) / ulp_ppm).round() == u).sum()),
                                                                                      round(float(A["within_1lsb"][(np.abs(A["clock_end_minus_serial_ppm"]) / ulp_ppm).round() == u].mean()), 5) if ((np.abs(A["clock_end_minus_serial_ppm"]) / ulp_ppm).round() == u).any() else None]
                                                                             for u in range(0, 7)},
           "tiles_by_ulps_at_body_START": {str(u): [int(((np.abs(A["clock_start_minus_serial_ppm"]) / ulp_ppm).round() == u).sum()),
                                                   round(float(A["first_quarter_within_1lsb"][(np.abs(A["clock_start_minus_serial_ppm"]) / ulp_ppm).round() == u].mean()), 5) if ((np.abs(A["clock_start_minus_serial_ppm"]) / ulp_ppm).round() == u).any() else None]
                                           for u in range(0, 9)},
           "first_vs_last_quarter_of_the_bodies": [round(float(A["first_quarter_within_1lsb"].mean()), 5), round(float(A["last_quarter_within_1lsb"].mean()), 5)],
           "tiles_with_a_window_below_0.99": int(has_low.sum()),
           "spearman_of_|predictor|_with_the_tile's_share_of_bad_symbols": {k: round(v, 3) for k, v in preds.items()},
           "median_|.|_tiles_with_a_low_window_vs_the_rest": {k: [med(k, has_low), med(k, ~has_low)] for k in preds},
           "quantiles_|clock_end_minus_serial_ppm|": [round(float(np.quantile(np.abs(A["clock_end_minus_serial_ppm"]), q)), 4) for q in (0.5, 0.9, 0.99, 1.0)],
           "quantiles_|clock_seed_minus_serial_ppm|": [round(float(np.quantile(np.abs(A["clock_seed_minus_serial_ppm"]), q)), 4) for q in (0.5, 0.9, 0.99, 1.0)],
           "worst_tiles": [{k: (round(v, 5) if isinstance(v, float) else v) for k, v in r.items()} for r in worst]}
    print(json.dumps(out), flush=True)
    if "out" in kw:
        with open(kw["out"], "a") as f:
            f.write(json.dumps(out) + "\n")
    del iq, soft
    torch.cuda.empty_cache()
