"""The yardstick for a tiled demodulation, measured with the oracle alone (CPU): how far apart are TWO CONVERGED runs of the
reference on the same samples?  Run A is the serial run.  Run B is run A's own state at sample K with its symbol-clock word
(timing.c's _freq) moved by `dppm` ppm - a perturbation the loop pulls in within a few time constants - and then both run on.
After `skip` symbols (the pull-in) the two are two converged runs of the same loop on the same input; they never meet again
(the loop is chaotic at the ulp level), and the fraction of their symbols within +-1 LSB of each other is what ANY independent
demodulation of the later samples can expect against the serial run.  The 1-LSB input perturbation of bench.py starts the two
runs in IDENTICAL states and measures how they drift apart; this measures where the drift ends.
    python tools/converged_floor.py [c1|c3|c4] [log2=25] [seed=2000] [dppm=1.0]"""
import json, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import oracle_py as O
from bench import demod_config
from meteor_demod_amd import synth

args = [a for a in sys.argv[1:] if "=" not in a] or ["c1"]
kw = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
n = 1 << int(kw.get("log2", 25)); W = 4096
for tag in args:
    cfg, name = demod_config(tag)
    st = synth.make_stream(int(kw.get("seed", 2000)), cfg.samplerate, cfg.symrate, oqpsk=cfg.oqpsk, f0_hz=1200.0, rms=2000.0 if tag == "c4" else 6000.0)
    iq = synth.generate_host(st, n)
    K = n // 4
    a = O.OracleStream(cfg); sa0 = a.run(iq[:K])[0]
    b = O.OracleStream(cfg); b.run(iq[:K])
    b.state.t_freq = np.float32(b.state.t_freq * (1.0 + float(kw.get("dppm", 1.0)) * 1e-6))
    sa = a.run(iq[K:])[0]; sb = b.run(iq[K:])[0]
    m = min(len(sa), len(sb))
    skip = int(kw.get("skip", 60000))
    d = np.abs(sa[skip:m].astype(np.int16) - sb[skip:m].astype(np.int16)).max(axis=1)
    ok = d <= 1
    wins = np.array([ok[i:i + W].mean() for i in range(0, len(ok) - W + 1, W)])
    # the 1-LSB perturbation run for comparison
    x = iq.copy(); x[K, 0] += 1
    c = O.oracle_demod(cfg, x)[0][len(sa0):]
    mm = min(len(sa), len(c)); dd = np.abs(sa[:mm].astype(np.int16) - c[:mm].astype(np.int16)).max(axis=1)
    first = int(np.argmax(dd > 0)); okc = dd[first:] <= 1
    winc = np.array([okc[i:i + W].mean() for i in range(0, len(okc) - W + 1, W)])
    print(json.dumps({"config": name.split(":")[0], "samples": n, "symbols_compared": int(len(ok)), "same_length": bool(len(sa) == len(sb)),
                      "converged_pair": {"within_1lsb": round(float(ok.mean()), 5), "worst_window": round(float(wins.min()), 4), "windows_below_0.99": int((wins < 0.99).sum()), "windows": len(wins),
                                         "by_quarter": [round(float(ok[i * len(ok) // 4:(i + 1) * len(ok) // 4].mean()), 5) for i in range(4)]},
                      "one_lsb_perturbation": {"within_1lsb": round(float(okc.mean()), 5), "worst_window": round(float(winc.min()), 4), "windows_below_0.99": int((winc < 0.99).sum()), "windows": len(winc),
                                               "by_quarter": [round(float(okc[i * len(okc) // 4:(i + 1) * len(okc) // 4].mean()), 5) for i in range(4)]}}), flush=True)
