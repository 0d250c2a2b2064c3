"""The yardstick for the tiled mode with enough windows behind it: meteor_demod_amd.recording.
converged_pair_yardstick on the bench signals - two CONVERGED runs of the reference on the same samples, while they are apart, many
times over, the 4096-symbol windows of all copies pooled.

    python tools/converged_pairs.py [c1 c3 c4] [log2=25] [copies=63] [seed=2000] [clock_ppm=0] [out=<jsonl>]
"""
import json, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
from bench import demod_config
from meteor_demod_amd import synth
from meteor_demod_amd.recording import converged_pair_yardstick

args = [a for a in sys.argv[1:] if "=" not in a] or ["c1"]
kw = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
n = 1 << int(kw.get("log2", 25))
for tag in args:
    cfg, name = demod_config(tag)
    rms = float(kw.get("rms", 2000.0 if tag == "c4" else 6000.0))
    st = synth.make_stream(int(kw.get("seed", 2000)), cfg.samplerate, cfg.symrate, oqpsk=cfg.oqpsk, f0_hz=1200.0, rms=rms, clock_ppm=float(kw.get("clock_ppm", 0.0)))
    iq = synth.generate_device([st], n)[0].contiguous()
    res = {"config": name.split(":")[0], "samples": n, **converged_pair_yardstick(cfg, iq, copies=int(kw.get("copies", 63)), seed=int(kw.get("seed", 2000)))}
    print(json.dumps(res), flush=True)
    if "out" in kw:
        with open(kw["out"], "a") as f:
            f.write(json.dumps(res) + "\n")
    del iq
    torch.cuda.empty_cache()
