"""Are the tiled mode's bad windows the SIGNAL's or the tiles'?  (round 5: the signal's.)

meteor_demod_amd.recording.tiled_vs_twins on a bench signal: the 4096-symbol windows of a tiled run next to what converged twins of the
serial run do IN THE SAME WINDOWS of the same recording.

    python tools/tail_vs_pairs.py [c1|c3|c4] [log2=27] [copies=47] [seed=1000] [clock_ppm=-3.5] [settle=<symbols>] [out=<jsonl>]
"""
import json, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
from bench import demod_config
from meteor_demod_amd import synth
from meteor_demod_amd.recording import demodulate_recording_native, tiled_vs_twins

args = [a for a in sys.argv[1:] if "=" not in a] or ["c1"]
kw = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
n = 1 << int(kw.get("log2", 27))
for tag in args:
    cfg, name = demod_config(tag)
    st = synth.make_stream(int(kw.get("seed", 1000)), cfg.samplerate, cfg.symrate, oqpsk=cfg.oqpsk, f0_hz=1200.0, clock_ppm=float(kw.get("clock_ppm", -3.5)),
                           rms=float(kw.get("rms", 2000.0 if tag == "c4" else 6000.0)))
    iq = synth.generate_device([st], n)[0].contiguous()
    opts = {"settle_samples": int(float(kw["settle"]) * cfg.samplerate / cfg.symrate)} if "settle" in kw else {}
    demodulate_recording_native(cfg, iq[: 1 << 21])
    soft, rep = demodulate_recording_native(cfg, iq, **opts)
    res = {"config": name.split(":")[0], "samples": n, "opts": opts, "tiles": int(rep.n_tiles),
           **tiled_vs_twins(cfg, iq, soft, int(rep.exact_symbols), copies=int(kw.get("copies", 47)))}
    print(json.dumps(res), flush=True)
    if "out" in kw:
        with open(kw["out"], "a") as f:
            f.write(json.dumps(res) + "\n")
    del iq, soft
    torch.cuda.empty_cache()
