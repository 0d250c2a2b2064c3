"""One stream through the latency kernel (2^20 samples, configs[1] / [2] / [3]): meant to run under rocprofv3 --pmc / --kernel-trace.
Usage: lat_profile.py [c1|c3|c4] [streams]"""
import os, sys
sys.path.insert(0, '.')
os.environ["MDEMOD_LAT"] = os.environ.get("MDEMOD_LAT", "1")
import torch
from meteor_demod_amd import DemodConfig, Demodulator, synth
tag = sys.argv[1] if len(sys.argv) > 1 else "c1"
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cfg = {"c1": DemodConfig(samplerate=230000), "c3": DemodConfig(samplerate=230000, symrate=80000, oqpsk=True),
       "c4": DemodConfig(samplerate=1000000, rrc_order=64, interp_factor=8)}[tag]
n = 1 << 20
st = synth.make_stream(1000, cfg.samplerate, cfg.symrate, oqpsk=cfg.oqpsk, f0_hz=1200.0, rms=2000.0 if tag == "c4" else 6000.0)
x = synth.generate_device([st], n)[0].unsqueeze(0).expand(ns, n, 2).contiguous()
with Demodulator(cfg, ns) as d:
    for _ in range(3):
        d.process(x)
    torch.cuda.synchronize()
    print(d.kernel_name, "symbols per stream", int(d.symbol_counts()[0]))
