import os, sys, numpy as np, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
from meteor_demod_amd import DemodConfig, Demodulator, synth
cfg = DemodConfig(samplerate=230000)
T, L = 393216, 8192
mode = sys.argv[1]
rec = synth.make_stream(1000, 230000, 72000, f0_hz=1200.0)
buf = torch.empty((T * L, 2), dtype=torch.int16, device="cuda")
synth.generate_device([rec], T * L, out=buf.view(1, T * L, 2))
x = buf.view(T, L, 2)
d = Demodulator(cfg, T)
outs = []
for it in range(3):
    if mode == "fresh":
        d.close(); d = Demodulator(cfg, T)
    else:
        d.reset()
    if mode == "sync": torch.cuda.synchronize()
    soft = d.process(x).clone(); torch.cuda.synchronize()
    cnt = torch.tensor([s.symbols_this_call for s in d.status()], device="cuda")
    outs.append((soft, cnt))
s0, c0 = outs[0]
mask = (torch.arange(s0.shape[1], device="cuda")[None, :] < c0[:, None])
for i in (1, 2):
    s, c = outs[i]
    neq = ((s0 != s).any(dim=2) & mask).any(dim=1) | (c0 != c)
    print(f"mode={mode} launch 0 vs {i}: {int(neq.sum())} differing tiles")
