import os, sys, numpy as np, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
from meteor_demod_amd import DemodConfig, Demodulator, synth
cfg = DemodConfig(samplerate=230000)
T, L = 393216, 4096
rec = synth.make_stream(1000, 230000, 72000, f0_hz=1200.0)
one = synth.generate_device([rec], L)            # [1, L, 2]
x = one.expand(T, L, 2)                          # every tile reads the same 16 KB
d = Demodulator(cfg, T)
cap = d.max_symbols(L)
for it in range(3):
    d.reset()
    soft = torch.zeros((T, cap, 2), dtype=torch.int8, device="cuda")
    d.process(x, soft=soft); torch.cuda.synchronize()
    neq = (soft != soft[:1]).flatten(1).any(dim=1)
    idx = torch.nonzero(neq).flatten().cpu().numpy()
    print(f"launch {it}: {len(idx)} tiles differ from tile 0; first {idx[:8]}")
    if len(idx):
        t = int(idx[0]); a = soft[0].cpu().numpy(); b = soft[t].cpu().numpy()
        k = np.flatnonzero((a != b).any(axis=1)); print("   first differing symbols", k[:5], "values", b[k[:3]].tolist(), "vs", a[k[:3]].tolist())
