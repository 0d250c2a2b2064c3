import os, sys, numpy as np, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import oracle_py as O
from meteor_demod_amd import DemodConfig, Demodulator, synth
cfg = DemodConfig(samplerate=230000)
T, L = 393216, 8192
rec = synth.make_stream(1000, 230000, 72000, f0_hz=1200.0)
buf = torch.empty((T * L, 2), dtype=torch.int16, device="cuda")
synth.generate_device([rec], T * L, out=buf.view(1, T * L, 2))
x = buf.view(T, L, 2)
d = Demodulator(cfg, T)
outs = []
for it in range(2):
    d.reset()
    soft = d.process(x).clone(); torch.cuda.synchronize()
    st = d.status()
    cnt = torch.tensor([s.symbols_this_call for s in st], device="cuda")
    outs.append((soft, cnt, st))
(s0, c0, st0), (s1, c1, st1) = outs
mask = (torch.arange(s0.shape[1], device="cuda")[None, :] < c0[:, None])
neq = ((s0 != s1).any(dim=2) & mask).any(dim=1) | (c0 != c1)
idx = torch.nonzero(neq).flatten().cpu().numpy()
print(len(idx), "tiles differ between launch 0 and 1")
for t in idx[:6]:
    want, tr, ev = O.oracle_demod(cfg, x[int(t)].cpu().numpy(), True)
    for li, (s, c, st) in enumerate(outs):
        g = s[int(t), : int(c[int(t)])].cpu().numpy()
        k = min(len(g), len(want)); dd = np.flatnonzero((g[:k] != want[:k]).any(axis=1))
        print(f" tile {t} launch {li}: n={len(g)}/{len(want)} ndiff={len(dd)} first={dd[:3]} gpu={g[dd[0]] if len(dd) else None} want={want[dd[0]] if len(dd) else None} "
              f"status freq={st[int(t)].pll_freq:.6g} gain={st[int(t)].gain:.6g} | oracle freq={tr[-1]['pll_freq']:.6g} gain={tr[-1]['gain']:.6g}")
