import os, sys, numpy as np, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
from meteor_demod_amd import DemodConfig, Demodulator, synth
cfg = DemodConfig(samplerate=230000)
T = 393216
rec = synth.make_stream(1000, 230000, 72000, f0_hz=1200.0)
for L in (64, 512, 4096):
    one = synth.generate_device([rec], L)
    x = one.expand(T, L, 2)
    d = Demodulator(cfg, T)
    cap = d.max_symbols(L)
    soft = torch.zeros((T, cap, 2), dtype=torch.int8, device="cuda")
    d.process(x, soft=soft); torch.cuda.synchronize()
    neq = (soft != soft[:1]).flatten(1).any(dim=1)
    idx = torch.nonzero(neq).flatten().cpu().numpy()
    st = d.status()
    bad_state = [i for i in idx[:200] if (st[i].gain != st[0].gain or st[i].pll_freq != st[0].pll_freq)]
    print(f"L={L}: {len(idx)} tiles differ from tile 0; first {idx[:6]}; tile0 syms {st[0].symbols_this_call}; of first 200 bad, state differs in {len(bad_state)}")
    if len(idx):
        t = int(idx[0]); print("   tile0:", soft[0, :6].flatten().tolist(), " bad:", soft[t, :6].flatten().tolist(), "gain", st[t].gain, st[0].gain)
    d.close()
