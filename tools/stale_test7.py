import os, sys, numpy as np, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
from meteor_demod_amd import DemodConfig, Demodulator, synth
cfg = DemodConfig(samplerate=230000)
T = 393216
d = Demodulator(cfg, T)
x = torch.zeros((1, 8, 2), dtype=torch.int16, device="cuda").expand(T, 8, 2)
soft = torch.zeros((T, 16, 2), dtype=torch.int8, device="cuda")
d.process(x, n_samples=0, soft=soft); torch.cuda.synchronize()
st = d.status()
bad = [i for i in range(T) if st[i].gain != 1.0 or st[i].pll_freq != 0.0 or st[i].n_samples != 0 or st[i].locked]
print("after an EMPTY block: streams whose state changed:", len(bad), bad[:8])
hb = [i for i in list(range(196608, 196700)) + list(range(0, 50)) if np.abs(d.get_history(i)).max() != 0]
print("streams (sampled) with non-zero history:", len(hb), hb[:8])
for i in bad[:3]:
    s = d.get_state(i); print(i, s.agc_gain, s.agc_bias_re, s.pll_phase, s.pll_freq, s.pll_err, s.t_phase, s.t_freq, s.t_prev, s.pll_locked, s.t_dual_state)
