import os, sys, numpy as np, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
from meteor_demod_amd import DemodConfig, Demodulator, synth
cfg = DemodConfig(samplerate=230000)
T, L = int(sys.argv[2]), 4096
os.environ["MDEMOD_KERNEL"] = sys.argv[1]
rec = synth.make_stream(1000, 230000, 72000, f0_hz=1200.0)
buf = torch.empty((T * L, 2), dtype=torch.int16, device="cuda")
synth.generate_device([rec], T * L, out=buf.view(1, T * L, 2))
x = buf.view(T, L, 2)
d = Demodulator(cfg, T)
cap = d.max_symbols(L)
outs = []
for it in range(3):
    d.reset()
    soft = torch.zeros((T, cap, 2), dtype=torch.int8, device="cuda")
    d.process(x, soft=soft); torch.cuda.synchronize()
    outs.append(soft)
for i in (1, 2):
    neq = (outs[0] != outs[i]).flatten(1).any(dim=1)
    idx = torch.nonzero(neq).flatten().cpu().numpy()
    print(f"kernel={sys.argv[1]} T={T}: launch 0 vs {i}: {len(idx)} differing tiles; first {idx[:6]}, blocks {np.unique(idx[:2000] // (192 if sys.argv[1]=='v1' else 256))[:8]}")
