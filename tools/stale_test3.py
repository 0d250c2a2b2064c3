import os, sys, numpy as np, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
from meteor_demod_amd import DemodConfig, Demodulator, synth
cfg = DemodConfig(samplerate=230000)
T, L = 393216, 8192
if len(sys.argv) > 1: os.environ["MDEMOD_KERNEL"] = sys.argv[1]
rec = synth.make_stream(1000, 230000, 72000, f0_hz=1200.0)
buf = torch.empty((T * L, 2), dtype=torch.int16, device="cuda")
synth.generate_device([rec], T * L, out=buf.view(1, T * L, 2))
x = buf.view(T, L, 2)
z = torch.zeros_like(buf).view(T, L, 2)
d = Demodulator(cfg, T)
d.process(x); torch.cuda.synchronize()          # launch 0: real signal -> history = last samples
d.reset(); torch.cuda.synchronize()
soft = d.process(z); torch.cuda.synchronize()   # launch 1: zeros in, zero history -> all-zero symbols expected
cnt = torch.tensor([s_.symbols_this_call for s_ in d.status()], device='cuda')
mask = (torch.arange(soft.shape[1], device='cuda')[None, :] < cnt[:, None])
nz = ((soft != 0).any(dim=2) & mask).any(dim=1)
idx = torch.nonzero(nz).flatten().cpu().numpy()
print(f"kernel={os.environ.get('MDEMOD_KERNEL','v2')}: tiles with non-zero output after reset + zero input: {len(idx)}", idx[:10])
# does get_history (hipMemcpy path) see zeros after reset?
d.reset(); torch.cuda.synchronize()
bad = [int(t) for t in idx[:50] if np.abs(d.get_history(int(t))).max() > 0]
print("streams whose history reads back non-zero through hipMemcpy after reset:", len(bad))
