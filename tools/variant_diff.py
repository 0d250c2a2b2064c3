"""Full-size cross-check: the packed and float kernel variants (and v1) must give identical bytes on
every tile of the bench buffer.  Lists differing tiles and checks them against the oracle."""
import os, sys, numpy as np, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import oracle_py as O
from meteor_demod_amd import DemodConfig, Demodulator, synth
cfg = DemodConfig(samplerate=230000)
T, L = int(sys.argv[1]) if len(sys.argv) > 1 else 393216, 16384
rec = synth.make_stream(1000, 230000, 72000, f0_hz=1200.0, clock_ppm=7.0 * (0 - 1 / 2))
buf = torch.empty((T * L, 2), dtype=torch.int16, device="cuda")
synth.generate_device([rec], T * L, out=buf.view(1, T * L, 2))
x = buf.view(T, L, 2)
res = {}
for name, env in (("packed", {"MDEMOD_RW_PACKED": "1"}), ("float", {"MDEMOD_RW_PACKED": "0"}), ("float2", {"MDEMOD_RW_PACKED": "0"})):
    os.environ.update(env)
    d = Demodulator(cfg, T)
    soft = d.process(x); torch.cuda.synchronize()
    cnt = torch.tensor([s.symbols_this_call for s in d.status()], device="cuda")
    res[name] = (soft, cnt)
    d.close()
sp, cp = res["packed"]
for other in ("float", "float2"):
    so, co = res[other]
    mask = (torch.arange(sp.shape[1], device="cuda")[None, :] < cp[:, None])
    neq = ((sp != so).any(dim=2) & mask).any(dim=1) | (cp != co)
    idx = torch.nonzero(neq).flatten().cpu().numpy()
    print(f"packed vs {other}: {len(idx)} differing tiles", idx[:20])
    for t in idx[:3]:
        want = O.oracle_demod(cfg, x[int(t)].cpu().numpy())[0]
        for nm, (s, c) in (("packed", (sp, cp)), (other, (so, co))):
            g = s[int(t), : int(c[int(t)])].cpu().numpy()
            k = min(len(g), len(want)); dd = np.flatnonzero((g[:k] != want[:k]).any(axis=1))
            print(f"   tile {t} {nm}: count {len(g)} vs oracle {len(want)}; first diff sym {dd[0] if len(dd) else None}; lane {t % 64} wave {t // 64 % 4} block {t // 256}")
