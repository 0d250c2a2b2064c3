import sys; sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo')
import numpy as np, torch, time
import oracle_py as O
from meteor_demod_amd import DemodConfig, Demodulator, synth
cfg = DemodConfig(samplerate=230000)
T, L = 1 << 22, 1040
one = synth.generate_device([synth.make_stream(5, 230000, 72000, f0_hz=100.0)], T * L // 64)[0]   # 68M samples, tiled 64x
x = one.view(T // 64, L, 2).repeat(64, 1, 1)
print(x.shape, x.element_size() * x.numel() / 1e9, "GB")
with Demodulator(cfg, T) as d:
    t0 = time.time(); soft = d.process(x); torch.cuda.synchronize(); print("ms", (time.time() - t0) * 1e3)
    cnt = d.symbol_counts()
    for t in (0, 12345, T // 64 - 1, T - 1):
        want = O.oracle_demod(cfg, x[t].cpu().numpy())[0]
        assert int(cnt[t]) == len(want) and np.array_equal(soft[t, :len(want)].cpu().numpy(), want), t
    assert bool((soft[: T // 64] == soft[T - T // 64:]).all())
print("4M streams ok")
