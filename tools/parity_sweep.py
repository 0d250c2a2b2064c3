"""Broad parity sweep: many tiles of the bench recording, GPU vs oracle, for a kernel variant."""
import sys, os, numpy as np, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import oracle_py as O
from meteor_demod_amd import DemodConfig, Demodulator, synth
cfg = DemodConfig(samplerate=230000)
L = 16384
NT = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
first = int(sys.argv[2]) if len(sys.argv) > 2 else 334476 - 100
rec = synth.make_stream(1000, 230000, 72000, f0_hz=1200.0, clock_ppm=7.0 * (0 - 1 / 2))
buf = torch.empty((NT * L, 2), dtype=torch.int16, device="cuda")
synth.generate_device([rec], NT * L, out=buf.view(1, NT * L, 2), n0=first * L)
x = buf.view(NT, L, 2)
with Demodulator(cfg, NT) as d:
    soft = d.process(x); torch.cuda.synchronize()
    st = d.status()
    xs = x.cpu().numpy(); sf = soft.cpu().numpy()
    bad = []
    for t in range(NT):
        want = O.oracle_demod(cfg, xs[t])[0]
        m = st[t].symbols_this_call
        if m != want.shape[0] or not np.array_equal(sf[t, :m], want):
            k = min(m, want.shape[0]); diff = np.flatnonzero((sf[t,:k] != want[:k]).any(axis=1))
            bad.append((t, first + t, m, want.shape[0], int(diff[0]) if len(diff) else -1))
    print(f"kernel={os.environ.get('MDEMOD_KERNEL','v3')}: {NT} tiles, {len(bad)} bad", bad[:8])
