import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np, torch
import oracle_py as O
from meteor_demod_amd import DemodConfig, Demodulator, synth
cfg = DemodConfig(samplerate=1072367, pll_bw=5.0, symrate=72000, interp_factor=4, rrc_order=33, oqpsk=True, freq_max=-1.0, bps=32)
for blocks in ([1000, 8051], [9051], [3000]):
    for seed in (1, 2, 3):
        total = sum(blocks)
        st = synth.make_stream(seed, cfg.samplerate, cfg.symrate, f0_hz=500.0, clock_ppm=10.0, esn0_db=15.0, rms=0.7, dc=(0.7/150, -0.7/250), oqpsk=True, fmt=32)
        iq = synth.generate_host(st, total)
        with Demodulator(cfg, 1) as d:
            parts = []; pos = 0
            for b in blocks:
                soft = d.process(torch.from_numpy(iq[None, pos:pos+b]).cuda()); torch.cuda.synchronize()
                parts.append(soft[0, :int(d.symbol_counts()[0])].cpu().numpy()); pos += b
            g = np.concatenate(parts)
            s = d.status()[0]
            ost = O.OracleStream(cfg); w, tr, ev = ost.run(iq, True)
            k = min(len(g), len(w)); diff = np.flatnonzero((g[:k] != w[:k]).any(axis=1))
            print(d.kernel_name, blocks, seed, "len", len(g), len(w), "first diff", diff[:5], "n diff", len(diff), "freq", s.pll_freq, ost.state.pll_freq, "gain", s.gain, ost.state.gain)
            if len(diff):
                i = diff[0]; print("   gpu", g[max(0,i-2):i+3].tolist(), "orc", w[max(0,i-2):i+3].tolist(), "trace", tr[i])
