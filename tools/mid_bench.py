"""Throughput of a configuration outside BASELINE.json's list: the default filter (-f 32 -O 5) at a high sample rate,
which the mid geometry of the register-window kernel serves.  Usage: mid_bench.py [samplerate] [bps=16] [tiles=393216]
(MDEMOD_KERNEL=v1 in the environment compares with the ring kernel)"""
import sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np, torch
import oracle_py as O
from meteor_demod_amd import DemodConfig, Demodulator, synth
sr = int(sys.argv[1]) if len(sys.argv) > 1 else 1024000
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 16
cfg = DemodConfig(samplerate=sr, bps=bits)
T, L = (int(sys.argv[3]) if len(sys.argv) > 3 else 393216), 16448
rec = synth.make_stream(7, cfg.samplerate, cfg.symrate, f0_hz=1200.0, fmt=bits, **({16: {}, 8: dict(rms=40.0, dc=(1.5, -1.0)), 32: dict(rms=0.25, dc=(0.001, -0.002))}[bits]))
buf = torch.empty((T * L, 2), dtype={16: torch.int16, 8: torch.uint8, 32: torch.float32}[bits], device="cuda")
synth.generate_device([rec], T * L, out=buf.view(1, T * L, 2))
x = buf.view(T, L, 2)
with Demodulator(cfg, T) as d:
    soft = torch.empty((T, d.max_symbols(L), 2), dtype=torch.int8, device="cuda")
    d.process(x, soft=soft); torch.cuda.synchronize()
    cnt = d.symbol_counts()
    for t in (0, 777, T - 1):
        w = O.oracle_demod(cfg, x[t].cpu().numpy())[0]
        assert int(cnt[t]) == len(w) and np.array_equal(soft[t, :len(w)].cpu().numpy(), w), t
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3):
        d.process(x, soft=soft)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 3
    bps = bits / 4 + 2 * cfg.symrate / cfg.samplerate
    print(f"{d.kernel_name}: {sr} S/s, {T*L/ms/1e6:.1f} GS/s, {ms:.2f} ms, {T*L*bps/ms/1e6/8000*100:.1f} % of HBM peak; 3 tiles byte-identical to oracle")
