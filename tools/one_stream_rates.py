import sys, time
sys.path.insert(0, ".")
import torch
from meteor_demod_amd import DemodConfig, Demodulator, synth
for fs, kw in ((230000, {}), (1024000, {}), (1800000, {}), (3200000, {}), (6000000, {}), (10000000, {}), (6000000, dict(oqpsk=True, symrate=80000))):
    cfg = DemodConfig(samplerate=fs, **kw)
    n = 1 << 22
    rec = synth.make_stream(7, cfg.samplerate, cfg.symrate, oqpsk=cfg.oqpsk, f0_hz=900.0)
    x = synth.generate_device([rec], n)
    with Demodulator(cfg, 1) as d:
        d.process(x); torch.cuda.synchronize()
        t0 = time.time(); d.process(x); torch.cuda.synchronize(); dt = time.time() - t0
        print(f"{fs/1e6:6.3f} MS/s {'oqpsk' if cfg.oqpsk else 'qpsk '}: one stream {n/dt/1e6:7.2f} MS/s ({n/dt/fs:5.1f}x real time)  {d.kernel_name[:60]}", flush=True)
