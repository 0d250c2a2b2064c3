import os, subprocess, sys
code = r'''
import sys, torch
sys.path.insert(0, ".")
from meteor_demod_amd import DemodConfig, Demodulator, synth
T, L = 196608, 16448
cfgs = {"1.024M f32": DemodConfig(samplerate=1024000, bps=32), "1M f32 -f64 -O8": DemodConfig(samplerate=1000000, rrc_order=64, interp_factor=8, bps=32),
        "2.048M f32": DemodConfig(samplerate=2048000, bps=32), "3.2M f32": DemodConfig(samplerate=3200000, bps=32), "oqpsk 2.4M f32": DemodConfig(samplerate=2400000, symrate=80000, oqpsk=True, bps=32)}
for name, cfg in cfgs.items():
    rec = synth.make_stream(2000, cfg.samplerate, cfg.symrate, oqpsk=cfg.oqpsk, f0_hz=1200.0, fmt=32, rms=0.25, dc=(0.001, -0.002))
    buf = torch.empty((T * L, 2), dtype=torch.float32, device="cuda")
    synth.generate_device([rec], T * L, out=buf.view(1, T * L, 2))
    x = buf.view(T, L, 2)
    with Demodulator(cfg, T) as d:
        soft = torch.empty((T, d.max_symbols(L), 2), dtype=torch.int8, device="cuda")
        d.process(x, soft=soft)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3): d.process(x, soft=soft)
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 3
        print(f"  {name:18s} {T * L / ms / 1e6:8.1f} GS/s  {d.kernel_name[:50]}", flush=True)
    del buf, x, soft
    torch.cuda.empty_cache()
'''
for lib in sys.argv[1:]:
    print(lib or "product", flush=True)
    subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **({"MDEMOD_LIB_PATH": lib} if lib else {})))
