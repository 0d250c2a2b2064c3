"""Rate of the float-input geometries (hybrid window for 66..129 taps; mid float pairs), for MDEMOD_KERNEL in sys.argv[1:] ("" = default)."""
import os, subprocess, sys
code = r'''
import sys, torch
sys.path.insert(0, ".")
from meteor_demod_amd import DemodConfig, Demodulator, synth
T, L = 196608, 16448
cfgs = {"f32 1MS/s -f64 -O8": DemodConfig(samplerate=1000000, rrc_order=64, interp_factor=8, bps=32),
        "f32 oqpsk 80k 1MS/s -f64 -O8": DemodConfig(samplerate=1000000, symrate=80000, oqpsk=True, rrc_order=64, interp_factor=8, bps=32),
        "f32 230k -f48": DemodConfig(samplerate=230000, rrc_order=48, bps=32),
        "f32 1.024MS/s default": DemodConfig(samplerate=1024000, bps=32),
        "f32 2.048MS/s default": DemodConfig(samplerate=2048000, bps=32),
        "f32 2.048MS/s -f64 -O4": DemodConfig(samplerate=2048000, rrc_order=64, interp_factor=4, bps=32),
        "f32 3.2MS/s default": DemodConfig(samplerate=3200000, bps=32)}
for name, cfg in cfgs.items():
    rec = synth.make_stream(2000, cfg.samplerate, cfg.symrate, oqpsk=cfg.oqpsk, f0_hz=1200.0, fmt=32, rms=0.3)
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
        print(f"  {name:30s} {T * L / ms / 1e6:8.1f} GS/s  {ms:7.3f} ms  {d.kernel_name}", flush=True)
    del buf, x, soft
    torch.cuda.empty_cache()
'''
for k in (sys.argv[1:] or ["", "v1"]):
    print("MDEMOD_KERNEL=%r" % k, flush=True)
    subprocess.run([sys.executable, "-c", code], env=dict(os.environ, MDEMOD_KERNEL=k))
