"""A/B of kernel generations on the geometries outside BASELINE.json's headline: wide (configs[3]), mid, far, for MDEMOD_KERNEL in
sys.argv[1:] ("" = default; NAME=value sets another environment switch instead).  Same tiling as bench.py's other_configs."""
import os, subprocess, sys, json
code = r'''
import sys, json, torch
sys.path.insert(0, ".")
from meteor_demod_amd import DemodConfig, Demodulator, synth
T, L = 393216, 16448
cfgs = {"c4 wide 1MS/s -f64 -O8": DemodConfig(samplerate=1000000, rrc_order=64, interp_factor=8),
        "mid 1.024MS/s": DemodConfig(samplerate=1024000), "far 1.8MS/s": DemodConfig(samplerate=1800000),
        "wide u8": DemodConfig(samplerate=1000000, rrc_order=64, interp_factor=8, bps=8),
        "c4 oqpsk 1MS/s": DemodConfig(samplerate=1000000, symrate=80000, oqpsk=True, rrc_order=64, interp_factor=8),
        "2.048MS/s -f64 -O4": DemodConfig(samplerate=2048000, rrc_order=64, interp_factor=4),
        "3.2MS/s default": DemodConfig(samplerate=3200000), "2.4MS/s u8": DemodConfig(samplerate=2400000, bps=8),
        "6MS/s default": DemodConfig(samplerate=6000000), "10MS/s default": DemodConfig(samplerate=10000000),
        "2.048MS/s f32": DemodConfig(samplerate=2048000, bps=32), "3.2MS/s f32": DemodConfig(samplerate=3200000, bps=32),
        "oqpsk 80k 2.4MS/s": DemodConfig(samplerate=2400000, symrate=80000, oqpsk=True),
        "oqpsk 80k 6MS/s": DemodConfig(samplerate=6000000, symrate=80000, oqpsk=True)}
for name, cfg in cfgs.items():
    T = 393216 // 2 if cfg.bps == 32 else 393216
    rec = synth.make_stream(2000, cfg.samplerate, cfg.symrate, oqpsk=cfg.oqpsk, f0_hz=1200.0, fmt=cfg.bps,
                            **(dict(rms=40.0) if cfg.bps == 8 else dict(rms=0.25, dc=(0.001, -0.002)) if cfg.bps == 32 else {}))
    buf = torch.empty((T * L, 2), dtype={8: torch.uint8, 16: torch.int16, 32: torch.float32}[cfg.bps], device="cuda")
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
        print(f"  {name:28s} {T * L / ms / 1e6:8.1f} GS/s  {ms:7.3f} ms  {d.kernel_name}", flush=True)
    del buf, x, soft
    torch.cuda.empty_cache()
'''
for k in (sys.argv[1:] or ["", "v1"]):
    env = dict([k.split("=", 1)]) if "=" in k else {"MDEMOD_KERNEL": k}          # "NAME=value": any other switch of the library
    print(env, flush=True)
    subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env))
