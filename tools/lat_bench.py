"""Single-stream and few-stream rates of the latency kernel (one stream per wave) against the lane-per-stream kernels.
Usage: lat_bench.py [c1|c3|c4 ...]"""
import os, sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np, torch
import oracle_py as O
from meteor_demod_amd import DemodConfig, Demodulator, synth
CFG = {"c1": DemodConfig(samplerate=230000), "c3": DemodConfig(samplerate=230000, symrate=80000, oqpsk=True),
       "c4": DemodConfig(samplerate=1000000, rrc_order=64, interp_factor=8),
       # other kernel families on the lane side (where does a context change over to them?)
       "c1f": DemodConfig(samplerate=230000, bps=32), "c4f": DemodConfig(samplerate=1000000, rrc_order=64, interp_factor=8, bps=32),
       "mid": DemodConfig(samplerate=1024000), "midf": DemodConfig(samplerate=1024000, bps=32), "far": DemodConfig(samplerate=2048000),
       "t97": DemodConfig(samplerate=230000, rrc_order=48, interp_factor=5, bps=32), "c1u8": DemodConfig(samplerate=230000, bps=8)}
for tag in [a for a in sys.argv[1:] if a in CFG] or ["c1", "c3", "c4"]:
    cfg = CFG[tag]
    n = 1 << 20
    st = synth.make_stream(1000, cfg.samplerate, cfg.symrate, oqpsk=cfg.oqpsk, f0_hz=1200.0, rms=2000.0 if tag == "c4" else 6000.0)
    for ns in (1, 256, 1024, 2048, 4096, 8192):
        x = synth.generate_device([st], n)[0].unsqueeze(0).expand(ns, n, 2).contiguous() if ns <= 1024 else None
        if x is None:
            x = synth.generate_device([st], n // 8)[0].unsqueeze(0).expand(ns, n // 8, 2).contiguous()
        if cfg.bps == 32: x = x.float()
        elif cfg.bps == 8: x = (torch.clamp(torch.div(x, 64, rounding_mode="floor"), -128, 127) + 128).to(torch.uint8)
        m = x.shape[1]
        res = {}
        for lat in ("0", "1"):
            os.environ["MDEMOD_LAT"] = lat
            with Demodulator(cfg, ns) as d:
                soft = d.process(x); torch.cuda.synchronize()
                d.reset(); torch.cuda.synchronize()
                t0 = time.time(); soft = d.process(x); torch.cuda.synchronize(); dt = time.time() - t0
                cnt = int(d.symbol_counts()[0])
                res[lat] = (dt, soft[0, :cnt].cpu().numpy(), d.kernel_name.split(" ")[0])
        same = np.array_equal(res["0"][1], res["1"][1])
        want = O.oracle_demod(cfg, x[0].cpu().numpy())[0] if ns == 1 else None
        ok = "" if want is None else f" oracle-equal {np.array_equal(res['1'][1], want)}"
        print(f"{tag} streams {ns:5d} x {m}: lane-per-stream {m/res['0'][0]/1e6:7.2f} MS/s/stream ({ns*m/res['0'][0]/1e9:6.2f} GS/s)   "
              f"wave-per-stream {m/res['1'][0]/1e6:7.2f} MS/s/stream ({ns*m/res['1'][0]/1e9:6.2f} GS/s)  same bytes {same}{ok}", flush=True)
