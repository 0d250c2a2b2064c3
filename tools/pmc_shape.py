"""One shape outside bench.py's three, launched a few times: the program behind a rocprofv3 --pmc pass when the question is what binds
a geometry (python3 tools/pmc_shape.py <samplerate> [bps] [launches] [rrc_order=64 interp_factor=8 oqpsk=1 symrate=80000 ...]).  Same tiling as bench.py's other_configs."""
import sys
sys.path.insert(0, ".")
import torch
from meteor_demod_amd import DemodConfig, Demodulator, synth

fs = int(sys.argv[1]); bps = int(sys.argv[2]) if len(sys.argv) > 2 else 16; reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
kw = {k: (float(v) if '.' in v else int(v)) for k, v in (a.split('=') for a in sys.argv[4:])}
if 'oqpsk' in kw: kw['oqpsk'] = bool(kw['oqpsk'])
cfg = DemodConfig(samplerate=fs, bps=bps, **kw)
T, L = (393216 // 2 if bps == 32 else 393216), 16448
rec = synth.make_stream(2000, cfg.samplerate, cfg.symrate, oqpsk=cfg.oqpsk, f0_hz=1200.0, fmt=bps,
                        **(dict(rms=40.0) if bps == 8 else dict(rms=0.25, dc=(0.001, -0.002)) if bps == 32 else {}))
buf = torch.empty((T * L, 2), dtype={8: torch.uint8, 16: torch.int16, 32: torch.float32}[bps], device="cuda")
synth.generate_device([rec], T * L, out=buf.view(1, T * L, 2))
x = buf.view(T, L, 2)
with Demodulator(cfg, T) as d:
    soft = torch.empty((T, d.max_symbols(L), 2), dtype=torch.int8, device="cuda")
    for _ in range(reps):
        d.process(x, soft=soft)
    torch.cuda.synchronize()
    print(d.kernel_name)
