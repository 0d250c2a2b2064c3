"""End-to-end time of ONE huge recording (configs[1] signal, no oracle check): recording_big.py [log2_samples=32] [key=value: tile= settle= margin= (symbols) clkseed=pilot|spectrum]"""
import sys, time
sys.path.insert(0, '.')
import torch
from meteor_demod_amd import DemodConfig, synth
from meteor_demod_amd.recording import demodulate_recording_native
cfg = DemodConfig(samplerate=230000)
n = int(float(sys.argv[1])) if len(sys.argv) > 1 and float(sys.argv[1]) > 64 else 1 << int(sys.argv[1] if len(sys.argv) > 1 else 32)
kv = dict(a.split("=") for a in sys.argv[2:] if "=" in a)
osf = 230000 / 72000
kw = {}
if "tile" in kv: kw["tile_samples"] = int(float(kv["tile"]) * osf) // 64 * 64
if "clkseed" in kv: kw["clock_seed"] = kv["clkseed"]
if "margin" in kv: kw["pilot_margin_symbols"] = int(kv["margin"])
if "settle" in kv: kw["settle_samples"] = int(float(kv["settle"]) * osf)
st = synth.make_stream(1000, 230000, 72000, f0_hz=1200.0, clock_ppm=-3.5)
buf = torch.empty((n, 2), dtype=torch.int16, device="cuda")
synth.generate_device([st], n, out=buf.view(1, n, 2))
demodulate_recording_native(cfg, buf[: 1 << 22], **kw); torch.cuda.synchronize()
for rep_i in range(2):
    t0 = time.time(); soft, rep = demodulate_recording_native(cfg, buf, **kw); torch.cuda.synchronize(); dt = time.time() - t0
    print(f"{n} samples: {dt*1e3:.0f} ms = {n/dt/1e9:.2f} GS/s (pilot {rep.pilot_seconds*1e3:.0f} ms, tiles {rep.tiles_seconds*1e3:.0f} ms) tiles {rep.n_tiles} x {rep.tile_samples} work {rep.samples_demodulated/n:.2f}x "
          f"symbols {rep.n_symbols} misses {rep.frame_misses} repaired {rep.repaired_tiles} jumps {rep.rotation_jumps} weak {rep.weak_seams} fixes {rep.seam_fixes} dr_rms {rep.frame_residual_rms:.3f}", flush=True)
    del soft
