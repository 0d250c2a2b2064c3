"""PCIe-inclusive rate of the host-buffer entry (mdemod_process_host): never the bench 'value', reported in DESIGN.md."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from meteor_demod_amd import DemodConfig, Demodulator, synth

cfg = DemodConfig(samplerate=230000)
import os
shapes = [(64, 4 << 20), (2048, 1 << 18), (16384, 1 << 15)] if len(sys.argv) < 2 else [(16384, 1 << 15)]
print("pack threads:", os.environ.get("MDEMOD_PACK_THREADS", "8 (default)"), flush=True)
for ns, n in shapes:
    one = synth.generate_host(synth.make_stream(1, 230000, 72000, f0_hz=300.0), n)
    blocks = [one] * ns                       # same host buffer for every stream: the copies are still made
    with Demodulator(cfg, ns) as d:
        d.process_host(blocks)                # warm-up: allocations
        d.reset()
        t0 = time.time(); out = d.process_host(blocks); dt = time.time() - t0
        print(f"streams={ns} x {n} samples ({ns*n*4/1e9:.2f} GB in, {sum(o.nbytes for o in out)/1e9:.2f} GB out): "
              f"{dt*1e3:.0f} ms -> {ns*n/dt/1e9:.2f} GS/s, {ns*n*4/dt/1e9:.1f} GB/s of input", flush=True)
