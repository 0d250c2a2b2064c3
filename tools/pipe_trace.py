"""One mdemod_process_host call on the bench's host-fed shape from rows the caller pinned, three times, wall time each: the program behind
a -DMDEMOD_PIPE_TRACE build (tools/build_exp_pipe.sh; MDEMOD_LIB_PATH) and its MDEMOD_PIPE_SKIP diagnosis knobs.
    MDEMOD_LIB_PATH=gpurun_exp/trace.so [MDEMOD_PIPE_SKIP=1] python tools/pipe_trace.py [staged] [warm] [thp]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from meteor_demod_amd import DemodConfig, Demodulator, synth

ns, n = 16384, 1 << 15
cfg = DemodConfig(samplerate=230000)
one = synth.generate_host(synth.make_stream(1, 230000, 72000, f0_hz=300.0), n)
if "thp" in sys.argv:                         # the caller's buffer on transparent huge pages: fewer translations for the copy engine?
    import mmap
    mm = mmap.mmap(-1, ns * n * 4 + (2 << 20))
    mm.madvise(mmap.MADV_HUGEPAGE)
    raw = np.frombuffer(mm, dtype=np.uint8)
    skip = (-raw.ctypes.data) % (2 << 20)
    buf = raw[skip: skip + ns * n * 4].view(np.int16).reshape(ns, n, 2)
else:
    buf = np.empty((ns, n, 2), np.int16)
buf[:] = one
with Demodulator(cfg, ns) as d:
    if "staged" not in sys.argv: d.pin_host(buf)
    rows = [buf[s] for s in range(ns)]
    a = torch.randn(4096, 4096, device="cuda")
    for rep in range(3):
        d.reset()
        if "warm" in sys.argv:                # the GPU busy right up to the call: are the first copy-ins still slow?
            for _ in range(40): a @ a
            torch.cuda.synchronize()
        t0 = time.time(); out = d.process_host(rows); dt = time.time() - t0
        print(f"call {rep}: {dt*1e3:.2f} ms = {ns*n*4/dt/1e9:.1f} GB/s of input", flush=True)
    if "staged" not in sys.argv: d.unpin_host(buf)
