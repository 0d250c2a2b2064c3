"""How memory-bound is the demod kernel?  Same compute, three memory footprints."""
import sys, time, torch
sys.path.insert(0, '.')
from meteor_demod_amd import DemodConfig, Demodulator, synth
cfg = DemodConfig(samplerate=230000)
T, L = 393216, 16384
rec = synth.make_stream(1000, 230000, 72000, f0_hz=1200.0)
buf = torch.empty((T * L, 2), dtype=torch.int16, device="cuda")
synth.generate_device([rec], T * L, out=buf.view(1, T * L, 2))
x = buf.view(T, L, 2)
d = Demodulator(cfg, T)
cap = d.max_symbols(L)
soft = torch.empty((T, cap, 2), dtype=torch.int8, device="cuda")
def run(xx, tag):
    for _ in range(2): d.process(xx, soft=soft)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(3): d.process(xx, soft=soft)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
    print(f"{tag}: {dt*1e3:.2f} ms  {T*L/dt/1e9:.1f} GS/s")
run(x, "distinct tiles (25.8 GB input)")
run(x[:1].expand(T, L, 2), "all lanes read the SAME tile (64 KB input, L2 resident)")
run(x[:4096].repeat(T // 4096, 1, 1) if False else x.view(96, 4096, L, 2)[0:1].expand(96, 4096, L, 2).reshape(T, L, 2) if False else x, "distinct again")
