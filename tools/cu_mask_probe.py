"""Can the serial head keep its speed while a bank of wave-tiles fills the GPU?  Pilot on a stream masked to one CU,
bank on a stream masked to the others (hipExtStreamCreateWithCUMask)."""
import sys, time, ctypes as C
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from meteor_demod_amd import DemodConfig, Demodulator, synth
hip = C.CDLL("libamdhip64.so")
def masked_stream(bits):
    words = (C.c_uint32 * 8)(*[sum(((b >> (32 * w + k)) & 1) << k for k in range(32)) for w in range(8) for b in [bits]])
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)
cfg = DemodConfig(samplerate=230000)
n = 1 << 24
iq = synth.generate_device([synth.make_stream(1000, 230000, 72000, f0_hz=1200.0, clock_ppm=-3.5)], n)[0]
T, L = int(sys.argv[1]) if len(sys.argv) > 1 else 2000, 150_000
all_bits = (1 << 256) - 1
for mode in ("plain streams", "pilot on CU 0 only, bank on the rest", "pilot on CUs 0-7 (one per XCD?), bank on the rest"):
    if mode.startswith("plain"):
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    elif "CU 0 only" in mode:
        s1, s2 = masked_stream(1), masked_stream(all_bits & ~1)
    else:
        s1, s2 = masked_stream(0xFF), masked_stream(all_bits & ~0xFF)
    with Demodulator(cfg, 1) as p, Demodulator(cfg, T) as b:
        softp = torch.empty((1, p.max_symbols(65536), 2), dtype=torch.int8, device="cuda")
        softb = torch.empty((T, b.max_symbols(L), 2), dtype=torch.int8, device="cuda")
        view = torch.as_strided(iq, (T, L, 2), (4096 * 2, 2, 1))
        def pilot(nblocks=6):
            t0 = time.perf_counter()
            with torch.cuda.stream(s1):
                for k in range(nblocks):
                    p.process(iq[k * 65536:(k + 1) * 65536].unsqueeze(0), soft=softp)
                    p.status()
            return time.perf_counter() - t0
        def bank():
            with torch.cuda.stream(s2):
                b.process(view, soft=softb)
        # warm
        pilot(1); bank(); torch.cuda.synchronize(); p.reset(); b.reset()
        tp = pilot(); torch.cuda.synchronize(); p.reset()
        t0 = time.perf_counter(); bank(); s2.synchronize(); tb = time.perf_counter() - t0; b.reset()
        t0 = time.perf_counter(); bank(); tp2 = pilot(); s2.synchronize(); tb2 = time.perf_counter() - t0
        print(f"{mode}: pilot alone {tp*1e3:.1f} ms, bank alone ({T} x {L}, {b.kernel_name.split('(')[0]}) {tb*1e3:.1f} ms; together: pilot {tp2*1e3:.1f} ms, bank {tb2*1e3:.1f} ms", flush=True)
