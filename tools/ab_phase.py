"""What the spread of symbol phases over a wave's lanes costs: the same tiling with a tile length that is a whole number of symbols
(every lane of the recording's tiles at the same symbol phase, so every lane's FIR window sits at the same alignment) against the
bench's 16448 (phases spread evenly).  An upper bound for what sorting the streams of a launch by symbol phase could gain."""
import sys
sys.path.insert(0, ".")
import torch
from meteor_demod_amd import DemodConfig, Demodulator, synth
T = 393216
cases = [("c1 230k", DemodConfig(samplerate=230000), 16445), ("c3 oqpsk 230k", DemodConfig(samplerate=230000, symrate=80000, oqpsk=True), 16445 // 23 * 23),
         ("c4 1M -f64 -O8", DemodConfig(samplerate=1000000, rrc_order=64, interp_factor=8), 16375), ("1.024M", DemodConfig(samplerate=1024000), 16384),
         ("1.8M", DemodConfig(samplerate=1800000), 16450), ("3.2M", DemodConfig(samplerate=3200000), 16400), ("2.4M u8", DemodConfig(samplerate=2400000, bps=8), 16400)]
for name, cfg, La in cases:
    for L in (16448, La):
        sym = L * cfg.symrate / cfg.samplerate
        rec = synth.make_stream(2000, cfg.samplerate, cfg.symrate, oqpsk=cfg.oqpsk, f0_hz=1200.0, fmt=cfg.bps, **(dict(rms=40.0) if cfg.bps == 8 else {}))
        buf = torch.empty((T * L, 2), dtype=torch.uint8 if cfg.bps == 8 else torch.int16, device="cuda")
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
            print(f"  {name:16s} L={L} ({sym:9.3f} symbols per tile) {T * L / ms / 1e6:8.1f} GS/s", flush=True)
        del buf, x, soft
        torch.cuda.empty_cache()
