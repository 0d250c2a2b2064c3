"""Latency of ONE recording demodulated as overlapped tiles (meteor_demod_amd/recording.py) on the GPU,
with the agreement against the serial oracle for the smaller sizes.  Usage: recording_bench.py [Msamples ...]"""
import sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np, torch
from meteor_demod_amd import DemodConfig, synth
from meteor_demod_amd.recording import RecordingDemodulator, agreement

cfg = DemodConfig(samplerate=230000)
sizes = [int(float(a) * 1e6) for a in sys.argv[1:]] or [16_000_000, 64_000_000, 207_000_000]
for n in sizes:
    st = synth.make_stream(99, 230000, 72000, f0_hz=-700.0, clock_ppm=-20.0, esn0_db=12.0)
    iq = synth.generate_device([st], n)[0]
    torch.cuda.synchronize()
    for refine in (True, False):
        rd = RecordingDemodulator(cfg, refine=refine)
        rd.demodulate(iq[:2_000_000])                      # warm the context / allocator
        torch.cuda.synchronize(); t0 = time.time()
        pilot, psoft, pend = rd._run_pilot(iq, __import__('meteor_demod_amd.recording', fromlist=['StitchReport']).StitchReport())
        torch.cuda.synchronize(); t_pilot = time.time() - t0; pilot.close()
        torch.cuda.synchronize(); t0 = time.time()
        res = rd.demodulate(iq)
        torch.cuda.synchronize(); dt = time.time() - t0
        r = res.report
        line = (f"n={n/1e6:.0f}M ({n/230000:.0f} s of signal) refine={refine}: total {dt*1e3:.0f} ms (pilot {t_pilot*1e3:.0f} ms for "
                f"{r.pilot_samples} samples, tiles {1e3*(dt-t_pilot):.0f} ms for {r.n_tiles} tiles), {n/dt/1e6:.0f} MS/s end to end, "
                f"work {r.samples_demodulated/n:.2f}x, weak seams {r.weak_seams}, seam fixes {sum(1 for s in r.seam_shifts if s)}")
        if n <= 64_000_000:
            import oracle_py as O
            t0 = time.time(); serial = O.oracle_demod(cfg, iq.cpu().numpy())[0]; tc = time.time() - t0
            a = agreement(res.soft.cpu().numpy(), serial)
            line += (f"; vs serial oracle ({tc:.1f} s on one core): len {a['len_stitched']}/{a['len_serial']}, "
                     f"+-1 LSB {100*a['within_1lsb']:.2f} %, decisions {100*a['hard_decisions_equal']:.4f} %, worst 4096-window {100*a['worst_window']:.1f} %")
        print(line, flush=True)
