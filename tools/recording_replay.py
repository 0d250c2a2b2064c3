"""Replay one case of tools/recording_fuzz.py through the PYTHON stitcher (HIP bank) and print per-tile diagnostics.
Usage: recording_replay.py n_cases seed case"""
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import oracle_py as O
from meteor_demod_amd import DemodConfig, synth
from meteor_demod_amd.recording import RecordingDemodulator, agreement

n_cases, seed, only = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(seed)
for ci in range(n_cases):
    oqpsk = bool(rng.random() < 0.35)
    symrate = 80000 if oqpsk else 72000
    osf = float(rng.choice([2.5, 2.875, 3.1944, 3.6, 4.0, 6.0, 14.2]))
    samplerate = int(symrate * osf)
    bps = int(rng.choice([8, 16, 16, 32]))
    ramp = float(rng.choice([0.0, 10.0, -25.0, 40.0, -40.0]))
    f0 = float(rng.uniform(100.0, 900.0)) if ramp <= 0 else float(rng.uniform(-200.0, 600.0))
    ppm = float(rng.uniform(-30.0, 30.0))
    n = int(rng.integers(3_000_000, 9_000_000))
    amp = {8: dict(rms=40.0, dc=(1.5, -1.0)), 16: dict(rms=float(rng.choice([1500.0, 6000.0])) * min(1.0, 3.2 / osf)),
           32: dict(rms=0.25, dc=(0.001, -0.002))}[bps]
    esn0 = float(rng.choice([10.0, 12.0, 15.0]))
    kw = {}
    if rng.random() < 0.5:
        kw["tile_samples"] = int(rng.choice([32768 + 64, 50000 // 64 * 64, 131072 + 64]) * max(1.0, osf / 3.2)) // 64 * 64
    if ci != only:
        continue
    cfg = DemodConfig(samplerate=samplerate, symrate=symrate, oqpsk=oqpsk, bps=bps)
    st = synth.make_stream(1000 + ci, samplerate, symrate, f0_hz=f0, clock_ppm=ppm, esn0_db=esn0, doppler_hz_per_s=ramp,
                           oqpsk=oqpsk, fmt=bps, **amp)
    iq = synth.generate_device([st], n)[0]
    serial, tr, ev = O.oracle_demod(cfg, iq.cpu().numpy(), True)
    res = RecordingDemodulator(cfg, carrier_seed="spectrum", **kw).demodulate(iq)
    r = res.report
    a = agreement(res.soft.cpu().numpy(), serial)
    w = a.pop("windows")
    print(cfg, "esn0", esn0, amp, kw)
    print(a, "weak", r.weak_seams, "tiles", r.n_tiles, "pilot", r.pilot_symbols)
    print("rot", r.rotations)
    print("seam", r.seam_shifts)
    print("refine_rot", r.refine_rotations)
    print("low windows", [(i, round(x, 3)) for i, x in enumerate(w) if x < 0.9])
    print("tile first symbols", list(res.tile_first_symbol[:80]))
    cs = np.asarray(r.carrier_seeds)
    print("seed 2nd diff", np.round(np.diff(cs, 2) * 1e5, 1))
    from meteor_demod_amd.recording import demodulate_recording_native
    soft, rep = demodulate_recording_native(cfg, iq, carrier_seed="spectrum", **kw)
    out = soft.cpu().numpy()
    print("native len", len(out), "serial", len(serial), "python", len(res.soft), "seam_fixes", rep.seam_fixes)
    # where do native and serial part ways?  compare from the front and from the back
    m = min(len(out), len(serial))
    front = ((out[:m] >= 0) == (serial[:m] >= 0)).all(axis=1)
    back = ((out[-m:] >= 0) == (serial[-m:] >= 0)).all(axis=1)
    W = 4096
    fw = [front[i:i + W].mean() for i in range(0, m, W)]
    bw = [back[i:i + W].mean() for i in range(0, m, W)]
    print("front-aligned low windows", [(i, round(x, 3)) for i, x in enumerate(fw) if x < 0.99][:10])
    print("back-aligned low windows", [(i, round(x, 3)) for i, x in enumerate(bw) if x < 0.99][:10])
    print("tile first symbols (python)", [int(x) for x in res.tile_first_symbol])
