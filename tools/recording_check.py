"""One recording through mdemod_demodulate_recording vs the serial oracle, for the three single-GPU configurations.
Usage: recording_check.py [c1|c3|c4 ...] [log2=24] [key=value ...] (doppler=Hz/s clkramp=ppm/s f0=Hz rms= esn0= tile= settle= acquire= frame= repair= seed=pilot|spectrum)"""
import sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np, torch
import oracle_py as O
from meteor_demod_amd import DemodConfig, synth
from meteor_demod_amd.recording import AUTO, agreement, demodulate_recording_native

tags = [a for a in sys.argv[1:] if a in ("c1", "c3", "c4")] or ["c1", "c3", "c4"]
kv = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
log2 = int(kv.get("log2", 24))
CFG = {"c1": DemodConfig(samplerate=230000), "c3": DemodConfig(samplerate=230000, symrate=80000, oqpsk=True),
       "c4": DemodConfig(samplerate=1000000, rrc_order=64, interp_factor=8)}
for tag in tags:
    cfg = CFG[tag]
    n = 1 << (log2 + (1 if tag == "c4" else 0))
    rms = float(kv.get("rms", 2000.0 if tag == "c4" else 6000.0))
    st = synth.make_stream(1000 if tag == "c1" else 2000, cfg.samplerate, cfg.symrate, oqpsk=cfg.oqpsk, f0_hz=float(kv.get("f0", 1200.0)),
                           clock_ppm=-3.5 if tag == "c1" else 0.0, doppler_hz_per_s=float(kv.get("doppler", 0.0)), clock_ppm_per_s=float(kv.get("clkramp", 0.0)), esn0_db=float(kv.get("esn0", 12.0)), rms=rms)
    iq = synth.generate_device([st], n)[0]
    t0 = time.time(); serial = O.oracle_demod(cfg, iq.cpu().numpy())[0]; t_cpu = time.time() - t0
    osf = cfg.samplerate / cfg.symrate
    kw = dict(tile_samples=int(float(kv.get("tile", 0)) * osf), carrier_seed=kv.get("seed", "spectrum"), repair=bool(int(kv.get("repair", 1))))
    if "clkseed" in kv: kw["clock_seed"] = kv["clkseed"]
    if "margin" in kv: kw["pilot_margin_symbols"] = int(kv["margin"])
    if "pblock" in kv: kw["pilot_block"] = int(kv["pblock"])
    for k, name in (("settle", "settle_samples"), ("acquire", "acquire_samples"), ("frame", "frame_samples")):
        if k in kv: kw[name] = int(float(kv[k]) * osf)
    demodulate_recording_native(cfg, iq[: 1 << 21], **kw)
    torch.cuda.synchronize(); t0 = time.time()
    soft, rep = demodulate_recording_native(cfg, iq, **kw)
    torch.cuda.synchronize(); dt = time.time() - t0
    a = agreement(soft.cpu().numpy(), serial); w = np.array(a.pop("windows"))
    ex = int(rep.exact_symbols)
    exact_ok = bool((soft[:ex].cpu().numpy() == serial[:ex]).all())
    print(f"{tag}: {n} samples {dt*1e3:.0f} ms (pilot {rep.pilot_seconds*1e3:.0f} tiles {rep.tiles_seconds*1e3:.0f}; serial oracle {t_cpu:.1f} s) tiles {rep.n_tiles} x {rep.tile_samples} "
          f"len {a['len_stitched']}/{a['len_serial']} within1 {a['within_1lsb']:.5f} decisions {a['hard_decisions_equal']:.6f} worst {a['worst_window']:.3f} "
          f"exact prefix {ex} ok={exact_ok} | weak {rep.weak_seams} fixes {rep.seam_fixes} frame_misses {rep.frame_misses} repaired {rep.repaired_tiles} jumps {rep.rotation_jumps} "
          f"dr_rms {rep.frame_residual_rms:.3f} weak_carrier {rep.weak_carrier_tiles} weak_clock {rep.weak_clock_tiles} work {rep.samples_demodulated / n:.2f}x", flush=True)
    bad = np.flatnonzero(w < 0.98)
    if len(bad): print("   windows < 0.98:", [(int(i), round(float(w[i]), 3)) for i in bad[:24]], "of", len(w))
