"""Latency and agreement of mdemod_demodulate_recording against the tile size (one 32 M-sample recording, configs[1] signal).
Usage: recording_tile_sweep.py [tile_samples ...]"""
import sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import torch
import oracle_py as O
from meteor_demod_amd import DemodConfig, synth
from meteor_demod_amd.recording import agreement, demodulate_recording_native

cfg = DemodConfig(samplerate=230000)
n = 32_000_000
st = synth.make_stream(99, 230000, 72000, f0_hz=300.0, clock_ppm=-20.0, esn0_db=12.0, doppler_hz_per_s=10.0)
iq = synth.generate_device([st], n)[0]
serial = O.oracle_demod(cfg, iq.cpu().numpy())[0]
demodulate_recording_native(cfg, iq[:3_000_000], carrier_seed="spectrum")          # warm up
for tile in [int(a) for a in sys.argv[1:]] or [65600, 32832, 16448, 8256]:
    for pre in (0xFFFFFFFF, 8192 * 3, 16384 * 3):
        torch.cuda.synchronize(); t0 = time.time()
        soft, rep = demodulate_recording_native(cfg, iq, carrier_seed="spectrum", tile_samples=tile, settle_samples=pre)
        torch.cuda.synchronize(); dt = time.time() - t0
        a = agreement(soft.cpu().numpy(), serial); a.pop("windows")
        print(f"tile {tile} pre {pre}: {dt*1e3:.0f} ms (pilot {rep.pilot_seconds*1e3:.0f}, tiles {rep.tiles_seconds*1e3:.0f}), tiles {rep.n_tiles}, "
              f"len {a['len_stitched'] - a['len_serial']:+d}, decisions {a['hard_decisions_equal']:.6f}, within1 {a['within_1lsb']:.4f}, "
              f"worst {a['worst_window']:.3f}, weak {rep.weak_seams}, frame misses {rep.frame_misses}, repaired {rep.repaired_tiles}, jumps {rep.rotation_jumps}", flush=True)
