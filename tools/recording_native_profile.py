"""One recording through mdemod_demodulate_recording, for `rocprofv3 --kernel-trace --stats -- python3 tools/recording_native_profile.py [Msamples]`:
which kernels a stitched recording spends its GPU time in."""
import sys
sys.path.insert(0, '.')
from meteor_demod_amd import DemodConfig, synth
from meteor_demod_amd.recording import demodulate_recording_native

n = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 64_000_000
cfg = DemodConfig(samplerate=230000)
st = synth.make_stream(99, 230000, 72000, f0_hz=300.0, clock_ppm=-20.0, esn0_db=12.0, doppler_hz_per_s=10.0)
iq = synth.generate_device([st], n)[0]
soft, rep = demodulate_recording_native(cfg, iq, carrier_seed="spectrum")
print(f"{n} samples: {rep.n_symbols} symbols, {rep.n_tiles} tiles, pilot {rep.pilot_seconds:.3f} s, tiles {rep.tiles_seconds:.3f} s")
