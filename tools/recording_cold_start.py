import sys, time
sys.path.insert(0, '.')
from meteor_demod_amd import DemodConfig, synth
from meteor_demod_amd.recording import demodulate_recording_native
import torch
mode = sys.argv[1]
cfg = DemodConfig(samplerate=230000)
st = synth.make_stream(99, 230000, 72000, f0_hz=300.0, clock_ppm=-20.0, esn0_db=12.0)
iq = synth.generate_device([st], 16_000_000)[0]
torch.cuda.synchronize()
for i in range(2):
    t0 = time.time(); soft, rep = demodulate_recording_native(cfg, iq, carrier_seed=mode); torch.cuda.synchronize()
    print(mode, "call", i, f"total {time.time()-t0:.3f} s pilot {rep.pilot_seconds:.3f} tiles {rep.tiles_seconds:.3f}")
