"""Wall time of the C CLI in --tiled mode on a generated configs[1] WAV file of 2^[log2] samples (default 26), three runs, with the
breakdown the CLI (MDEMOD_CLI_TIMING) and the library (MDEMOD_RECORDING_DEBUG) print."""
import sys, time, subprocess, os
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch
from meteor_demod_amd import synth
from golden_cases import wav_header
n = 1 << int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 26
st = synth.make_stream(1000, 230000, 72000, f0_hz=1200.0, clock_ppm=-3.5)
iq = synth.generate_device([st], n)[0].cpu().numpy()
path = "/tmp/rec.wav"
with open(path, "wb") as f:
    f.write(wav_header(230000, 16, iq.nbytes)); f.write(iq.tobytes())
del iq
cli = "meteor_demod_amd/lib/meteor_demod_amd"
for args in (["-q", "--tiled"], ["-q", "--tiled"], ["-q", "--tiled"]):
    t0 = time.time()
    r = subprocess.run([cli, *args, "-o", "/tmp/out.s", path], capture_output=True, text=True, env=dict(os.environ, MDEMOD_CLI_TIMING="1", MDEMOD_RECORDING_DEBUG="1"))
    dt = time.time() - t0
    print(args, f"{dt:.3f} s rc {r.returncode} out {os.path.getsize('/tmp/out.s')} bytes", " | ".join(l for l in r.stderr.splitlines() if "host]" in l or "read " in l or "pilot" in l.lower()[:30] or "plan" in l or "assembled" in l))
