"""Wall time of the C CLI's --tiled mode over K generated configs[1] WAV files of 2^[log2] samples with --jobs 1, 2, 4, 8
(files of one GPU in flight at once).  Usage: cli_jobs_time.py [K=8] [log2=25]"""
import sys, time, subprocess
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from meteor_demod_amd import synth
from golden_cases import wav_header
K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 25)
paths = []
for k in range(K):
    st = synth.make_stream(1000 + k, 230000, 72000, f0_hz=1200.0 - 100.0 * k, clock_ppm=-3.5)
    iq = synth.generate_device([st], n)[0].cpu().numpy()
    p = f"/tmp/rec{k}.wav"
    with open(p, "wb") as f:
        f.write(wav_header(230000, 16, iq.nbytes)); f.write(iq.tobytes())
    paths.append(p)
cli = "meteor_demod_amd/lib/meteor_demod_amd"
subprocess.run([cli, "-q", "--tiled", paths[0]], capture_output=True)          # page cache, code objects
for jobs in (1, 2, 4, 8):
    t0 = time.time()
    r = subprocess.run([cli, "-q", "--tiled", "--jobs", str(jobs), *paths], capture_output=True, text=True)
    dt = time.time() - t0
    print(f"--jobs {jobs}: {K} files x 2^{n.bit_length() - 1} samples in {dt:.2f} s = {K * n / dt / 1e6:.0f} MS/s (rc {r.returncode})", flush=True)
