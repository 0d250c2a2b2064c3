"""K independent recordings through mdemod_demodulate_recording at once, one host thread and one HIP stream each: the serial
head of a recording occupies one wave, so the heads of the others and their tile banks overlap with it.  Prints the aggregate
rate against the one-at-a-time rate and checks that every result is byte-identical to the one the same recording gives alone.
Usage: recordings_concurrent.py [K=8] [log2 samples=26]"""
import sys
import threading
import time

sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch
from meteor_demod_amd import DemodConfig, synth
from meteor_demod_amd.recording import demodulate_recording_native

K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 26)
cfg = DemodConfig(samplerate=230000)
recs = [synth.generate_device([synth.make_stream(2000 + k, cfg.samplerate, cfg.symrate, f0_hz=1200.0 - 150.0 * k)], n)[0] for k in range(K)]
torch.cuda.synchronize()

demodulate_recording_native(cfg, recs[0][: 1 << 22])                     # first call: hipFFT plans, code objects
alone, t_alone = [], []
for k in range(K):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    soft, rep = demodulate_recording_native(cfg, recs[k])
    torch.cuda.synchronize(); t_alone.append(time.perf_counter() - t0)
    alone.append(soft.clone())
print(f"one at a time: {sum(t_alone):.3f} s for {K} recordings of 2^{n.bit_length() - 1} samples = {K * n / sum(t_alone) / 1e6:.0f} MS/s "
      f"(each {min(t_alone) * 1e3:.0f}-{max(t_alone) * 1e3:.0f} ms)")

out = [None] * K
err = []
def work(k):
    try:
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            soft, rep = demodulate_recording_native(cfg, recs[k])
            s.synchronize()
        out[k] = soft
    except Exception as e:                                                 # noqa: BLE001 - report and fail below
        err.append((k, repr(e)))
for rounds in range(2):
    th = [threading.Thread(target=work, args=(k,)) for k in range(K)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    assert not err, err
    same = all(out[k].shape == alone[k].shape and bool((out[k] == alone[k]).all()) for k in range(K))
    print(f"{K} threads: {dt:.3f} s = {K * n / dt / 1e6:.0f} MS/s ({sum(t_alone) / dt:.2f}x), results identical to the one-at-a-time ones: {same}")
    assert same
