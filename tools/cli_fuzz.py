"""Soak test of the C host CLI (host/meteor_demod_amd.c): random option lines (-r -s -O -f -b -d -m --bps, WAV or raw
container, odd file lengths, several files per invocation) against the oracle's file model (orc_file_model: 32 KiB-truncated
input, lock-gated 1024-byte chunks, final flush).  Byte for byte.  Usage: cli_fuzz.py [n_cases] [seed]"""
import subprocess
import sys
import tempfile
import time
from pathlib import Path

sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import oracle_py as O
from golden_cases import wav_header
from meteor_demod_amd import DemodConfig, scale_freq_max, synth

ROOT = Path(__file__).resolve().parent.parent
CLI = ROOT / "meteor_demod_amd" / "lib" / "meteor_demod_amd"
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
DT = {8: np.uint8, 16: np.int16, 32: np.float32}
bad, skipped, t0 = [], 0, time.time()
with tempfile.TemporaryDirectory() as td:
    td = Path(td)
    for ci in range(n_cases):
        oqpsk = bool(rng.random() < 0.3)
        symrate = int(rng.choice([72000, 80000]))
        samplerate = int(symrate * float(rng.choice([2.5, 3.1944, 4.0, 6.0])))
        bps = int(rng.choice([8, 16, 16, 32]))
        order = int(rng.choice([16, 32, 32, 48]))
        interp = int(rng.choice([2, 4, 5, 5, 8]))
        pll_bw_arg = str(rng.choice(["0.5", "1", "1", "2", "2.9"]))
        pll_bw = float(int(float(pll_bw_arg)))                               # the reference's human_to_float() returns an int (utils.c:60-84)
        fmax_hz = float(rng.choice([0.0, 1500.0, 3500.0]))                 # 0 = option not given
        wav = bool(rng.random() < 0.6)
        nfiles = int(rng.choice([1, 1, 2, 3]))
        args = ["-q", "-B"]
        if symrate != 72000 or rng.random() < 0.3:
            args += ["-r", str(symrate) if rng.random() < 0.5 else f"{symrate // 1000}k"]
        if oqpsk:
            args += ["-m", "oqpsk"]
        if order != 32 or rng.random() < 0.3:
            args += ["-f", str(order)]
        if interp != 5 or rng.random() < 0.3:
            args += ["-O", str(interp)]
        if pll_bw_arg != "1":
            args += ["-b", pll_bw_arg]
        if fmax_hz:
            args += ["-d", str(int(fmax_hz))]
        if not wav:
            args += ["-s", str(samplerate), "--bps", str(bps)]
        cfg = DemodConfig(samplerate=samplerate, symrate=symrate, oqpsk=oqpsk, rrc_order=order, interp_factor=interp, pll_bw=pll_bw,
                          freq_max=scale_freq_max(fmax_hz, symrate) if fmax_hz else -1.0, bps=bps)
        try:
            if not np.isfinite(O.OracleStream(cfg).rrc_table()).all():
                skipped += 1
                continue
        except Exception:
            skipped += 1
            continue
        files, wants, ok_case = [], [], True
        for fi in range(nfiles):
            n = int(rng.integers(20_000, 400_000))
            amp = {8: dict(rms=40.0, dc=(1.5, -1.0)), 16: dict(rms=1500.0), 32: dict(rms=0.25, dc=(0.001, -0.002))}[bps]
            st = synth.make_stream(7000 + 10 * ci + fi, samplerate, symrate, f0_hz=float(rng.uniform(-300, 300)), esn0_db=14.0,
                                   oqpsk=oqpsk, fmt=bps, **amp)
            iq = synth.generate_host(st, n)
            data = iq.tobytes()
            tail = bytes(rng.integers(0, 256, int(rng.choice([0, 0, 1, 3, 1000])), dtype=np.uint8))     # ragged end of file
            body = data + tail
            path = td / f"c{ci}_{fi}.{'wav' if wav else 'raw'}"
            path.write_bytes((wav_header(samplerate, bps, len(body)) if wav else b"") + body)
            try:
                wants.append(O.OracleStream(cfg).file_model(body, bps))
            except RuntimeError:                                  # the reference's final flush is undefined here (ring_idx > 512)
                ok_case = False
                break
            files.append(path)
        if not ok_case:
            skipped += 1
            continue
        tag = f"case {ci}: {' '.join(args)} {'wav' if wav else 'raw'} fs={samplerate} bps={bps} files={nfiles}"
        one = ["-o", str(files[0]) + ".s"] if nfiles == 1 else []          # one file without -o gets the reference's LRPT_<date>.s name
        r = subprocess.run([str(CLI), *args, *one, *map(str, files)], capture_output=True, text=True, cwd=td)
        if r.returncode != 0:
            print(tag, "-> rc", r.returncode, r.stderr[-300:], flush=True)
            bad.append(tag)
            continue
        good = all(Path(str(p) + ".s").read_bytes() == w for p, w in zip(files, wants))
        print(tag, "->", "ok" if good else "FAIL", [len(w) for w in wants], flush=True)
        if not good:
            bad.append(tag)
print(f"{n_cases} cases in {time.time() - t0:.0f} s, skipped {skipped}, failures {len(bad)}")
for b in bad:
    print("  ", b)
sys.exit(1 if bad else 0)
