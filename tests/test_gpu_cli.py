"""GPU tests of the C host CLI (host/meteor_demod_amd.c): file in -> .s out must equal what the
reference's own binary produced (tests/golden/file_*.npz, recorded by make_golden.py)."""
from __future__ import annotations

import hashlib
import subprocess
from pathlib import Path

import numpy as np
import pytest

from conftest import ROOT, load_npz
from golden_cases import file_case_bytes

pytestmark = pytest.mark.gpu

CLI = ROOT / "meteor_demod_amd" / "lib" / "meteor_demod_amd"
FILE_CASES = ["file_wav_s16", "file_raw_u8", "file_wav_f32", "file_wav_oqpsk", "file_never_locks"]


@pytest.mark.parametrize("name", FILE_CASES)
def test_cli_output_equals_reference_binary(name, manifest, tmp_path, gpu_device):
    """Same command line as the reference (`-q -B -o out [flags] input`), byte-identical .s file:
    32 KiB-truncated input, lock-gated 1024-byte chunks, double-length final flush."""
    meta = manifest["file_cases"][name]
    data = file_case_bytes(meta)
    assert hashlib.sha256(data).hexdigest() == meta["file_sha256"]
    inp = tmp_path / ("in." + meta["container"])
    out = tmp_path / "out.s"
    inp.write_bytes(data)
    r = subprocess.run([str(CLI), "-q", "-B", "-o", str(out), *meta["cli_args"], str(inp)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = out.read_bytes()
    assert len(got) == meta["out_bytes"]
    assert got == load_npz(name)["out"].tobytes()


def test_cli_batch_of_files_equals_one_by_one(manifest, tmp_path, gpu_device):
    """Extension: several recordings in one invocation = one stream per file, same bytes as separate runs."""
    names = ["file_wav_s16", "file_never_locks"]
    paths = []
    for n in names:
        p = tmp_path / f"{n}.wav"
        p.write_bytes(file_case_bytes(manifest["file_cases"][n]))
        paths.append(p)
    r = subprocess.run([str(CLI), "-q", *map(str, paths)], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0, r.stderr
    for n, p in zip(names, paths):
        assert Path(str(p) + ".s").read_bytes() == load_npz(n)["out"].tobytes()


def test_cli_spreads_files_over_devices(manifest, tmp_path, gpu_device):
    """--devices a,b,...: one worker thread and one library context per listed GPU, file i on the (i mod G)-th of them, every
    worker its own batch and its own outputs (streams shard, nothing crosses GPUs).  Two workers on THIS box's one
    GPU exercise all of it (threads, contexts side by side); with two GPUs the same on both.  Outputs = the reference binary's."""
    from meteor_demod_amd import _capi
    names = ["file_wav_s16", "file_never_locks", "file_wav_s16", "file_never_locks", "file_wav_s16", "file_never_locks", "file_wav_s16", "file_wav_s16"]
    paths = []
    for k, n in enumerate(names):
        p = tmp_path / f"{k}_{n}.wav"
        p.write_bytes(file_case_bytes(manifest["file_cases"][n]))
        paths.append(p)
    lists = ["0,0"] + (["0,1"] if _capi.lib().mdemod_device_count() >= 2 else [])
    for devs in lists:
        for extra in ([], ["--tiled"]):
            for p in paths:
                Path(str(p) + ".s").unlink(missing_ok=True)
            r = subprocess.run([str(CLI), "-q", "--devices", devs, *extra, *map(str, paths)], capture_output=True, text=True, cwd=tmp_path)
            assert r.returncode == 0, r.stderr
            for n, p in zip(names, paths):
                got = Path(str(p) + ".s").read_bytes()
                if not extra:
                    assert got == load_npz(n)["out"].tobytes(), (devs, n)
                    # ... and what the same file gives when it is the only one (one worker, one stream)
                    single = tmp_path / "single.s"
                    r1 = subprocess.run([str(CLI), "-q", "-o", str(single), str(p)], capture_output=True, text=True, cwd=tmp_path)
                    assert r1.returncode == 0 and single.read_bytes() == got, (devs, n)
                else:                            # tiled: same length and lock gate (short files: the head is most of them)
                    assert len(got) == len(load_npz(n)["out"].tobytes()), (devs, n)


def test_cli_status_line_and_banner(manifest, tmp_path, gpu_device):
    """Without -q: `Input: ..., output: ...` (main.c:200), `Demodulator initialized` (main.c:219) and, with -B -R 0, one status
    line per block (main.c:249-261); the last one carries the final carrier and symbol-rate words of the oracle's run."""
    import re
    import oracle_py as O
    from meteor_demod_amd import DemodConfig
    meta = manifest["file_cases"]["file_wav_s16"]
    data = file_case_bytes(meta)
    inp, out = tmp_path / "in.wav", tmp_path / "out.s"
    inp.write_bytes(data)
    r = subprocess.run([str(CLI), "-B", "-R", "0", "-o", str(out), *meta["cli_args"], str(inp)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert f"Input: {inp}, output: {out}" in r.stdout and "Demodulator initialized" in r.stdout
    lines = re.findall(r"\(\s*([\d.]+)%\) Carrier:\s*([+-][\d.]+) Hz, Symbol rate: ([\d.]+) Hz, Locked: (Yes|No)", r.stdout)
    assert lines, r.stdout
    cfg = DemodConfig(**meta["cfg"])
    payload = data[44:]
    n = len(payload) // 32768 * 32768 // 4
    iq = np.frombuffer(payload[: n * 4], dtype=np.int16).reshape(-1, 2)
    ost = O.OracleStream(cfg)
    ost.run(iq)
    freq_hz = float(ost.state.pll_freq) * cfg.symrate / (2 * np.pi)
    rate_hz = float(ost.state.t_freq) * cfg.samplerate * cfg.interp_factor / (2 * np.pi)
    pct, carrier, rate, locked = lines[-1]
    assert abs(float(carrier) - freq_hz) < 0.06 and abs(float(rate) - rate_hz) < 0.06, (lines[-1], freq_hz, rate_hz)
    assert (locked == "Yes") == bool(ost.state.locked) and float(pct) > 99.0


def test_cli_full_screen_display(manifest, tmp_path, gpu_device):
    """On a terminal and without -B the reference draws its ncurses display (main.c:197,224-245); so does the C host: panes with the
    final loop words, `Demodulation complete`, a key to leave - and the same output file as with -B."""
    import re
    from ptyrun import run_on_pipes
    meta = manifest["file_cases"]["file_wav_s16"]
    inp, out = tmp_path / "in.wav", tmp_path / "out.s"
    inp.write_bytes(file_case_bytes(meta))
    rc, screen, err = run_on_pipes([str(CLI), "--tui", "-R", "0", "-o", str(out), *meta["cli_args"], str(inp)], cols=300, timeout=120)      # wide: the paths of the Input line must not wrap
    if "PLL status" not in screen:
        r = subprocess.run([str(CLI), "--tui-selftest"], stdin=subprocess.DEVNULL, capture_output=True, text=True)
        if "built without ncurses" in r.stderr:
            pytest.skip("no ncurses in this image")
    assert rc == 0, (err, screen[-2000:])
    for piece in (f"Input: {inp}, output: {out}", "Demodulator initialized", "PLL status: ", "Carrier freq", "Data in", "(100.0%)", "Data out",
                  "Demodulation complete", "Press any key to exit..."):
        assert piece in screen, (piece, screen[-3000:])
    assert out.read_bytes() == load_npz("file_wav_s16")["out"].tobytes()
    # the last PLL pane: the words the status line test pins (test_cli_status_line_and_banner), to the display's one decimal
    words = re.findall(r"([+-]\d+\.\d) Hz\s+(\d+\.\d) Hz", screen)
    assert words, screen[-3000:]
    r = subprocess.run([str(CLI), "-B", "-R", "0", "-o", str(tmp_path / "b.s"), *meta["cli_args"], str(inp)], capture_output=True, text=True)
    last = re.findall(r"Carrier:\s*([+-][\d.]+) Hz, Symbol rate: ([\d.]+) Hz", r.stdout)[-1]
    assert abs(float(words[-1][0]) - float(last[0])) < 0.11 and abs(float(words[-1][1]) - float(last[1])) < 0.11, (words[-1], last)


def test_cli_full_screen_display_q_ends_the_run(tmp_path, gpu_device):
    """q: the run stops after the block in flight (the reference's `done = 1`, main.c:226-229) and what was demodulated is on disk."""
    from ptyrun import run_on_pipes
    import torch
    from meteor_demod_amd import synth
    from golden_cases import wav_header
    n = 1 << 24                                   # 16 blocks of 4 MiB
    st = synth.make_stream(1000, 230000, 72000, f0_hz=300.0, clock_ppm=0.0)
    iq = synth.generate_device([st], n)[0].cpu().numpy()
    inp, out = tmp_path / "long.wav", tmp_path / "long.s"
    with open(inp, "wb") as f:
        f.write(wav_header(230000, 16, iq.nbytes)); f.write(iq.tobytes())
    rc, screen, err = run_on_pipes([str(CLI), "--tui", "-R", "0", "-o", str(out), str(inp)], script=((b"Data out", b"q"), (b"Press any key", b"x")), timeout=180)
    if "PLL status" not in screen:
        pytest.skip("no ncurses in this image")
    assert rc == 0, err
    assert "Demodulation complete" in screen
    full = int(n * 72000 / 230000) * 2
    assert 0 < out.stat().st_size < 0.9 * full, (out.stat().st_size, full)


def test_cli_stdout_mode_and_errors(manifest, tmp_path, gpu_device):
    meta = manifest["file_cases"]["file_wav_s16"]
    inp = tmp_path / "in.wav"
    inp.write_bytes(file_case_bytes(meta))
    r = subprocess.run([str(CLI), "--stdout", str(inp)], capture_output=True)
    assert r.returncode == 0 and r.stdout == load_npz("file_wav_s16")["out"].tobytes()
    raw = tmp_path / "in.raw"
    raw.write_bytes(b"\0" * 70000)
    r = subprocess.run([str(CLI), "-q", str(raw)], capture_output=True, text=True)      # raw without -s
    assert r.returncode == 1 and "sample rate" in r.stderr


def test_cli_reads_a_pipe_in_short_blocks(manifest, tmp_path, gpu_device):
    """`-` = stdin (the reference's README: pipe from an SDR), raw samples with -s/--bps, 256 KiB blocks.  Like the
    reference, the WAV probe consumes 44 bytes and the rewind (main.c:164-166) does nothing on a pipe: the stream starts
    at byte 44.  Compared with the oracle's file model and, where the reference binary is present, with the binary."""
    import oracle_py as O
    from meteor_demod_amd import DemodConfig
    meta = manifest["file_cases"]["file_raw_u8"]
    data = file_case_bytes(meta)
    out = tmp_path / "out.s"
    r = subprocess.run([str(CLI), "-q", "-B", "-o", str(out), *meta["cli_args"], "-"], input=data, capture_output=True)
    assert r.returncode == 0, r.stderr
    got = out.read_bytes()
    assert got == O.OracleStream(DemodConfig(**meta["cfg"])).file_model(data[44:], 8)
    if O.REF_BINARY.exists():
        ref_out = tmp_path / "ref.s"
        rr = subprocess.run([str(O.REF_BINARY), "-q", "-B", "-o", str(ref_out), *meta["cli_args"], "-"], input=data, capture_output=True)
        if rr.returncode == 0 and ref_out.exists():
            assert got == ref_out.read_bytes()


def test_cli_tiled_mode_single_file(tmp_path, gpu_device):
    """--tiled: one file on many lanes.  The .s file is the lock-gated stream of mdemod_demodulate_recording:
    its head is the reference's own bytes and the whole file agrees with the serial oracle's file to the loops'
    noise (hard decisions equal, same length)."""
    import oracle_py as O
    from golden_cases import wav_header
    from meteor_demod_amd import DemodConfig, synth
    cfg = DemodConfig(samplerate=230000)
    n = 3_000_000 // 8192 * 8192                                  # whole 32 KiB reads
    iq = synth.generate_host(synth.make_stream(21, 230000, 72000, f0_hz=300.0, esn0_db=12.0), n)
    inp, out = tmp_path / "in.wav", tmp_path / "out.s"
    inp.write_bytes(wav_header(230000, 16, iq.nbytes) + iq.tobytes())
    r = subprocess.run([str(CLI), "-B", "--tiled", "--tile-samples", "32768", "--pilot-margin", "100k", "-o", str(out), str(inp)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "tiles" in r.stderr and "first lock" in r.stderr
    got = np.frombuffer(out.read_bytes(), dtype=np.int8).reshape(-1, 2)
    want = np.frombuffer(O.OracleStream(cfg).file_model(iq.tobytes(), 16), dtype=np.int8).reshape(-1, 2)
    assert got.shape == want.shape
    head = 100_000
    assert np.array_equal(got[:head], want[:head])              # inside the pilot: the reference's bytes, same lock gate
    body = slice(0, len(got) - 512)                              # the final flush repeats stale ring bytes
    assert ((got[body] >= 0) == (want[body] >= 0)).all(axis=1).mean() > 0.9999
    assert (np.abs(got[body].astype(int) - want[body].astype(int)).max(axis=1) <= 1).mean() > 0.93
    # several files: each one tiled, outputs <input>.s, same bytes as the single-file run
    inp2 = tmp_path / "copy.wav"
    inp2.write_bytes(inp.read_bytes())
    r = subprocess.run([str(CLI), "-q", "--tiled", "--tile-samples", "32768", "--pilot-margin", "100k", str(inp), str(inp2)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert Path(str(inp) + ".s").read_bytes() == out.read_bytes() == Path(str(inp2) + ".s").read_bytes()
    # ... with several of them in flight at once (--jobs) or one after the other: the same files
    more = [tmp_path / f"m{i}.wav" for i in range(5)]
    for m in more:
        m.write_bytes(inp.read_bytes())
    for jobs in ("3", "1"):
        r = subprocess.run([str(CLI), "-q", "--tiled", "--jobs", jobs, "--tile-samples", "32768", "--pilot-margin", "100k", *map(str, more)],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        for m in more:
            assert Path(str(m) + ".s").read_bytes() == out.read_bytes(), (jobs, m)
            Path(str(m) + ".s").unlink()
    # OQPSK works too (state rotation pass): same length as the serial file, hard decisions equal
    cfg_o = DemodConfig(samplerate=230000, symrate=80000, oqpsk=True)
    iq_o = synth.generate_host(synth.make_stream(22, 230000, 80000, f0_hz=250.0, esn0_db=14.0, oqpsk=True), n)
    for cut in range(0, 64):          # the reference's final flush is only defined for <= 256 symbols left in its ring
        try:
            O.OracleStream(cfg_o).file_model(iq_o[: n - 8192 * cut].tobytes(), 16)
            iq_o = iq_o[: n - 8192 * cut]
            break
        except RuntimeError:
            continue
    inp.write_bytes(wav_header(230000, 16, iq_o.nbytes) + iq_o.tobytes())
    r = subprocess.run([str(CLI), "-q", "--tiled", "-m", "oqpsk", "-r", "80000", "--tile-samples", "32768", "--pilot-margin", "100k",
                        "-o", str(out), str(inp)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = np.frombuffer(out.read_bytes(), dtype=np.int8).reshape(-1, 2)
    want = np.frombuffer(O.OracleStream(cfg_o).file_model(iq_o.tobytes(), 16), dtype=np.int8).reshape(-1, 2)
    assert got.shape == want.shape
    body = slice(0, len(got) - 512)
    assert ((got[body] >= 0) == (want[body] >= 0)).all(axis=1).mean() > 0.9999


def test_cli_tiled_follows_doppler(tmp_path, gpu_device):
    """--tiled on a recording whose carrier drifts 40 Hz/s (a pass's Doppler ramp): the default per-tile spectral carrier
    seeds keep the stitched file on the serial oracle's decisions; --carrier-seed is parsed and rejects junk."""
    import oracle_py as O
    from golden_cases import wav_header
    from meteor_demod_amd import DemodConfig, synth
    cfg = DemodConfig(samplerate=230000)
    n = 9_000_000 // 8192 * 8192
    st = synth.make_stream(23, 230000, 72000, f0_hz=-250.0, clock_ppm=4.0, esn0_db=12.0, doppler_hz_per_s=40.0)
    iq = synth.generate_device([st], n)[0].cpu().numpy()
    for cut in range(0, 64):          # the reference's final flush is only defined for <= 256 symbols left in its ring
        try:
            want = O.OracleStream(cfg).file_model(iq[: n - 8192 * cut].tobytes(), 16)
            iq = iq[: n - 8192 * cut]
            break
        except RuntimeError:
            continue
    want = np.frombuffer(want, dtype=np.int8).reshape(-1, 2)
    inp, out = tmp_path / "in.wav", tmp_path / "out.s"
    inp.write_bytes(wav_header(230000, 16, iq.nbytes) + iq.tobytes())
    r = subprocess.run([str(CLI), "-q", "--tiled", "-o", str(out), str(inp)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = np.frombuffer(out.read_bytes(), dtype=np.int8).reshape(-1, 2)
    assert got.shape == want.shape
    body = slice(0, len(got) - 512)
    assert ((got[body] >= 0) == (want[body] >= 0)).all(axis=1).mean() > 0.9999
    assert (np.abs(got[body].astype(int) - want[body].astype(int)).max(axis=1) <= 1).mean() > 0.95
    r = subprocess.run([str(CLI), "-q", "--tiled", "--carrier-seed", "pilot", "-o", str(out), str(inp)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(CLI), "-q", "--tiled", "--carrier-seed", "bogus", "-o", str(out), str(inp)], capture_output=True, text=True)
    assert r.returncode != 0 and "carrier-seed" in r.stderr


def test_cli_tiled_short_file_is_the_serial_file(tmp_path, gpu_device):
    """A file shorter than the pilot: --tiled must give the bytes of the normal (exact) mode - the whole file is the
    serial head.  (It used to fail with "soft-symbol capacity too small": the pilot wanted room for one symbol per sample
    of a 65 536-sample block in a buffer sized for the file.)"""
    from golden_cases import wav_header
    from meteor_demod_amd import synth
    for n in (8192 * 3, 8192 * 6):                    # 15 400 symbols at most: the hand-over is 15 000 symbols after the lock
        iq = synth.generate_host(synth.make_stream(77, 230000, 72000, f0_hz=0.0, esn0_db=14.0), n)
        inp, a, b = tmp_path / "short.wav", tmp_path / "a.s", tmp_path / "b.s"
        inp.write_bytes(wav_header(230000, 16, iq.nbytes) + iq.tobytes())
        r1 = subprocess.run([str(CLI), "-q", "-o", str(a), str(inp)], capture_output=True, text=True)
        r2 = subprocess.run([str(CLI), "-q", "--tiled", "-o", str(b), str(inp)], capture_output=True, text=True)
        assert r1.returncode == 0 and r2.returncode == 0, (r1.stderr, r2.stderr)
        assert a.read_bytes() == b.read_bytes() and len(a.read_bytes()) > 0
