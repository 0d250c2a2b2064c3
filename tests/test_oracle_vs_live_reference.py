"""The CPU restatement (oracle/lrpt_oracle.c) against the REFERENCE ITSELF, run live: oracle/_ref/ref_harness is the reference's own
objects (compiled where they lie by oracle/Makefile, strict flags) behind a harness of ours.  The committed goldens pin 21 settings;
this draws a few hundred more - sample rates from 0.4 to 150 samples per symbol, every -O, filter orders, loop bandwidths, carrier
ranges, the three input formats, QPSK and OQPSK, the region where every interpolated step fires - and asks for the same soft bytes and
the same per-symbol floats (symbol, PLL frequency, clock word, AGC gain, lock flag), bit for bit.

Runs wherever oracle/_ref was built (this container: build() makes it; the prebuilt binary travels to the GPU box); skipped elsewhere.
Test infrastructure only: nothing here touches the product path."""
from __future__ import annotations

import numpy as np
import pytest

import oracle_py as O
from meteor_demod_amd import DemodConfig, synth

pytestmark = pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref (the reference built from /root/reference) is not here")


def _settings(seed: int, count: int):
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < count:
        oqpsk = bool(rng.random() < 0.4)
        symrate = int(rng.choice([72000, 80000, int(rng.integers(20000, 120000))]))
        kind = rng.random()
        if kind < 0.15:                                       # symbol rates at and above the interpolated rate: every step fires
            interp = int(rng.integers(1, 4))
            fs = int(symrate / (interp * float(rng.choice([0.5, 1.0, 1.5, 2.0, 3.0, 3.9]))))
        elif kind < 0.3:                                      # many samples per symbol
            interp = int(rng.integers(1, 6))
            fs = int(symrate * float(rng.uniform(12.0, 150.0)))
        else:                                                 # the everyday range
            interp = int(rng.integers(1, 9))
            fs = int(symrate * float(rng.uniform(1.3, 12.0)))
        fs = max(fs, 1000)
        cfg = DemodConfig(samplerate=fs, symrate=symrate, oqpsk=oqpsk, interp_factor=interp, rrc_order=int(rng.integers(4, 70)),
                          pll_bw=float(rng.choice([0.3, 1.0, 1.0, 2.5, 6.0])), freq_max=float(rng.choice([500.0, 3500.0, 3500.0, 9000.0])),
                          bps=int(rng.choice([8, 16, 16, 32])))
        out.append((cfg, int(rng.integers(0, 1 << 30)), float(rng.uniform(-2500.0, 2500.0)), float(rng.choice([6.0, 12.0, 25.0]))))
    return out


@pytest.mark.timeout(600)
@pytest.mark.parametrize("seed", [11, 12, 13, 14])
def test_restatement_equals_the_live_reference(seed):
    checked = 0
    for cfg, s, f0, esn0 in _settings(seed, 150):
        table = O.OracleStream(cfg).rrc_table()
        if not np.isfinite(table).all():                      # 0/0 on a tap (filter.c:86-93): the reference then indexes its LUT out of bounds
            continue
        st = synth.make_stream(s, cfg.samplerate, cfg.symrate, oqpsk=cfg.oqpsk, f0_hz=f0, esn0_db=esn0, fmt=cfg.bps,
                               **({"rms": 40.0} if cfg.bps == 8 else {"rms": 0.3} if cfg.bps == 32 else {}))
        n = int(min(60000, max(4000, 3000 * cfg.samplerate / cfg.symrate)))
        iq = synth.generate_host(st, n)
        want_soft, want_trace = O.ref_demod(cfg, iq, want_trace=True)
        got_soft, got_trace, _ev = O.oracle_demod(cfg, iq, want_trace=True)
        assert got_soft.shape == want_soft.shape and np.array_equal(got_soft, want_soft), (cfg, s)
        assert got_trace.tobytes() == want_trace.tobytes(), (cfg, s)        # every float of every symbol, bit for bit
        checked += 1
    assert checked >= 130


@pytest.mark.timeout(600)
@pytest.mark.skipif(not O.REF_BINARY.exists(), reason="the reference's own CLI binary is not here")
def test_file_model_equals_the_live_reference_binary(tmp_path):
    """The file level (whole 32 KiB reads, 1024-byte chunks behind the lock gate, the double-length final flush): the
    oracle's file model against the reference's own CLI binary on random files - WAV and raw, the three formats, QPSK / OQPSK, -O / -f /
    -b, lengths that end anywhere inside a read.  (Lengths whose last chunk is more than half full make the reference's final flush
    read past its ring, main.c:321: those are trimmed away here as in tests/golden/make_golden.py.)"""
    import subprocess
    from golden_cases import wav_header
    rng = np.random.default_rng(2025)
    checked = locked_cases = 0
    for case in range(44):
        oqpsk = bool(rng.random() < 0.35)
        bps = int(rng.choice([8, 16, 16, 32]))
        fs = int(rng.choice([230000, 230400, 288000, 144000, 460000]))
        symrate = 80000 if oqpsk else 72000
        interp, order, bw = int(rng.choice([5, 5, 3, 8])), int(rng.choice([32, 32, 17, 48])), float(rng.choice([1.0, 1.0, 2.0]))
        cfg = DemodConfig(samplerate=fs, symrate=symrate, oqpsk=oqpsk, bps=bps, interp_factor=interp, rrc_order=order, pll_bw=bw)
        if not np.isfinite(O.OracleStream(cfg).rrc_table()).all():            # (230.4 kS/s at 80k symbols/s -O 5: a zero denominator on a tap, filter.c:86-93 - inf in the table, NaN out of the FIR, and the reference indexes its LUT with it)
            continue
        never = case % 9 == 8                                                 # a carrier outside the loop's range: the gate never opens
        amp = {8: dict(rms=50.0, dc=(2.0, -1.0)), 16: {}, 32: dict(rms=0.4, dc=(0.0, 0.0))}[bps]
        st = synth.make_stream(5000 + case, fs, symrate, oqpsk=oqpsk, fmt=bps, f0_hz=6000.0 if never else float(rng.uniform(-150, 150)), esn0_db=18.0, **amp)
        n0 = int(rng.integers(40000, 260000))
        for trim in range(64):
            iq = synth.generate_host(st, n0 - trim * 997)
            body = iq.tobytes()
            used = (len(body) // 32768) * 32768
            nsym = len(O.oracle_demod(cfg, np.frombuffer(body[:used], dtype=iq.dtype).reshape(-1, 2))[0])
            if 2 * (nsym % 512) <= 512:
                break
        else:
            continue
        container = "wav" if rng.random() < 0.6 else "raw"
        args = ["-O", str(interp), "-f", str(order), "-b", repr(bw)] + (["-m", "oqpsk", "-r", str(symrate)] if oqpsk else [])
        if container == "raw":
            args += ["-s", str(fs), "--bps", str(bps)]
        data = (wav_header(fs, bps, len(body)) if container == "wav" else b"") + body
        inp, out = tmp_path / f"in{case}.{container}", tmp_path / f"out{case}.s"
        inp.write_bytes(data)
        subprocess.run([str(O.REF_BINARY), "-q", "-B", "-o", str(out), *args, str(inp)], check=True, capture_output=True, timeout=120)
        want = out.read_bytes()
        got = O.OracleStream(cfg).file_model(body, bps)
        assert got == want, (case, cfg, container, len(body), len(got), len(want))
        checked += 1
        locked_cases += len(want) > 2048
        inp.unlink(); out.unlink()
    assert checked >= 30 and locked_cases >= 20
