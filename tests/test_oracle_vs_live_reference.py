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
