"""Definitions of the golden parity cases (shared by make_golden.py and the tests).

Every case is a deterministic synthetic recording (meteor_demod_amd.synth) plus
a demodulator configuration.  `tests/golden/make_golden.py` runs the REAL
reference (oracle/_ref, built from /root/reference) on each and commits the
outputs under tests/golden/; the tests then check the oracle (CPU, here) and
the HIP path (GPU box) against those files.  /root/reference is never needed
at test time.
"""
from __future__ import annotations

import hashlib
from dataclasses import dataclass, field

import numpy as np

from meteor_demod_amd import DemodConfig, synth


@dataclass
class Segment:
    n: int
    kw: dict


@dataclass
class Case:
    name: str
    cfg: DemodConfig
    segments: list            # list[Segment]; concatenated
    store_input: bool = False  # commit the input clip itself (small cases only)
    note: str = ""
    seed: int = 0
    blocks: list = field(default_factory=list)   # block sizes used by chaining tests

    def symrate(self):
        return self.cfg.symrate

    def generate(self) -> np.ndarray:
        parts = []
        n0 = 0
        for seg in self.segments:
            st = synth.make_stream(self.seed, self.cfg.samplerate, self.cfg.symrate, oqpsk=self.cfg.oqpsk,
                                   fmt=self.cfg.bps, **seg.kw)
            parts.append(synth.generate_host(st, seg.n, n0))
            n0 += seg.n
        return np.concatenate(parts, axis=0)


def wav_header(samplerate: int, bps: int, nbytes: int) -> bytes:
    """Canonical 44-byte RIFF header, the only layout the reference parses (wavfile.c:16-48)."""
    import struct
    fmt_tag = 3 if bps == 32 else 1
    return (b"RIFF" + struct.pack("<I", 36 + nbytes) + b"WAVE" + b"fmt " +
            struct.pack("<IHHIIHH", 16, fmt_tag, 2, samplerate, samplerate * 2 * bps // 8, 2 * bps // 8, bps) +
            b"data" + struct.pack("<I", nbytes))


def file_case_bytes(meta: dict) -> bytes:
    """Rebuild the input file of a MANIFEST file_case from its seed."""
    cfg = DemodConfig(**meta["cfg"])
    kw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in meta["stream"].items()}
    st = synth.make_stream(meta["seed"], cfg.samplerate, cfg.symrate, oqpsk=cfg.oqpsk, fmt=cfg.bps, **kw)
    body = synth.generate_host(st, meta["n_samples"]).tobytes()
    return (wav_header(cfg.samplerate, cfg.bps, len(body)) if meta["container"] == "wav" else b"") + body


def sha(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


C1 = dict(samplerate=230000)                                         # BASELINE configs[0..1]
C3 = dict(samplerate=230000, symrate=80000, oqpsk=True)              # configs[2]
C4 = dict(samplerate=1000000, rrc_order=64, interp_factor=8)         # configs[3]

CASES = [
    # --- short clips, input committed (F4) --------------------------------------------
    Case("c1_short", DemodConfig(**C1), [Segment(65536, dict(f0_hz=0.0, esn0_db=20.0))], True,
         "QPSK 72k defaults, 0 Hz offset: locks after ~2.7k symbols", seed=11),
    Case("c3_short", DemodConfig(**C3), [Segment(65536, dict(f0_hz=100.0, esn0_db=15.0))], True,
         "OQPSK 80k: half-symbol-offset timing path", seed=13),
    Case("u8_short", DemodConfig(bps=8, **C1), [Segment(65536, dict(f0_hz=300.0, rms=60.0, dc=(3.0, -2.0)))], True,
         "rtl_sdr style unsigned 8-bit input", seed=15),
    Case("f32_short", DemodConfig(bps=32, **C1), [Segment(32768, dict(f0_hz=-200.0, rms=0.5, dc=(0.01, -0.02), esn0_db=18.0))], True,
         "float input", seed=17),
    # --- longer clips, input pinned by SHA-256 of the deterministic generator (F5/F6) --
    Case("c1_lock1200", DemodConfig(**C1), [Segment(400000, dict(f0_hz=1200.0, clock_ppm=20.0))],
         note="+1.2 kHz: lock only after the 1e-6/symbol sweep reaches the carrier (~85k symbols)", seed=1001),
    Case("c1_neg_offset", DemodConfig(**C1), [Segment(300000, dict(f0_hz=-900.0, clock_ppm=-35.0, esn0_db=9.0))],
         note="negative offset: sweep goes up first, bounces at +fmax", seed=1002),
    Case("c1_never_locks", DemodConfig(**C1), [Segment(200000, dict(f0_hz=5000.0))],
         note="+5 kHz is outside fmax=0.3 rad/sym (3.44 kHz): never locks", seed=1003),
    Case("c1_fade", DemodConfig(**C1),
         [Segment(60000, dict(f0_hz=0.0, esn0_db=20.0)), Segment(90000, dict(f0_hz=0.0, rms=1.0, esn0_db=-30.0)),
          Segment(90000, dict(f0_hz=0.0, esn0_db=20.0))],
         note="signal / noise-only / signal: lock, unlock, relock events", seed=1004),
    Case("c1_narrow_d", DemodConfig(freq_max=0.05, pll_bw=2.0, **C1), [Segment(200000, dict(f0_hz=400.0))],
         note="-d / -b options: clamp at +-0.05 rad/sym, wider loop", seed=1005),
    Case("c3_lock1200", DemodConfig(**C3), [Segment(200000, dict(f0_hz=1200.0))],
         note="OQPSK 80k +1.2 kHz", seed=3001),
    Case("c4_os8", DemodConfig(**C4), [Segment(1500000, dict(f0_hz=1200.0))],
         note="1 MS/s, order 64, x8: 129 taps x 8 banks", seed=4001),
    Case("odd_cfg", DemodConfig(samplerate=144000, symrate=72000, rrc_order=17, interp_factor=3, bps=16),
         [Segment(100000, dict(f0_hz=-50.0, esn0_db=25.0))],
         note="non-default -f 17 -O 3 at 2 samples/symbol (generic kernel path)", seed=6001),
    # --- at most one input sample per symbol: the reference's per-sample loop fires more than once per sample and keeps only the
    #     LAST symbol of each (demod.c:33-47, 62-90: `*sample` and `ret` are overwritten) -------------------------------------
    Case("one_per_symbol", DemodConfig(samplerate=72000), [Segment(60000, dict(f0_hz=100.0, esn0_db=20.0))],
         note="-s 72000 at 72k symbols/s: exactly one sample per symbol, two firings inside one sample whenever the clock runs fast", seed=7001),
    Case("sub_sample", DemodConfig(samplerate=60000), [Segment(50000, dict(f0_hz=-150.0, esn0_db=20.0))],
         note="-s 60000 at 72k symbols/s: 0.83 samples per symbol, every sixth sample holds two firings: only the last survives", seed=7002),
    Case("sub_sample_oqpsk", DemodConfig(samplerate=64000, symrate=80000, oqpsk=True, interp_factor=4), [Segment(50000, dict(f0_hz=60.0, esn0_db=20.0))],
         note="OQPSK 80k in 64 kS/s -O 4: 0.4 samples per firing", seed=7003),
    # --- symbol rates at and above the INTERPOLATED rate (symrate >= fs x O): the clock word is 2 pi or more, every interpolated step
    #     fires (timing.c:32-57), the phase accumulator is never brought back under its threshold and climbs through the float binades
    #     (timing.c:79 takes 2 pi off, the step puts 2 pi x ratio on).  Nothing a receiver would be set to; the reference takes the
    #     options all the same (main.c:109-123) and so does mdemod_create, up to the bound in demod_host.cpp.  Round 4's cj_schedule
    #     never returned for ratios from 2 (OQPSK) / 4 (QPSK); ratios of the cases: 2, 3, 6, 4 (-O 2), 3.9, 4 ----------------------
    Case("sub_step_oqpsk_r2", DemodConfig(samplerate=36000, interp_factor=1, oqpsk=True), [Segment(30000, dict(f0_hz=50.0, esn0_db=20.0))],
         note="OQPSK 72k in 36 kS/s -O 1: symrate = 2 fs O, one half-symbol firing per step", seed=7101),
    Case("sub_step_oqpsk_r3", DemodConfig(samplerate=24000, interp_factor=1, oqpsk=True), [Segment(30000, dict(f0_hz=-40.0, esn0_db=20.0))],
         note="OQPSK 72k in 24 kS/s -O 1: symrate = 3 fs O", seed=7102),
    Case("sub_step_oqpsk_r6", DemodConfig(samplerate=12000, interp_factor=1, oqpsk=True), [Segment(30000, dict(f0_hz=20.0, esn0_db=20.0))],
         note="OQPSK 72k in 12 kS/s -O 1: symrate = 6 fs O (the accepted region ends at 8)", seed=7103),
    Case("sub_step_oqpsk_o2", DemodConfig(samplerate=9000, interp_factor=2, oqpsk=True), [Segment(30000, dict(f0_hz=10.0, esn0_db=20.0))],
         note="OQPSK 72k in 9 kS/s -O 2: symrate = 4 fs O on the edge of the accepted region (8 fs)", seed=7104),
    Case("sub_step_qpsk_r3p9", DemodConfig(samplerate=18462, interp_factor=1), [Segment(30000, dict(f0_hz=30.0, esn0_db=20.0))],
         note="QPSK 72k in 18.462 kS/s -O 1: symrate = 3.9 fs O", seed=7105),
    Case("sub_step_qpsk_r4", DemodConfig(samplerate=18000, interp_factor=1), [Segment(30000, dict(f0_hz=-25.0, esn0_db=20.0))],
         note="QPSK 72k in 18 kS/s -O 1: symrate = 4 fs O exactly, the edge of the accepted region", seed=7106),
]

BY_NAME = {c.name: c for c in CASES}
