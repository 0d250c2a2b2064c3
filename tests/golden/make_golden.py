#!/usr/bin/env python3
"""Generates tests/golden/*.npz + MANIFEST.json from the REAL reference.

Runs only in the development container (needs /root/reference, built into
oracle/_ref by oracle/Makefile — the reference's own sources compiled with the
strict -ffp-contract=off flags, nothing copied).  For every case in
tests/golden_cases.py it feeds the deterministic synthetic recording to
oracle/_ref/ref_harness (per-sample demod_qpsk/demod_oqpsk + getters) and records
the soft symbols, a decimated per-symbol trace and the lock transitions; for the
file-level cases it runs the reference's own CLI binary on a WAV/raw file.

What is committed is data only: inputs (small clips), expected outputs, hashes.
"""
from __future__ import annotations

import hashlib
import json
import subprocess
import sys
import tempfile
from dataclasses import asdict
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(HERE.parent))

import oracle_py as O                      # noqa: E402
from golden_cases import CASES, sha, wav_header        # noqa: E402
from meteor_demod_amd import DemodConfig, synth   # noqa: E402

TRACE_STEP = 512


def lock_events(trace) -> list:
    locked = trace["locked"].astype(np.int8)
    prev = np.concatenate([[0], locked[:-1]])
    idx = np.flatnonzero(locked != prev)
    return [[int(i), int(locked[i])] for i in idx]


def run_case(case) -> dict:
    iq = case.generate()
    soft, trace = O.ref_demod(case.cfg, iq, want_trace=True)
    ev = lock_events(trace)
    first = next((e[0] for e in ev if e[1] == 1), -1)
    arrays = {"trace_ckpt": trace[::TRACE_STEP], "trace_head": trace[:256]}
    if case.store_input:
        arrays["input"] = iq
        arrays["soft"] = soft
    np.savez_compressed(HERE / f"{case.name}.npz", **arrays)
    last = trace[-1]
    meta = {
        "name": case.name, "note": case.note, "cfg": asdict(case.cfg), "seed": case.seed,
        "segments": [{"n": s.n, **s.kw} for s in case.segments],
        "n_samples": int(iq.shape[0]), "input_sha256": sha(iq),
        "n_symbols": int(soft.shape[0]), "soft_sha256": sha(soft),
        "trace_sha256": hashlib.sha256(trace.tobytes()).hexdigest(),
        "first_lock_symbol": int(first), "lock_events": ev,
        "final": {"pll_freq": float(last["pll_freq"]), "omega": float(last["omega"]), "gain": float(last["gain"]),
                  "locked": int(last["locked"])},
        "stored_input": bool(case.store_input),
    }
    print(f"{case.name:16s} n={meta['n_samples']:8d} sym={meta['n_symbols']:7d} first_lock={first:7d} "
          f"events={len(ev)} final_locked={meta['final']['locked']}")
    return meta


# ---- file-level behaviour of the reference CLI ----------------------

FILE_CASES = [
    # name, cfg kwargs, stream kwargs, n_samples, container ("wav"/"raw"), extra CLI args
    ("file_wav_s16", dict(samplerate=230000), dict(f0_hz=0.0, esn0_db=20.0), 100000, "wav", []),
    ("file_raw_u8", dict(samplerate=230000, bps=8), dict(f0_hz=0.0, esn0_db=20.0, rms=60.0, dc=(2.0, -1.0)), 140000, "raw",
     ["-s", "230k", "--bps", "8"]),
    ("file_wav_f32", dict(samplerate=230000, bps=32), dict(f0_hz=0.0, esn0_db=20.0, rms=0.4, dc=(0.0, 0.0)), 60000, "wav", []),
    ("file_wav_oqpsk", dict(samplerate=230000, symrate=80000, oqpsk=True), dict(f0_hz=50.0, esn0_db=18.0), 120000, "wav",
     ["-m", "oqpsk", "-r", "80k"]),
    ("file_never_locks", dict(samplerate=230000), dict(f0_hz=5000.0), 70000, "wav", []),
]


def run_file_case(name, cfgkw, stkw, n, container, extra, seed) -> dict | None:
    cfg = DemodConfig(**cfgkw)
    # choose a length whose EOF flush is deterministic (ring_idx <= 512 bytes): trim until it is
    for trim in range(0, 64):
        nn = n - trim * 1024
        st = synth.make_stream(seed, cfg.samplerate, cfg.symrate, oqpsk=cfg.oqpsk, fmt=cfg.bps, **stkw)
        iq = synth.generate_host(st, nn)
        body = iq.tobytes()
        used = (len(body) // 32768) * 32768
        nsym = len(O.oracle_demod(cfg, np.frombuffer(body[:used], dtype=iq.dtype).reshape(-1, 2))[0])
        if 2 * (nsym % 512) <= 512:
            break
    else:
        raise RuntimeError("no deterministic length found")
    data = (wav_header(cfg.samplerate, cfg.bps, len(body)) if container == "wav" else b"") + body
    with tempfile.TemporaryDirectory() as td:
        inp, out = Path(td) / ("in." + container), Path(td) / "out.s"
        inp.write_bytes(data)
        subprocess.run([str(O.REF_BINARY), "-q", "-B", "-o", str(out), *extra, str(inp)], check=True,
                       capture_output=True)
        ref_out = out.read_bytes()
    # the input file is regenerated from the seed by the tests (pinned by file_sha256)
    np.savez_compressed(HERE / f"{name}.npz", out=np.frombuffer(ref_out, dtype=np.uint8))
    meta = {"name": name, "cfg": asdict(cfg), "seed": seed, "stream": {k: (list(v) if isinstance(v, tuple) else v) for k, v in stkw.items()},
            "n_samples": nn, "container": container, "cli_args": extra,
            "file_sha256": hashlib.sha256(data).hexdigest(), "file_bytes": len(data),
            "out_sha256": hashlib.sha256(ref_out).hexdigest(), "out_bytes": len(ref_out)}
    print(f"{name:16s} file={len(data)} B -> out={len(ref_out)} B")
    return meta


# ---- known-answer tables ------------------------------------------------------------------

def run_tables() -> dict:
    out = {}
    arrays = {}
    for tag, kw in [("c1", dict(samplerate=230000)), ("c3", dict(samplerate=230000, symrate=80000, oqpsk=True)),
                    ("c4", dict(samplerate=1000000, rrc_order=64, interp_factor=8)),
                    ("odd", dict(samplerate=144000, rrc_order=17, interp_factor=3))]:
        cfg = DemodConfig(**kw)
        arrays[f"rrc_{tag}"] = O.ref_rrc(cfg)
        out[f"rrc_{tag}"] = {"cfg": asdict(cfg), "sha256": sha(arrays[f"rrc_{tag}"])}
    # fast_sin / fast_cos over a dense float sweep that reaches every 16-bit turn code
    # several times plus the wrap region the PLL can produce (|x| < 2*pi + fmax + pi/2)
    x = np.concatenate([np.linspace(-9.0, 9.0, 1 << 20, dtype=np.float32),
                        np.random.default_rng(7).uniform(-9, 9, 1 << 18).astype(np.float32),
                        np.array([0.0, -0.0, 6.2831855, -6.2831855, 3.1415927, 1.5707964], dtype=np.float32)])
    s, c = O.ref_sincos(x)
    out["sincos"] = {"n": int(x.size), "x_sha256": sha(x), "sin_sha256": sha(s), "cos_sha256": sha(c)}
    arrays["sincos_x"] = x[:: 64]
    arrays["sincos_sin"] = s[:: 64]
    arrays["sincos_cos"] = c[:: 64]
    arrays["tanh_lut"] = np.array([np.float32(np.tanh(np.float64(i - 16))) for i in range(32)], dtype=np.float32)
    np.savez_compressed(HERE / "tables.npz", **arrays)
    return out


def main() -> None:
    if not O.have_ref():
        sys.exit("oracle/_ref is not built (needs /root/reference): run make -C oracle")
    manifest = {"generator": "tests/golden/make_golden.py", "reference_build": "gcc -std=gnu99 -O2 -ffp-contract=off",
                "trace_step": TRACE_STEP, "cases": {}, "file_cases": {}, "tables": {}}
    for case in CASES:
        manifest["cases"][case.name] = run_case(case)
    for i, fc in enumerate(FILE_CASES):
        manifest["file_cases"][fc[0]] = run_file_case(*fc, seed=7001 + i)
    manifest["tables"] = run_tables()
    (HERE / "MANIFEST.json").write_text(json.dumps(manifest, indent=1))
    total = sum(p.stat().st_size for p in HERE.glob("*.npz"))
    print(f"fixtures: {total / 1e6:.2f} MB")


if __name__ == "__main__":
    main()
