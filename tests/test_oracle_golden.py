"""CPU tests: the oracle (CPU restatement) against the golden vectors recorded
from the REAL reference build (tests/golden/make_golden.py).  This is what pins
the oracle; the GPU tests then compare the HIP path with the oracle."""
from __future__ import annotations

import hashlib

import numpy as np
import pytest

import oracle_py as O
from conftest import load_npz
from golden_cases import BY_NAME, CASES, file_case_bytes, sha
from meteor_demod_amd import DemodConfig


@pytest.fixture(scope="module")
def runs():
    """Each case demodulated once by the oracle (with trace)."""
    out = {}
    for case in CASES:
        iq = case.generate()
        soft, trace, ev = O.oracle_demod(case.cfg, iq, want_trace=True)
        out[case.name] = (iq, soft, trace, ev)
    return out


@pytest.mark.parametrize("name", [c.name for c in CASES])
def test_generator_is_pinned(name, runs, manifest):
    """The deterministic synthetic input is byte-identical to the one the goldens were made from."""
    meta = manifest["cases"][name]
    iq = runs[name][0]
    assert iq.shape[0] == meta["n_samples"]
    assert sha(iq) == meta["input_sha256"]
    if meta["stored_input"]:
        assert np.array_equal(load_npz(name)["input"], iq)


@pytest.mark.parametrize("name", [c.name for c in CASES])
def test_oracle_soft_symbols_match_reference(name, runs, manifest):
    meta = manifest["cases"][name]
    _, soft, _, _ = runs[name]
    assert soft.shape[0] == meta["n_symbols"]
    assert sha(soft) == meta["soft_sha256"]          # every byte, not +-1 LSB
    if meta["stored_input"]:
        assert np.array_equal(load_npz(name)["soft"], soft)


@pytest.mark.parametrize("name", [c.name for c in CASES])
def test_oracle_trace_matches_reference(name, runs, manifest):
    """Per-symbol float trace (sample index, re, im, pll_freq, omega, gain, locked): bit-exact."""
    meta = manifest["cases"][name]
    _, _, trace, _ = runs[name]
    assert hashlib.sha256(trace.tobytes()).hexdigest() == meta["trace_sha256"]
    g = load_npz(name)
    assert trace[:: manifest["trace_step"]].tobytes() == g["trace_ckpt"].tobytes()
    assert trace[:256].tobytes() == g["trace_head"].tobytes()


@pytest.mark.parametrize("name", [c.name for c in CASES])
def test_oracle_lock_behaviour_matches_reference(name, runs, manifest):
    meta = manifest["cases"][name]
    _, _, trace, ev = runs[name]
    assert [list(e) for e in ev] == meta["lock_events"]
    first = next((e[0] for e in ev if e[1] == 1), -1)
    assert first == meta["first_lock_symbol"]
    assert int(trace[-1]["locked"]) == meta["final"]["locked"]
    assert float(trace[-1]["pll_freq"]) == meta["final"]["pll_freq"]


def test_lock_scenarios_cover_the_domain(manifest):
    """The fixture set holds: quick lock, sweep lock, never-lock, unlock->relock."""
    c = manifest["cases"]
    assert 0 <= c["c1_short"]["first_lock_symbol"] < 5000
    assert c["c1_lock1200"]["first_lock_symbol"] > 50000
    assert c["c1_never_locks"]["first_lock_symbol"] == -1
    assert [e[1] for e in c["c1_fade"]["lock_events"]] == [1, 0, 1]


@pytest.mark.parametrize("name", ["c1_short", "c3_short", "u8_short", "c1_fade"])
@pytest.mark.parametrize("blocks", [[1, 2, 3, 5, 64, 1000], [4097], [65, 129, 7]])
def test_oracle_block_chaining_is_invariant(name, blocks, runs):
    """Feeding a recording in blocks of any size gives the same bytes (state carries over)."""
    iq, soft, _, _ = runs[name]
    n = min(iq.shape[0], 30000)
    st = O.OracleStream(BY_NAME[name].cfg)
    parts, pos, k = [], 0, 0
    while pos < n:
        b = min(blocks[k % len(blocks)], n - pos)
        parts.append(st.run(iq[pos:pos + b])[0])
        pos += b
        k += 1
    got = np.concatenate(parts)
    assert np.array_equal(got, soft[:got.shape[0]])
    assert st.state.n_samples == n


def test_oracle_empty_and_tiny_inputs():
    cfg = DemodConfig(samplerate=230000)
    st = O.OracleStream(cfg)
    soft, _, ev = st.run(np.zeros((0, 2), dtype=np.int16))
    assert soft.shape == (0, 2) and ev == []
    soft, _, _ = st.run(np.zeros((1, 2), dtype=np.int16))
    assert soft.shape[0] == 0
    assert st.state.n_samples == 1


def test_quantiser_edges():
    """main.c:305-306: v/2, clamp to +-127, truncate toward zero."""
    q = O.lib().orc_quantise
    assert [q(v) for v in (0.0, 1.9, -1.9, 2.0, -2.0, 253.9, 254.0, 255.9, 1e9, -1e9, -253.9, -255.0)] == \
           [0, 0, 0, 1, -1, 126, 127, 127, 127, -127, -126, -127]


# ---- file-level model ---------------------------------------------------------

@pytest.mark.parametrize("name", ["file_wav_s16", "file_raw_u8", "file_wav_f32", "file_wav_oqpsk", "file_never_locks"])
def test_oracle_file_model_matches_reference_binary(name, manifest):
    """32 KiB-truncated reads, 1024-byte chunks gated on locked_once, double-length final flush."""
    meta = manifest["file_cases"][name]
    data = file_case_bytes(meta)
    assert hashlib.sha256(data).hexdigest() == meta["file_sha256"]
    cfg = DemodConfig(**meta["cfg"])
    body = data[44:] if meta["container"] == "wav" else data
    out = O.OracleStream(cfg).file_model(body, cfg.bps)
    ref = load_npz(name)["out"].tobytes()
    assert len(out) == meta["out_bytes"]
    assert out == ref


# ---- known-answer tables --------------------------------------------------------------------

@pytest.mark.parametrize("tag", ["c1", "c3", "c4", "odd"])
def test_oracle_rrc_table_matches_reference(tag, manifest):
    cfg = DemodConfig(**manifest["tables"][f"rrc_{tag}"]["cfg"])
    ref = load_npz("tables")[f"rrc_{tag}"]
    st = O.OracleStream(cfg)
    assert np.array_equal(st.rrc_table(), ref)
    assert sha(ref) == manifest["tables"][f"rrc_{tag}"]["sha256"]


def test_oracle_fast_sin_cos_match_reference():
    g = load_npz("tables")
    L = O.lib()
    x = g["sincos_x"]
    s = np.array([L.orc_fast_sin(float(v)) for v in x], dtype=np.float32)
    c = np.array([L.orc_fast_cos(float(v)) for v in x], dtype=np.float32)
    assert np.array_equal(s, g["sincos_sin"])
    assert np.array_equal(c, g["sincos_cos"])


def test_fast_sin_all_codes_properties():
    """All 65536 turn codes: odd symmetry about half a turn and the 0.003 error bound."""
    L = O.lib()
    codes = np.arange(-32768, 32768, dtype=np.int32)
    y = np.array([L.orc_fast_sin_code(int(c)) for c in codes], dtype=np.float64)
    true = np.sin(codes.astype(np.float64) * (2 * np.pi / 65536))
    assert np.max(np.abs(y - true)) < 0.0031
    assert np.all(np.abs(y) <= 1.0)


def test_oracle_tanh_lut_matches_reference_values():
    g = load_npz("tables")
    st = O.OracleStream(DemodConfig(samplerate=230000))
    assert np.array_equal(np.array(st.consts.tanh_lut[:], dtype=np.float32), g["tanh_lut"])


def test_cabsf_is_double_sqrt_of_double_sum():
    """glibc's cabsf/hypotf (agc.c:21) equals (float)sqrt((double)re^2 + (double)im^2): the
    formula the HIP kernel uses."""
    import ctypes as C
    libm = C.CDLL("libm.so.6")
    libm.hypotf.restype = C.c_float
    libm.hypotf.argtypes = [C.c_float, C.c_float]
    rng = np.random.default_rng(5)
    xy = np.concatenate([rng.normal(0, 200, (20000, 2)), rng.normal(0, 1e-3, (2000, 2)),
                         rng.normal(0, 3e4, (2000, 2))]).astype(np.float32)
    got = np.array([libm.hypotf(float(a), float(b)) for a, b in xy], dtype=np.float32)
    want = np.sqrt(xy[:, 0].astype(np.float64) ** 2 + xy[:, 1].astype(np.float64) ** 2).astype(np.float32)
    assert np.array_equal(got, want)
