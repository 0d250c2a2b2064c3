"""GPU parity tests (-m gpu): the HIP path, called through the C-ABI, against the
oracle and the committed golden vectors.  Bar: every soft-symbol byte identical
(not +-1 LSB), identical lock/unlock events, identical loop state."""
from __future__ import annotations

import hashlib

import numpy as np
import pytest

import oracle_py as O
from conftest import load_npz
from golden_cases import BY_NAME, CASES, sha
from meteor_demod_amd import DemodConfig, Demodulator, synth

pytestmark = pytest.mark.gpu

C1 = DemodConfig(samplerate=230000)


def _torch():
    import torch
    return torch


KERNEL_VARIANTS = {
    # MDEMOD_LAT=0: contexts with few streams would otherwise pick the latency kernel (one stream per wave) by themselves
    "v3-rot": {"MDEMOD_KERNEL": "", "MDEMOD_LAT": "0"},       # rotating register windows (the default wherever one fits)
    "v1-ring": {"MDEMOD_KERNEL": "v1", "MDEMOD_LAT": "0"},    # LDS ring (generic fallback, > 129 taps)
    "lat": {"MDEMOD_KERNEL": "", "MDEMOD_LAT": "1"},          # one stream per wave, register-window state layout
    "lat-v1-state": {"MDEMOD_KERNEL": "v1", "MDEMOD_LAT": "1"},   # ... on the ring kernel's state layout
}


@pytest.fixture(autouse=True)
def _lane_kernels_unless_asked(monkeypatch):
    """The tests of this module are about the lane-per-stream kernels and their geometries; most of them use a few dozen
    streams, for which a context would pick the latency kernel on its own.  `kernel_variant` (below) overrides this for its
    two latency variants; tests/test_gpu_recording.py and the tools run with the library's own choice."""
    monkeypatch.setenv("MDEMOD_LAT", "0")


@pytest.fixture(params=list(KERNEL_VARIANTS))
def kernel_variant(request, monkeypatch):
    """Every kernel implementation must produce the same bytes (selection knobs are env vars
    read by mdemod_create / launch)."""
    for k, v in KERNEL_VARIANTS[request.param].items():
        if v:
            monkeypatch.setenv(k, v)
        else:
            monkeypatch.delenv(k, raising=False)
    return request.param


def gpu_demod(cfg, blocks_per_stream, n_streams=None):
    """blocks_per_stream: list (per stream) of numpy [n,2] -> list of soft arrays, statuses, demod."""
    torch = _torch()
    ns = len(blocks_per_stream)
    n = blocks_per_stream[0].shape[0]
    assert all(b.shape[0] == n for b in blocks_per_stream)
    d = Demodulator(cfg, ns)
    x = torch.from_numpy(np.stack(blocks_per_stream)).cuda()
    soft = d.process(x)
    torch.cuda.synchronize()
    st = d.status()
    outs = [soft[s, : st[s].symbols_this_call].cpu().numpy() for s in range(ns)]
    return outs, st, d


# ---- every golden case, byte for byte ------------------------------------------------------

@pytest.mark.parametrize("name", [c.name for c in CASES])
def test_hip_matches_reference_golden_and_oracle(name, manifest, gpu_device, kernel_variant):
    case = BY_NAME[name]
    meta = manifest["cases"][name]
    iq = case.generate()
    assert sha(iq) == meta["input_sha256"]
    ost = O.OracleStream(case.cfg)
    want, trace, ev = ost.run(iq, want_trace=True)

    (got,), (st,), d = gpu_demod(case.cfg, [iq])
    # vs the reference's own output (golden)
    assert got.shape[0] == meta["n_symbols"]
    assert sha(got) == meta["soft_sha256"]
    if meta["stored_input"]:
        assert np.array_equal(got, load_npz(name)["soft"])
    # vs the oracle
    assert np.array_equal(got, want)
    # lock behaviour
    assert st.first_lock_symbol == meta["first_lock_symbol"]
    assert [list(e) for e in d.lock_events(0)] == meta["lock_events"] == [list(e) for e in ev]
    assert st.locked == meta["final"]["locked"] and st.locked_once == int(meta["first_lock_symbol"] >= 0)
    # status getters (pll_get_freq, mm_omega, agc_get_gain) as read after the last SAMPLE:
    # bit-exact floats (for OQPSK the AGC may have run once more after the last symbol)
    assert np.float32(st.pll_freq) == np.float32(ost.state.pll_freq) == trace[-1]["pll_freq"]
    assert np.float32(st.omega) == np.float32(ost.state.t_freq) == trace[-1]["omega"]
    assert np.float32(st.gain) == np.float32(ost.state.gain)
    assert st.n_samples == iq.shape[0] and st.n_symbols == want.shape[0] and st.overflow == 0
    d.close()


def test_full_loop_state_matches_oracle(gpu_device, kernel_variant):
    """Every field of the per-stream state after a run, incl. filter history."""
    for name in ("c1_short", "c3_short"):
        case = BY_NAME[name]
        iq = case.generate()[:20011]                      # odd length on purpose
        ost = O.OracleStream(case.cfg)
        ost.run(iq)
        (_,), _, d = gpu_demod(case.cfg, [iq])
        g, s = d.get_state(0), ost.state
        for a, b in [(g.agc_gain, s.gain), (g.agc_bias_re, s.bias.re), (g.agc_bias_im, s.bias.im),
                     (g.pll_phase, s.pll_phase), (g.pll_freq, s.pll_freq), (g.pll_err, s.pll_err),
                     (g.t_phase, s.t_phase), (g.t_freq, s.t_freq), (g.t_prev, s.t_prev),
                     (g.oqpsk_inphase, s.inphase)]:
            assert np.float32(a) == np.float32(b)
        assert (g.pll_locked, g.pll_locked_once, g.pll_updown, g.t_dual_state) == \
               (s.locked, s.locked_once, s.updown, s.dual_state)
        assert (g.n_samples, g.n_symbols, g.first_lock_symbol) == (s.n_samples, s.n_symbols, s.first_lock_symbol)
        hist = d.get_history(0)
        want = ost.history()
        assert np.array_equal(hist[-(case.cfg.taps - 1):], want[-(case.cfg.taps - 1):])
        d.close()


# ---- block chaining: state persists across calls like the reference's globals ----------------

@pytest.mark.parametrize("name", ["c1_short", "c3_short", "u8_short", "f32_short", "odd_cfg"])
@pytest.mark.parametrize("blocks", [[1, 2, 3, 5, 64, 1000, 4099], [8192], [0, 7, 0, 130, 1]])
def test_block_chaining_equals_one_shot(name, blocks, gpu_device, kernel_variant):
    torch = _torch()
    case = BY_NAME[name]
    iq = case.generate()[:24000]
    want = O.oracle_demod(case.cfg, iq)[0]
    with Demodulator(case.cfg, 1) as d:
        parts, pos, k = [], 0, 0
        while pos < iq.shape[0]:
            b = min(blocks[k % len(blocks)], iq.shape[0] - pos)
            k += 1
            if b == 0:      # an empty block is legal: state must not move
                soft = d.process(torch.from_numpy(np.zeros((1, 1, 2), dtype=iq.dtype)).cuda(), n_samples=0)
            else:
                soft = d.process(torch.from_numpy(iq[pos:pos + b][None]).cuda())
            torch.cuda.synchronize()
            m = d.status()[0].symbols_this_call
            parts.append(soft[0, :m].cpu().numpy())
            pos += b
        got = np.concatenate(parts)
        assert np.array_equal(got, want)
        assert d.status()[0].n_samples == iq.shape[0]


def test_state_export_import_continues_exactly(gpu_device, kernel_variant):
    """Checkpoint/hand-off: get_state+history from one context, set into another, continue."""
    torch = _torch()
    case = BY_NAME["c1_short"]
    iq = case.generate()[:30000]
    want = O.oracle_demod(case.cfg, iq)[0]
    cut = 12345
    with Demodulator(case.cfg, 1) as a, Demodulator(case.cfg, 3) as b:
        s1 = a.process(torch.from_numpy(iq[None, :cut]).cuda())
        torch.cuda.synchronize()
        m1 = a.status()[0].symbols_this_call
        b.set_state(2, a.get_state(0))
        b.set_history(2, a.get_history(0))
        x = torch.from_numpy(np.stack([iq[cut:]] * 3)).cuda()
        s2 = b.process(x)
        torch.cuda.synchronize()
        st = b.status()
        got = np.concatenate([s1[0, :m1].cpu().numpy(), s2[2, : st[2].symbols_this_call].cpu().numpy()])
        assert np.array_equal(got, want)
        assert st[2].n_samples == iq.shape[0] and st[2].n_symbols == want.shape[0]
        # streams 0,1 of b started cold: they are a different (valid) demodulation
        assert st[0].n_samples == iq.shape[0] - cut
        # a clock word the reference's timing loop cannot hold (timing.c:80-86: centre +- centre / 4096) is refused, like a carrier
        # word outside +-fmax: the kernels' symbol clock counts on the bound
        bad = a.get_state(0)
        centre = 2 * np.pi * case.cfg.symrate / (case.cfg.samplerate * case.cfg.interp_factor)
        for f in (centre * (1 + 1.5 / 4096), centre * (1 - 1.5 / 4096), 0.0, -centre):
            bad.t_freq = float(f)
            with pytest.raises(Exception, match="mdemod_set_state"):
                b.set_state(1, bad)
            with pytest.raises(Exception, match="mdemod_set_state_all"):
                b.set_state_all(bad)
        bad.t_freq = float(np.float32(centre * (1 + 0.9 / 4096)))
        b.set_state(1, bad)


# ---- batches: many independent streams, one per lane -----------------------------------------

def test_batch_of_distinct_streams_each_matches_oracle(gpu_device, kernel_variant):
    """BASELINE configs[4] in miniature: N independent recordings with different carrier
    offsets, clock errors and noise; every stream equals its own serial demodulation."""
    ns, n = 200, 9000
    streams = [synth.make_stream(5000 + i, 230000, 72000, f0_hz=(i % 13 - 6) * 450.0, clock_ppm=(i % 9 - 4) * 12.5,
                                 esn0_db=6.0 + (i % 5) * 4.0, rms=3000.0 + 40.0 * i) for i in range(ns)]
    iqs = [synth.generate_host(s, n) for s in streams]
    outs, st, d = gpu_demod(C1, iqs)
    for i in range(ns):
        want, tr, ev = O.oracle_demod(C1, iqs[i], want_trace=True)
        assert np.array_equal(outs[i], want), i
        assert np.float32(st[i].pll_freq) == tr[-1]["pll_freq"] and st[i].locked == tr[-1]["locked"]
    d.close()


def test_batch_chained_over_several_calls(gpu_device, kernel_variant):
    """Many streams x several blocks: per-stream state AND filter history must carry over for every
    stream of a batch (a history layout mismatch between prologue and epilogue only shows with > 1 stream)."""
    torch = _torch()
    ns, blocks = 130, [3000, 1, 4097, 2500]
    streams = [synth.make_stream(7000 + i, 230000, 72000, f0_hz=(i % 11 - 5) * 300.0, esn0_db=14.0) for i in range(ns)]
    iqs = [synth.generate_host(s, sum(blocks)) for s in streams]
    with Demodulator(C1, ns) as d:
        parts = [[] for _ in range(ns)]
        pos = 0
        for b in blocks:
            x = torch.from_numpy(np.stack([a[pos:pos + b] for a in iqs])).cuda()
            soft = d.process(x)
            torch.cuda.synchronize()
            st = d.status()
            for i in range(ns):
                parts[i].append(soft[i, : st[i].symbols_this_call].cpu().numpy())
            pos += b
        for i in range(ns):
            assert np.array_equal(np.concatenate(parts[i]), O.oracle_demod(C1, iqs[i])[0]), i


WIDE_CFGS = {
    "c4_s16": DemodConfig(samplerate=1000000, rrc_order=64, interp_factor=8),
    "c4_u8": DemodConfig(samplerate=1000000, rrc_order=64, interp_factor=8, bps=8),
    "oqpsk80k_1M": DemodConfig(samplerate=1000000, symrate=80000, oqpsk=True, rrc_order=64, interp_factor=8),
    "taps97_os6": DemodConfig(samplerate=500000, rrc_order=48, interp_factor=6),
    "edge_15_per_symbol": DemodConfig(samplerate=1075000, rrc_order=40, interp_factor=4),   # 14.93 samples per symbol (1080000 would put an RRC singularity 0/0 on a tap: NaN, UB in the reference)
    "taps65_slow_clock": DemodConfig(samplerate=460000, rrc_order=32, interp_factor=5),   # 65 taps at 6.4 samples/firing: mid geometry
    "defaults_1024k": DemodConfig(samplerate=1024000),                                    # RTL-SDR rate, default -f 32 -O 5: mid geometry
    "defaults_1024k_oqpsk_u8": DemodConfig(samplerate=1024000, oqpsk=True, bps=8),
    "defaults_2048k": DemodConfig(samplerate=2048000),                                    # 28.4 samples per symbol: far geometry
    "far_edge_u8": DemodConfig(samplerate=2150000, rrc_order=20, interp_factor=3, bps=8), # 29.9 samples per symbol
    "defaults_1024k_f32": DemodConfig(samplerate=1024000, bps=32),                        # float input: hybrid window
    "oqpsk_640k_f32": DemodConfig(samplerate=640000, symrate=80000, oqpsk=True, rrc_order=24, interp_factor=4, bps=32),
    "c4_f32": DemodConfig(samplerate=1000000, rrc_order=64, interp_factor=8, bps=32),      # float input, 129 taps: hybrid window (VGPRs + AccVGPRs)
    "oqpsk80k_1M_f32": DemodConfig(samplerate=1000000, symrate=80000, oqpsk=True, rrc_order=64, interp_factor=8, bps=32),
    "taps97_230k_f32": DemodConfig(samplerate=230000, rrc_order=48, interp_factor=5, bps=32),   # the long filter at the LRPT rate: 3.2 samples per firing
    "taps129_O12_f32": DemodConfig(samplerate=900000, rrc_order=64, interp_factor=12, bps=32),
    "defaults_2048k_f32": DemodConfig(samplerate=2048000, bps=32),                          # 28.4 samples per symbol, float input
    "taps129_2048k_f32": DemodConfig(samplerate=2048000, rrc_order=64, interp_factor=4, bps=32),   # ... with the long filter
    "taps129_2048k": DemodConfig(samplerate=2048000, rrc_order=64, interp_factor=4),              # the long filter at 28.4 samples per symbol, s16
    "taps97_1800k_u8": DemodConfig(samplerate=1800000, rrc_order=48, interp_factor=3, bps=8),
    "defaults_3200k": DemodConfig(samplerate=3200000),                                            # RTL-SDR's top rate: 44.4 samples per symbol
    "defaults_2400k_u8": DemodConfig(samplerate=2400000, bps=8),
    "defaults_3200k_f32": DemodConfig(samplerate=3200000, bps=32),                                # ... with float input: the 120-slot hybrid window
    "defaults_6000k": DemodConfig(samplerate=6000000),                                             # an Airspy's rates: 83 / 139 samples per symbol
    "defaults_10000k": DemodConfig(samplerate=10000000),
    "taps129_4000k": DemodConfig(samplerate=4000000, rrc_order=64, interp_factor=4),               # the long filter at 55.6 samples per symbol
    "oqpsk_8000k": DemodConfig(samplerate=8000000, symrate=80000, oqpsk=True),                     # 50 samples per firing
    "defaults_8000k_u8": DemodConfig(samplerate=8000000, bps=8),                                   # a HackRF's 8 bits
    "taps129_6000k_u8": DemodConfig(samplerate=6000000, rrc_order=64, interp_factor=3, bps=8),
    "defaults_6000k_f32": DemodConfig(samplerate=6000000, bps=32),
    "oqpsk_7200k_f32": DemodConfig(samplerate=7200000, symrate=80000, oqpsk=True, rrc_order=20, interp_factor=3, bps=32),   # 45 samples per firing
    # symbol rates at and above the interpolated rate (every step fires: round 5, the region whose schedule search hung) on the long filter
    "sub_step_wide_oqpsk_r2": DemodConfig(samplerate=36000, interp_factor=1, rrc_order=48, oqpsk=True),          # 97 taps, symrate = 2 fs O
    "sub_step_wide_qpsk_r4_u8": DemodConfig(samplerate=18000, interp_factor=1, rrc_order=64, bps=8),              # 129 taps, symrate = 4 fs O
    "sub_step_hybrid_oqpsk_r3_f32": DemodConfig(samplerate=12000, interp_factor=2, rrc_order=64, oqpsk=True, bps=32),   # float input, symrate = 3 fs O
}
WIDE_KERNEL = {"sub_step_wide_oqpsk_r2": "wide", "sub_step_wide_qpsk_r4_u8": "wide", "sub_step_hybrid_oqpsk_r3_f32": "hybrid", "c4_s16": "wide", "c4_u8": "wide", "oqpsk80k_1M": "wide", "taps97_os6": "wide", "edge_15_per_symbol": "wide",
               "taps65_slow_clock": "mid", "defaults_1024k": "mid", "defaults_1024k_oqpsk_u8": "mid", "defaults_2048k": "far",
               "far_edge_u8": "far", "defaults_1024k_f32": "mid", "oqpsk_640k_f32": "mid",
               "c4_f32": "hybrid", "oqpsk80k_1M_f32": "hybrid", "taps97_230k_f32": "hybrid", "taps129_O12_f32": "hybrid", "defaults_2048k_f32": "far-f32", "taps129_2048k_f32": "hybrid", "taps129_2048k": "wide-far", "taps97_1800k_u8": "wide-far", "defaults_3200k": "far-far", "defaults_2400k_u8": "far-far", "defaults_3200k_f32": "hyb-far", "oqpsk_7200k_f32": "hyb-far", "defaults_6000k": "gather", "defaults_10000k": "gather", "taps129_4000k": "gather", "oqpsk_8000k": "gather", "defaults_8000k_u8": "gather", "taps129_6000k_u8": "gather", "defaults_6000k_f32": "gather"}


@pytest.mark.parametrize("name", list(WIDE_CFGS))
def test_wide_window_batch_chained(name, gpu_device, monkeypatch):
    """The wide / mid / far / hybrid / gather geometries (up to 129 taps at up to 30 samples per firing, up to 65 at up to 46, and beyond) on the v3
    kernels: 70 distinct streams (more than one wave, every symbol phase) x chained blocks, byte-identical to the oracle, loop state
    included."""
    torch = _torch()
    cfg = WIDE_CFGS[name]
    generation = "v3"
    rms = {8: 50.0, 16: 5000.0, 32: 0.3}[cfg.bps]
    ns, blocks = 70, [5000, 3, 9000, 1, 2047]
    streams = [synth.make_stream(8100 + i, cfg.samplerate, cfg.symrate, f0_hz=(i % 9 - 4) * 350.0, clock_ppm=(i % 7 - 3) * 15.0,
                                 esn0_db=15.0, rms=rms, oqpsk=cfg.oqpsk, fmt=cfg.bps) for i in range(ns)]
    iqs = [synth.generate_host(s, sum(blocks)) for s in streams]
    with Demodulator(cfg, ns) as d:
        want_name = ("v3 rotating packed window, " if generation == "v3" and cfg.bps != 32 else "v2 register window, ") + WIDE_KERNEL[name]
        if WIDE_KERNEL[name] == "hybrid":
            want_name = "v3 hybrid window: float input, 129 taps" if generation == "v3" else "v1 LDS ring"
        elif WIDE_KERNEL[name] == "wide-far":
            want_name = "v3 rotating packed window, wide" if generation == "v3" else "v1 LDS ring"
        elif WIDE_KERNEL[name] == "far-far":
            want_name = "v3 rotating packed window, far" if generation == "v3" else "v1 LDS ring"
        elif WIDE_KERNEL[name] == "gather":
            want_name = "v3 gather" if generation == "v3" else "v1 LDS ring"
        elif WIDE_KERNEL[name] == "hyb-far":
            want_name = "v3 hybrid window, far" if generation == "v3" else "v1 LDS ring"
        elif WIDE_KERNEL[name] == "far-f32":
            want_name = "v3 hybrid window, mid" if generation == "v3" else "v1 LDS ring"
        elif cfg.bps == 32 and generation == "v3":
            want_name = "v3 hybrid window, mid"           # float input with up to 65 taps: the 96-slot window
        assert want_name in d.kernel_name, d.kernel_name
        parts = [[] for _ in range(ns)]
        pos = 0
        for b in blocks:
            soft = d.process(torch.from_numpy(np.stack([a[pos:pos + b] for a in iqs])).cuda())
            torch.cuda.synchronize()
            st = d.status()
            for i in range(ns):
                parts[i].append(soft[i, : st[i].symbols_this_call].cpu().numpy())
            pos += b
        for i in range(ns):
            ost = O.OracleStream(cfg)
            want = ost.run(iqs[i])[0]
            assert np.array_equal(np.concatenate(parts[i]), want), (name, i)
            assert np.float32(st[i].pll_freq) == np.float32(ost.state.pll_freq) and st[i].locked == ost.state.locked
            assert st[i].n_samples == sum(blocks)


def test_wide_window_multi_round_and_ragged(gpu_device):
    """Wide geometry with more tiles than are resident (131072 lanes) and ragged lengths."""
    torch = _torch()
    cfg = WIDE_CFGS["c4_s16"]
    T, L = 131072 + 3000, 1500
    one = synth.generate_device([synth.make_stream(77, cfg.samplerate, cfg.symrate, f0_hz=-600.0)], L)
    want = O.oracle_demod(cfg, one[0].cpu().numpy())[0]
    with Demodulator(cfg, T) as d:
        cap = d.max_symbols(L)
        soft = torch.zeros((T, cap, 2), dtype=torch.int8, device="cuda")
        d.process(one.expand(T, L, 2), soft=soft)
        torch.cuda.synchronize()
        assert bool((soft == soft[:1]).all())
        assert np.array_equal(soft[T - 1, : want.shape[0]].cpu().numpy(), want)
    lens = [0, 1, 127, 128, 129, 151, 152, 153, 700, 1499, 1500]
    buf = one[0].cpu().numpy()
    offsets = [1 + sum(lens[:i]) + i for i in range(len(lens))]          # odd, unaligned starts
    flat = np.zeros((offsets[-1] + lens[-1] + 8, 2), dtype=np.int16)
    for o, n in zip(offsets, lens):
        flat[o:o + n] = buf[:n]
    with Demodulator(cfg, len(lens)) as d:
        soft = torch.zeros((len(lens), d.max_symbols(max(lens)), 2), dtype=torch.int8, device="cuda")
        d.process_ragged(torch.from_numpy(flat).cuda(), torch.tensor(offsets, dtype=torch.int64).cuda(),
                         torch.tensor(lens, dtype=torch.int32).cuda(), soft)
        torch.cuda.synchronize()
        st = d.status()
        for i, n in enumerate(lens):
            w = O.oracle_demod(cfg, buf[:n])[0] if n else np.zeros((0, 2), np.int8)
            assert st[i].symbols_this_call == w.shape[0], (i, n)
            assert np.array_equal(soft[i, : w.shape[0]].cpu().numpy(), w), (i, n)


def test_multi_round_launch_is_deterministic_and_exact(gpu_device, kernel_variant):
    """More tiles than the GPU can hold at once (blocks that start after others have finished, on CUs
    that are still busy): every tile of an all-identical batch must give the same bytes, twice in a row,
    and equal the oracle."""
    torch = _torch()
    T, L = 262144 + 4096, 2048
    one = synth.generate_device([synth.make_stream(31, 230000, 72000, f0_hz=700.0)], L)
    x = one.expand(T, L, 2)
    want = O.oracle_demod(C1, one[0].cpu().numpy())[0]
    with Demodulator(C1, T) as d:
        cap = d.max_symbols(L)
        for rep in range(2):
            d.reset()
            soft = torch.zeros((T, cap, 2), dtype=torch.int8, device="cuda")
            d.process(x, soft=soft)
            torch.cuda.synchronize()
            assert bool((soft == soft[:1]).all()), f"launch {rep}: tiles differ"
            assert np.array_equal(soft[T - 1, : want.shape[0]].cpu().numpy(), want)
        assert all(s.symbols_this_call == want.shape[0] for s in d.status(T - 300, 300))


def test_result_is_independent_of_lane_and_neighbours(gpu_device):
    """The same recording placed in different lanes / waves / blocks, next to different
    neighbours, demodulates to the same bytes."""
    torch = _torch()
    n = 8000
    probe = synth.generate_host(synth.make_stream(77, 230000, 72000, f0_hz=250.0, esn0_db=15.0), n)
    rng = np.random.default_rng(0)
    other = [synth.generate_host(synth.make_stream(8000 + i, 230000, 72000, f0_hz=float(rng.integers(-3000, 3000))), n)
             for i in range(8)]
    ns = 500
    slots = [0, 1, 63, 64, 127, 191, 192, 255, 256, 499]
    batch = [other[i % 8] for i in range(ns)]
    for s in slots:
        batch[s] = probe
    outs, st, d = gpu_demod(C1, batch)
    want = O.oracle_demod(C1, probe)[0]
    for s in slots:
        assert np.array_equal(outs[s], want), s
    d.close()


def test_ragged_batch_with_empty_and_short_streams(gpu_device, kernel_variant):
    torch = _torch()
    lens = [0, 1, 3, 4, 5, 63, 64, 65, 66, 129, 1000, 4097, 12000, 0, 2, 7777]
    streams = [synth.make_stream(6000 + i, 230000, 72000, f0_hz=100.0 * i, esn0_db=18.0) for i in range(len(lens))]
    iqs = [synth.generate_host(s, n) if n else np.zeros((0, 2), np.int16) for s, n in zip(streams, lens)]
    pad = 3                                              # odd sample offsets: streams are NOT 16-byte aligned
    offsets, pos = [], pad
    for a in iqs:
        offsets.append(pos)
        pos += a.shape[0] + 1
    flat = np.zeros((pos + 8, 2), dtype=np.int16)
    for o, a in zip(offsets, iqs):
        flat[o:o + a.shape[0]] = a
    with Demodulator(C1, len(lens)) as d:
        cap = d.max_symbols(max(lens))
        soft = torch.zeros((len(lens), cap, 2), dtype=torch.int8, device="cuda")
        d.process_ragged(torch.from_numpy(flat).cuda(), torch.tensor(offsets, dtype=torch.int64).cuda(),
                         torch.tensor(lens, dtype=torch.int32).cuda(), soft)
        torch.cuda.synchronize()
        st = d.status()
        for i, a in enumerate(iqs):
            want = O.oracle_demod(C1, a)[0] if a.shape[0] else np.zeros((0, 2), np.int8)
            assert st[i].symbols_this_call == want.shape[0], i
            assert st[i].n_samples == lens[i]
            assert np.array_equal(soft[i, : want.shape[0]].cpu().numpy(), want), i


def test_soft_capacity_overflow_is_reported_not_fatal(gpu_device, kernel_variant):
    torch = _torch()
    iq = BY_NAME["c1_short"].generate()[:10000]
    want = O.oracle_demod(C1, iq)[0]
    with Demodulator(C1, 1) as d:
        soft = torch.full((1, 100, 2), 99, dtype=torch.int8, device="cuda")
        d.process(torch.from_numpy(iq[None]).cuda(), soft=soft)
        torch.cuda.synchronize()
        st = d.status()[0]
        assert st.overflow == 1 and st.symbols_this_call == want.shape[0] and st.n_symbols == want.shape[0]
        assert np.array_equal(soft[0].cpu().numpy(), want[:100])


def test_host_buffer_path(gpu_device):
    """mdemod_process_host: host pointers in, host soft symbols out (PCIe inclusive)."""
    iqs = [BY_NAME["c1_short"].generate()[:n] for n in (5000, 0, 12001)]
    with Demodulator(C1, 3) as d:
        outs = d.process_host(iqs)
        for a, o in zip(iqs, outs):
            want = O.oracle_demod(C1, a)[0] if a.shape[0] else np.zeros((0, 2), np.int8)
            assert np.array_equal(o, want)


def test_host_buffer_path_pipelined_sub_blocks(gpu_device):
    """Large host batches run as a pipeline of chained sub-blocks (csrc/host_pipe.cpp): bytes, "this call" counters and
    lock events must be those of ONE call, and a second call must chain exactly."""
    ns, n = 96, 600_000                                   # 230 MB of input -> 7 sub-blocks
    streams = [synth.make_stream(9000 + i, 230000, 72000, f0_hz=(i % 5) * 60.0, esn0_db=14.0) for i in range(4)]
    base = [synth.generate_host(s, 2 * n) for s in streams]
    lens = [n - 1000 * (i % 7) for i in range(ns)]        # ragged
    with Demodulator(C1, ns) as d:
        outs1 = d.process_host([base[i % 4][: lens[i]] for i in range(ns)])
        st1 = d.status()
        ev1 = [d.lock_events(i) for i in range(4)]
        outs2 = d.process_host([base[i % 4][lens[i]: lens[i] + 50_000] for i in range(ns)])
        for i in range(0, ns, 5):
            ost = O.OracleStream(C1)
            w1, _, e1 = ost.run(base[i % 4][: lens[i]])
            w2 = ost.run(base[i % 4][lens[i]: lens[i] + 50_000])[0]
            assert np.array_equal(outs1[i], w1) and np.array_equal(outs2[i], w2), i
            assert st1[i].symbols_this_call == w1.shape[0] and st1[i].n_samples == lens[i]
            assert st1[i].lock_events_this_call == len(e1)
            if i < 4:
                assert ev1[i] == e1[:32] and len(e1) >= 1


@pytest.mark.parametrize("fmt", [16, 8, 32])
def test_host_buffer_path_from_pinned_rows(fmt, gpu_device):
    """mdemod_pin_host_buffer: a batch inside a pinned range, rows of equal length one stride apart, goes to the GPU straight from
    the caller's pages (one 2-D copy per sub-block, csrc/host_pipe.cpp) - the bytes are those of the staged path and of the oracle;
    ragged or scattered batches in the same context fall back to the staged path; chained calls stay exact across both."""
    import ctypes as C
    from meteor_demod_amd import _capi
    cfg = DemodConfig(samplerate=230000, bps=fmt)
    ns, n, pad = 80, 700_000 if fmt != 32 else 350_000, 24         # > 32 MiB of input per call: several sub-blocks; rows `pad` samples apart
    kw = dict(rms=60.0, dc=(3.0, -2.0)) if fmt == 8 else (dict(rms=0.4, dc=(0.01, -0.02)) if fmt == 32 else {})
    streams = [synth.make_stream(9100 + i, 230000, 72000, f0_hz=(i % 5) * 50.0, esn0_db=15.0, fmt=fmt, **kw) for i in range(4)]
    base = [synth.generate_host(s, n + 40_000) for s in streams]
    big = np.zeros((ns, n + pad, 2), dtype=base[0].dtype)          # ONE allocation: what a host reading a batch into one buffer has
    for i in range(ns):
        big[i, :n] = base[i % 4][:n]
    with Demodulator(cfg, ns) as d, Demodulator(cfg, ns) as ref:
        d.pin_host(big)
        with pytest.raises(_capi.MdemodError):
            d.pin_host(big[3:5])                                     # overlaps a pinned range
        outs = d.process_host([big[i, :n] for i in range(ns)])     # uniform rows inside the pin: the direct path
        want = ref.process_host([base[i % 4][:n].copy() for i in range(ns)])     # scattered copies: the staged path
        for i in range(ns):
            assert np.array_equal(outs[i], want[i]), i
        for i in range(4):
            assert np.array_equal(outs[i], O.oracle_demod(cfg, base[i][:n])[0]), i
        # second call, ragged this time (staged although inside the pin), chained on the first: still the oracle's bytes
        lens = [30_000 - 500 * (i % 9) for i in range(ns)]
        tail = np.zeros((ns, 40_000, 2), dtype=base[0].dtype)
        for i in range(ns):
            tail[i] = base[i % 4][n:]
        outs2 = d.process_host([tail[i, : lens[i]] for i in range(ns)])
        for i in range(0, ns, 7):
            ost = O.OracleStream(cfg)
            ost.run(base[i % 4][:n])
            assert np.array_equal(outs2[i], ost.run(base[i % 4][n: n + lens[i]])[0]), i
        d.unpin_host(big)
        with pytest.raises(_capi.MdemodError):
            d.unpin_host(big)                                        # not pinned any more
        d.pin_host(big)                                              # and again: destroy unpins what is left


def test_host_buffer_path_from_memory_that_is_pinned_already(gpu_device):
    """mdemod_pin_host_buffer on memory the caller got pinned from the runtime (torch's pinned allocator = hipHostMalloc): accepted, the
    rows are copied from where they are, and unpinning / destroying the context leaves the caller's allocation alone."""
    torch = _torch()
    ns, n = 64, 1 << 16
    buf_t = torch.empty((ns, n, 2), dtype=torch.int16).pin_memory()
    buf = buf_t.numpy()
    st = synth.make_stream(31, 230000, 72000, f0_hz=120.0, esn0_db=15.0)
    one = synth.generate_host(st, n + ns)
    for s in range(ns):
        buf[s] = one[s: s + n]
    want = [O.oracle_demod(C1, np.ascontiguousarray(buf[s]))[0] for s in (0, 17, 63)]
    for rounds in range(2):                                              # the second context pins the same memory again
        with Demodulator(C1, ns) as d:
            d.pin_host(buf)
            out = d.process_host([buf[s] for s in range(ns)])
            if rounds == 0:
                d.unpin_host(buf)                                        # ... and the first one lets go by hand, the second by closing
        for k, s in enumerate((0, 17, 63)):
            assert np.array_equal(out[s], want[k]), (rounds, s)
    assert buf_t.is_pinned() and int(buf_t[5, 7, 1]) == int(one[5 + 7, 1])   # still there, still the caller's


def test_pinning_that_covers_only_a_part_of_the_range_is_refused(gpu_device):
    """mdemod_pin_host_buffer took memory as "pinned anyway" on the evidence of its FIRST byte and recorded the caller's
    full length - the direct path would then hand hipMemcpy2DAsync pages nobody locked.  Now: a range whose HEAD is somebody's
    registration that stops short of its end is refused (MDEMOD_ERR_PARAM, mdemod_last_error says why; this runtime answers
    hipMemGetAddressRange for registered memory with a NULL base, so every page is asked); the rows still go through the staged path
    and give the oracle's bytes.  A range whose TAIL is registered starts on pageable memory: the library registers all of it
    itself (the runtime allows overlapping registrations) and the direct path gives the oracle's bytes.  The caller's own
    registration is left alone either way."""
    from meteor_demod_amd import _capi
    torch = _torch()
    rt = torch.cuda.cudart()
    ns, n = 32, 1 << 15
    raw = np.zeros(ns * n * 2 + 8192, dtype=np.int16)
    off = (-raw.ctypes.data % 4096) // 2                                   # page-aligned start
    buf = raw[off: off + ns * n * 2].reshape(ns, n, 2)
    st = synth.make_stream(77, 230000, 72000, f0_hz=90.0, esn0_db=15.0)
    one = synth.generate_host(st, n + ns)
    for s in range(ns):
        buf[s] = one[s: s + n]
    want = {s: O.oracle_demod(C1, np.ascontiguousarray(buf[s]))[0] for s in (0, 13, ns - 1)}
    half = buf.nbytes // 2
    for lo in (0, half):                                                     # the head, then the tail, registered by the caller
        assert int(rt.cudaHostRegister(buf.ctypes.data + lo, half, 0)) == 0
        try:
            with Demodulator(C1, ns) as d:
                if lo == 0:
                    with pytest.raises(_capi.MdemodError) as e:
                        d.pin_host(buf)
                    assert e.value.code == _capi.MDEMOD_ERR_PARAM and "mdemod_pin_host_buffer" in e.value.detail, e.value.detail
                else:
                    d.pin_host(buf)                                          # pageable first byte: registered as a whole by the library
                out = d.process_host([buf[s] for s in range(ns)])            # staged (refused) or direct (pinned): the same bytes
                for s, w in want.items():
                    assert np.array_equal(out[s], w), (lo, s)
        finally:
            assert int(rt.cudaHostUnregister(buf.ctypes.data + lo)) == 0
    with Demodulator(C1, ns) as d:                                           # nobody's registration left: the library's own pin works
        d.pin_host(buf)
        out = d.process_host([buf[s] for s in range(ns)])
        assert np.array_equal(out[13], want[13])


def test_more_symbols_than_the_nominal_rate(gpu_device):
    """While the symbol clock drains a large phase excursion (full-scale burst after silence, wide loop) it fires on
    every sample: more symbols than samples * symrate / samplerate.  mdemod_max_symbols is the hard bound (one per
    sample), the device path must not flag overflow and the host path must hand back every symbol (its copy-out uses the
    nominal pitch and falls back to a 2-D copy here)."""
    torch = _torch()
    cfg = DemodConfig(samplerate=575303, pll_bw=3000.0, symrate=144000, interp_factor=1, rrc_order=48, freq_max=0.3)
    rng = np.random.default_rng(1)
    a = rng.choice(np.array([-32767, 32767], dtype=np.int16), size=(9000, 2))
    a[:5000] = 0
    ost = O.OracleStream(cfg)
    w1 = ost.run(a[:5000])[0]
    w2 = ost.run(a[5000:5500])[0]
    nominal = int(500 * 144000 / 575303 * 1.01) + 16
    assert len(w2) > nominal                                          # the second block really exceeds the nominal bound
    quiet = synth.generate_host(synth.make_stream(8, cfg.samplerate, cfg.symrate, f0_hz=100.0), 5500)
    with Demodulator(cfg, 3) as d:                                    # stream 1 is the burst, 0 and 2 are ordinary signals
        assert d.max_symbols(500) >= 500
        d.process(torch.from_numpy(np.stack([quiet[:5000], a[:5000], quiet[:5000]])).cuda())
        soft = d.process(torch.from_numpy(np.stack([quiet[5000:], a[5000:5500], quiet[5000:]])).cuda())
        torch.cuda.synchronize()
        st = d.status()
        assert st[1].overflow == 0 and st[1].symbols_this_call == len(w2)
        assert np.array_equal(soft[1, : len(w2)].cpu().numpy(), w2)
    with Demodulator(cfg, 3) as d:
        o1 = d.process_host([quiet[:5000], a[:5000], quiet[:5000]])
        o2 = d.process_host([quiet[5000:], a[5000:5500], quiet[5000:]])
        assert np.array_equal(o1[1], w1) and np.array_equal(o2[1], w2)
        wq = O.oracle_demod(cfg, quiet)[0]
        assert np.array_equal(np.concatenate([o1[0], o2[0]]), wq) and np.array_equal(np.concatenate([o1[2], o2[2]]), wq)


def test_reset_restores_power_on_state(gpu_device):
    torch = _torch()
    iq = BY_NAME["c1_short"].generate()[:15000]
    x = torch.from_numpy(iq[None]).cuda()
    with Demodulator(C1, 1) as d:
        a = d.process(x).clone()
        d.reset()
        b = d.process(x)
        torch.cuda.synchronize()
        assert torch.equal(a, b) and d.status()[0].n_samples == iq.shape[0]


# ---- scalar primitives on the device, pinned against the reference ------------------------------

def test_device_fast_sin_cos_match_reference(manifest, gpu_device):
    """fast_sin/fast_cos (sincos.c:13-40) over the same 1.3M-point sweep the reference was run on."""
    x = np.concatenate([np.linspace(-9.0, 9.0, 1 << 20, dtype=np.float32),
                        np.random.default_rng(7).uniform(-9, 9, 1 << 18).astype(np.float32),
                        np.array([0.0, -0.0, 6.2831855, -6.2831855, 3.1415927, 1.5707964], dtype=np.float32)])
    meta = manifest["tables"]["sincos"]
    assert sha(x) == meta["x_sha256"]
    with Demodulator(C1, 1) as d:
        s, c = d.selftest_sincos(x)
    assert sha(s) == meta["sin_sha256"]
    assert sha(c) == meta["cos_sha256"]


def test_device_turn_code_shortcut_is_exact_everywhere(gpu_device):
    """The division-free fast_sin turn code equals the real double division for EVERY float
    with |x| < 16 (2.2e9 values), on the device."""
    with Demodulator(C1, 1) as d:
        n, bad = d.selftest_turncode()
    assert n == 2 * 0x41800000 and bad == 0


def test_device_short_cabsf_equals_the_correctly_rounded_one(gpu_device):
    """agc.c:21's cabsf is (float)sqrt((double)re^2 + (double)im^2).  The kernels take a square root good to 2^-46 (v_rsq_f64 + one
    Newton step with an exact residual) and fall back to the correctly rounded one wherever the result lies within 2^-40 of a float
    rounding boundary or the argument is out of the ordinary (demod_device.h: md_cabsf): 2^32 pseudo-random pairs (AGC-like
    magnitudes, the whole float range, zeros) through both on the device, not one different float; the fallback is taken by ~2^-16
    of the AGC-like pairs (2 x 2^12 of 2^29 patterns) plus the out-of-range ones."""
    with Demodulator(C1, 1) as d:
        bad, fallbacks = d.selftest_cabsf(1 << 32)
    assert bad == 0, bad
    assert (1 << 32) // 65536 < fallbacks < (1 << 32) // 8, fallbacks       # near-ties (2^-16 of the pairs) + the pairs drawn out of range on purpose (~5 %)


def test_device_sine_table_equals_the_parabola_for_every_turn_code(gpu_device):
    """The kernel instances of the BASELINE settings read fast_sin's Q14 parabola (sincos.c:26-34) from a table in LDS
    (md_sin_from_code_lut): bit-identical to the integer arithmetic for all 65 536 turn codes, whatever the upper half of the word
    (the parabola's own values are pinned against the reference by test_device_fast_sin_cos_match_reference)."""
    with Demodulator(C1, 1) as d:
        n, bad = d.selftest_sinlut()
    assert n == 4 * 65536 and bad == 0


def test_device_fast_sin_cos_outside_the_shortcut_range(gpu_device):
    """|x| >= 16 takes the real-division fallback (one wave-uniform test in the kernel): mixed waves of in-range and
    out-of-range arguments against the oracle, up to where the reference's int conversion is defined (|x| < 2e5)."""
    rng = np.random.default_rng(11)
    x = np.concatenate([rng.uniform(-9, 9, 4096), rng.uniform(-300, 300, 4096), rng.uniform(-1e5, 1e5, 4096),
                        [15.999999, 16.0, -16.0, 16.000002, 100.0, -1e5]]).astype(np.float32)
    rng.shuffle(x)
    with Demodulator(C1, 1) as d:
        s, c = d.selftest_sincos(x)
    L = O.lib()
    assert np.array_equal(s, np.array([L.orc_fast_sin(float(v)) for v in x], dtype=np.float32))
    assert np.array_equal(c, np.array([L.orc_fast_cos(float(v)) for v in x], dtype=np.float32))


def test_device_cabsf_is_correctly_rounded(gpu_device):
    rng = np.random.default_rng(3)
    xy = np.concatenate([rng.normal(0, 300, (1 << 20, 2)), rng.normal(0, 1e-4, (1 << 16, 2)),
                         rng.normal(0, 1e6, (1 << 16, 2)), np.array([[0, 0], [3, 4], [-0.0, 1e-30]])]).astype(np.float32)
    with Demodulator(C1, 1) as d:
        got = d.selftest_hypot(xy)
    want = np.sqrt(xy[:, 0].astype(np.float64) ** 2 + xy[:, 1].astype(np.float64) ** 2).astype(np.float32)
    assert np.array_equal(got, want)


def test_device_generator_equals_host_generator(gpu_device):
    """The synthetic-input generator is bit-identical on gfx950 and on the host (all formats)."""
    for fmt, rms in ((16, 6000.0), (8, 60.0), (32, 0.5)):
        streams = [synth.make_stream(300 + i, 230000, 72000, f0_hz=-700.0 + 300 * i, clock_ppm=5.0 * i,
                                     oqpsk=bool(i & 1), fmt=fmt, rms=rms) for i in range(5)]
        dev = synth.generate_device(streams, 20000).cpu().numpy()
        for i, st in enumerate(streams):
            assert np.array_equal(dev[i], synth.generate_host(st, 20000)), (fmt, i)


# ---- full-size properties (BASELINE configs[1]/[4] scale) ------------------------------------------

def test_full_size_batch_properties(gpu_device, kernel_variant):
    """65536 streams x 16384 samples (1.07 G samples): replicas of a recording agree byte for
    byte wherever they sit, and randomly sampled streams equal the oracle."""
    torch = _torch()
    ns, n, distinct = 65536, 16384, 256
    streams = [synth.make_stream(100 + i, 230000, 72000, f0_hz=(i % 7 - 3) * 400.0, clock_ppm=(i % 11 - 5) * 8.0)
               for i in range(distinct)]
    base = synth.generate_device(streams, n)
    x = base.repeat(ns // distinct, 1, 1)
    with Demodulator(C1, ns) as d:
        soft = d.process(x)
        torch.cuda.synchronize()
        st = d.status()
        counts = torch.tensor([s.symbols_this_call for s in st], device="cuda")
        # checksum of checksums: every replica group identical
        s3 = soft.view(ns // distinct, distinct, -1)
        c3 = counts.view(ns // distinct, distinct)
        assert torch.equal(c3, c3[:1].expand_as(c3))
        mask = (torch.arange(soft.shape[1], device="cuda")[None, :] < c3[0][:, None]).repeat_interleave(2, dim=1)
        assert torch.equal(s3 * mask, (s3[:1] * mask).expand_as(s3))
        # sampled streams against the oracle
        for i in np.random.default_rng(1).choice(ns, 12, replace=False):
            iq = synth.generate_host(streams[i % distinct], n)
            want = O.oracle_demod(C1, iq)[0]
            assert st[i].symbols_this_call == want.shape[0]
            assert np.array_equal(soft[i, : want.shape[0]].cpu().numpy(), want), i


# ---- the multi-GPU layer on the real backend (RCCL) ------------------------------------------------------

def _rccl_worker(rank, world, port, q):
    import os
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    from meteor_demod_amd.sharding import fanin_soft, shard_range
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    n_streams, n = 70, 6000
    lo, hi = shard_range(n_streams, rank, world)
    streams = [synth.make_stream(40 + i, 230000, 72000, f0_hz=50.0 * i, esn0_db=20.0) for i in range(n_streams)]
    x = synth.generate_device(streams[lo:hi], n, device=rank)
    ok = True
    with Demodulator(C1, hi - lo, device=rank) as d:
        soft = d.process(x)
        torch.cuda.synchronize()
        counts = torch.from_numpy(d.status_array()["symbols_this_call"].astype(np.int32)).to(f"cuda:{rank}")
        pitch = d.nominal_pitch(n)
        assert pitch < soft.shape[1] and int(counts.max()) <= pitch
        packed = d.compact(soft, pitch)                     # nominal pitch: what goes over xGMI
        all_soft, all_cnt = fanin_soft(packed, counts, n_streams, dst=0)
        torch.cuda.synchronize()
        if rank == 0:
            ok = all_soft.shape == (n_streams, pitch, 2) and torch.equal(all_cnt[lo:hi], counts)
            for i in (0, 1, 34, 35, 69):                    # both shards when world == 2
                want = O.oracle_demod(C1, synth.generate_host(streams[i], n))[0]
                ok = ok and int(all_cnt[i]) == want.shape[0] and np.array_equal(all_soft[i, : want.shape[0]].cpu().numpy(), want)
        else:
            ok = all_soft is None
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, bool(ok)))


def _run_rccl(world):
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rccl_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(res.values()), res


@pytest.mark.timeout(300)
def test_rccl_fanin_path_world_size_one(gpu_device):
    """sharding.fanin_soft over the nccl (= RCCL) backend with device tensors compacted to the nominal pitch."""
    _run_rccl(1)


@pytest.mark.timeout(300)
def test_rccl_fanin_path_two_ranks(gpu_device):
    """The same with two ranks on two GPUs (streams sharded, symbols gathered over xGMI); skips itself on a one-GPU box.
    The world_size-2 logic is also covered by the gloo test in test_dist_cpu.py."""
    if _torch().cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    _run_rccl(2)


# ---- random option combinations ------------------------------------------------------------------------

def _random_cfgs(n, seed=2026):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        symrate = int(rng.choice([72000, 80000, 64000, 100000]))
        osf = float(rng.choice([2.0, 2.5, 2.875, 3.19444, 3.5, 3.9, 4.6, 6.0, 9.0]))
        cfg = DemodConfig(samplerate=int(round(symrate * osf)), symrate=symrate,
                          rrc_order=int(rng.choice([8, 16, 17, 24, 32, 33, 40])),
                          interp_factor=int(rng.choice([2, 3, 4, 5, 6, 8])),
                          oqpsk=bool(rng.integers(0, 2)), pll_bw=float(rng.choice([0.5, 1.0, 2.0, 5.0])),
                          freq_max=float(rng.choice([-1.0, 0.05, 0.3, 0.8, 2.0])),
                          bps=int(rng.choice([8, 16, 16, 32])))
        out.append(cfg)
    return out


@pytest.mark.parametrize("cfg", [
    DemodConfig(samplerate=228838, symrate=80000, interp_factor=1, rrc_order=33, oqpsk=True),     # wide geometry, -O 1 (hung once)
    DemodConfig(samplerate=230000, interp_factor=1),                                                # std geometry, -O 1
    DemodConfig(samplerate=230000, interp_factor=1, rrc_order=70),                                  # ring kernel, -O 1
    DemodConfig(samplerate=144000, interp_factor=1, rrc_order=8, oqpsk=True, bps=8),
], ids=["wide-O1-oqpsk", "std-O1", "ring-O1", "std-O1-oqpsk-u8"])
def test_oversampling_factor_one(cfg, gpu_device):
    _check_cfg_against_oracle(cfg)


@pytest.mark.parametrize("cfg", [
    DemodConfig(samplerate=230000, interp_factor=28),                       # largest -O whose std-geometry rows fit in LDS
    DemodConfig(samplerate=230000, interp_factor=29),                       # first one that does not: falls back to the ring kernel
    DemodConfig(samplerate=230000, interp_factor=32, rrc_order=16),
    DemodConfig(samplerate=230000, interp_factor=64),
    DemodConfig(samplerate=230000, symrate=80000, interp_factor=64, oqpsk=True, bps=8),
], ids=["O28", "O29", "O32-f16", "O64", "O64-oqpsk-u8"])
def test_large_oversampling_factors(cfg, gpu_device):
    """-O 29 and above: the per-alignment coefficient rows of the std geometry exceed the 160 KB of LDS; mdemod_create
    used to refuse these valid reference configurations instead of using the v1 kernel.  The v3 kernel takes the
    compact4 table from -O 19 on and keeps them."""
    _check_cfg_against_oracle(cfg)
    from meteor_demod_amd import Demodulator as D
    with D(cfg, 4096) as d:
        assert "v3 rotating register window" in d.kernel_name, d.kernel_name


@pytest.mark.parametrize("cfg", [
    DemodConfig(samplerate=1000000, rrc_order=80, interp_factor=64),                 # 161 taps x 64 banks: 4 alignments of them are 176 KB
    DemodConfig(samplerate=460000, rrc_order=65, interp_factor=64, oqpsk=True, symrate=80000, bps=32),
], ids=["161taps-O64", "131taps-O64-oqpsk-f32"])
def test_tables_that_fit_no_lds_are_read_from_global_memory(cfg, gpu_device):
    """The reference takes any -f / -O; a coefficient table larger than the LDS used to be refused (MDEMOD_ERR_PARAM).  The v1 ring
    kernel now leaves such a table in global memory."""
    torch = _torch()
    rms = {8: 50.0, 16: 5000.0, 32: 0.5}[cfg.bps]
    streams = [synth.make_stream(700 + i, cfg.samplerate, cfg.symrate, f0_hz=150.0 * i, esn0_db=15.0, rms=rms, oqpsk=cfg.oqpsk, fmt=cfg.bps) for i in range(3)]
    iqs = [synth.generate_host(s, 20000) for s in streams]
    with Demodulator(cfg, 3) as d:
        assert "v1 LDS ring" in d.kernel_name, d.kernel_name
        got = [[] for _ in range(3)]
        for lo, hi in ((0, 7001), (7001, 20000)):
            soft = d.process(torch.from_numpy(np.stack([a[lo:hi] for a in iqs])).cuda())
            torch.cuda.synchronize()
            cnt = d.symbol_counts()
            for i in range(3):
                got[i].append(soft[i, : int(cnt[i])].cpu().numpy())
        for i in range(3):
            assert np.array_equal(np.concatenate(got[i]), O.oracle_demod(cfg, iqs[i])[0]), i


def _check_cfg_against_oracle(cfg):
    """-O 1: floor(x / 1) cannot go through the 32-bit reciprocal the symbol clock uses for x / interp (the reciprocal
    of 1 is 2^32); found by tools/config_fuzz.py as an endless loop in the kernel."""
    torch = _torch()
    rms = {8: 50.0, 16: 5000.0}[cfg.bps]
    streams = [synth.make_stream(300 + i, cfg.samplerate, cfg.symrate, f0_hz=200.0 * i, esn0_db=15.0, rms=rms, oqpsk=cfg.oqpsk, fmt=cfg.bps)
               for i in range(5)]
    iqs = [synth.generate_host(s, 9000) for s in streams]
    with Demodulator(cfg, 5) as d:
        soft = d.process(torch.from_numpy(np.stack(iqs)).cuda())
        torch.cuda.synchronize()
        cnt = d.symbol_counts()
        for i in range(5):
            assert np.array_equal(soft[i, : int(cnt[i])].cpu().numpy(), O.oracle_demod(cfg, iqs[i])[0]), i
        assert all(s.overflow == 0 for s in d.status())


@pytest.mark.parametrize("kernel", ["v1", ""], ids=["ring", "hybrid"])
def test_float_input_ring_kernel_ignores_stale_lds(kernel, gpu_device, monkeypatch):
    """Float input with window slots outside a lane's taps (the v1 ring kernel's LDS ring; the v3 hybrid window's registers): they are
    multiplied by zero coefficients, so they must never hold stale NaN bits (0 * NaN = NaN).  A first context fills the CUs' LDS
    and registers with NaN samples."""
    torch = _torch()
    monkeypatch.setenv("MDEMOD_KERNEL", kernel)
    cfg = DemodConfig(samplerate=1072367, pll_bw=5.0, symrate=72000, interp_factor=4, rrc_order=33, oqpsk=True, bps=32)
    with Demodulator(cfg, 8192) as poison:
        poison.process(torch.full((8192, 600, 2), float("nan"), dtype=torch.float32, device="cuda"))
        torch.cuda.synchronize()
    streams = [synth.make_stream(40 + i, cfg.samplerate, cfg.symrate, f0_hz=300.0, esn0_db=15.0, rms=0.7, oqpsk=True, fmt=32) for i in range(6)]
    iqs = [synth.generate_host(s, 9051) for s in streams]
    with Demodulator(cfg, 29) as d:
        assert ("ring" if kernel == "v1" else "hybrid") in d.kernel_name, d.kernel_name
        got = [[] for _ in range(29)]
        for lo, hi in ((0, 1000), (1000, 9051)):
            soft = d.process(torch.from_numpy(np.stack([iqs[i % 6][lo:hi] for i in range(29)])).cuda())
            torch.cuda.synchronize()
            cnt = d.symbol_counts()
            for i in range(29):
                got[i].append(soft[i, : int(cnt[i])].cpu().numpy())
        for i in range(29):
            assert np.array_equal(np.concatenate(got[i]), O.oracle_demod(cfg, iqs[i % 6])[0]), i


def test_few_streams_take_the_kernel_that_is_faster_for_them(gpu_device, monkeypatch):
    """wants_latency_kernel (demod_api.cpp): a context with few streams runs one stream per WAVE up to 32 samples per firing and one per
    LANE of a v3 kernel above that (where a batch of the wave kernel's FIR farm holds too few firings: tools/one_stream_rates.py); both
    byte-identical to the oracle, as every variant is."""
    torch = _torch()
    monkeypatch.delenv("MDEMOD_LAT", raising=False)
    monkeypatch.delenv("MDEMOD_KERNEL", raising=False)
    for rate, oq, want in ((230000, False, "demod_kernel_lat"), (1800000, False, "demod_kernel_lat"), (3200000, False, "demod_kernel_rotp"),
                           (2400000, True, "demod_kernel_lat"), (6000000, True, "demod_kernel_rotp"), (10000000, False, "demod_kernel_gat")):
        cfg = DemodConfig(samplerate=rate, symrate=80000 if oq else 72000, oqpsk=oq)
        st = synth.make_stream(17, cfg.samplerate, cfg.symrate, f0_hz=-500.0, esn0_db=15.0, oqpsk=oq)
        iq = synth.generate_host(st, 9000 * max(1, rate // 1000000))
        with Demodulator(cfg, 2) as d:
            assert want in d.kernel_name, (rate, oq, d.kernel_name)
            soft = d.process(torch.from_numpy(np.stack([iq, iq])).cuda())
            torch.cuda.synchronize()
            w = O.oracle_demod(cfg, iq)[0]
            assert np.array_equal(soft[1, : d.status(1, 1)[0].symbols_this_call].cpu().numpy(), w), (rate, oq)


@pytest.mark.timeout(120)
@pytest.mark.parametrize("rate,streams", [(3200000, 300), (3200000, 1), (6000000, 300), (1800000, 1)], ids=["hybrid-far", "one-stream", "gather", "lat-1800k"])
def test_nan_and_inf_samples_end_the_launch_at_rates_with_a_clock_schedule(rate, streams, gpu_device):
    """Float input gone bad (NaN / Inf samples) at sample rates whose symbol clock runs on the closed-form schedule (clock_jump.h): the
    launch ends, the samples are consumed, and a reset context demodulates clean input byte for byte afterwards.  What the loops hold
    meanwhile is outside anything the reference defines (its tanh look-up reads out of bounds one symbol after a non-finite sample,
    pll.c:154-159): here the AGC, the phases and the soft values turn NaN, while the two clamped words - the carrier word and the clock
    word's deviation - come out of md_clamp_sym (v_med3_f32) as -fmax / -maxdev rather than NaN, and a NaN soft value is emitted as the
    byte 0x81 (-127; x86's (int8)NaN is 0).  The schedule's stepping-up loop is bounded by a host count and `p <= lo` is false for a NaN."""
    torch = _torch()
    cfg = DemodConfig(samplerate=rate, bps=32)
    st = synth.make_stream(91, cfg.samplerate, cfg.symrate, f0_hz=400.0, esn0_db=15.0, rms=0.3, fmt=32)
    n = 12000 * max(1, rate // 3200000)
    good = synth.generate_host(st, n)
    bad = good.copy()
    bad[n // 3:, 0] = np.nan
    bad[n // 2:, 1] = np.inf
    with Demodulator(cfg, streams) as d:
        d.process(torch.from_numpy(np.stack([bad] * streams)).cuda())
        torch.cuda.synchronize()
        assert all(s.n_samples == n for s in d.status())
        # pinned: the clock word stays a finite word of the loop's range, so no later launch can meet a word the schedules do not cover
        tf = np.array([s.omega for s in d.status()], dtype=np.float64)
        assert np.all(np.isfinite(tf)), tf[:4]
        d.reset()
        soft = d.process(torch.from_numpy(np.stack([good] * streams)).cuda())
        torch.cuda.synchronize()
        want = O.oracle_demod(cfg, good)[0]
        for i in (0, streams - 1):
            assert np.array_equal(soft[i, : d.status(i, 1)[0].symbols_this_call].cpu().numpy(), want), i


@pytest.mark.parametrize("cfg", [
    DemodConfig(samplerate=324459, symrate=36000, interp_factor=2, rrc_order=33, pll_bw=100.0, freq_max=1.5),   # wide window
    DemodConfig(samplerate=230000, pll_bw=100.0, freq_max=1.5),                                                 # std window
    DemodConfig(samplerate=230000, symrate=80000, oqpsk=True, pll_bw=100.0),
    DemodConfig(samplerate=230000),
], ids=["wide", "std", "std-oqpsk", "defaults"])
def test_full_scale_input_only_last_symbol_of_a_sample_is_kept(cfg, gpu_device, kernel_variant):
    """A full-scale burst after silence (AGC gain high): the timing error term (alpha * e, e ~ 1e5) pulls the symbol clock
    back over its threshold at once and several symbols fire inside one input sample; the reference keeps only the last
    one (demod.c:33-47).  Happens with the default options too.  Found by tools/config_fuzz.py with pathological inputs."""
    torch = _torch()
    iqs = []
    for sd in range(4):                    # silence lets the AGC gain climb, then a full-scale burst arrives
        rng = np.random.default_rng(sd)
        iqs.append(np.concatenate([np.zeros((3000, 2), np.int16), rng.choice(np.array([-32767, 32767], dtype=np.int16), size=(6000, 2))]))
    iqs.append(np.random.default_rng(9).choice(np.array([-32767, 32767], dtype=np.int16), size=(9000, 2)))
    iqs.append(np.zeros((9000, 2), np.int16))                      # silence only: the gain ramps up, nothing else happens
    with Demodulator(cfg, len(iqs)) as d:
        soft = d.process(torch.from_numpy(np.stack(iqs)).cuda())
        torch.cuda.synchronize()
        cnt = d.symbol_counts()
        st = d.status()
        for i, a in enumerate(iqs):
            ost = O.OracleStream(cfg)
            want = ost.run(a)[0]
            assert int(cnt[i]) == len(want) and np.array_equal(soft[i, : len(want)].cpu().numpy(), want), i
            assert np.float32(st[i].gain) == np.float32(ost.state.gain) and st[i].n_symbols == len(want)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("seed", [21, 22])
def test_option_fuzz(seed, gpu_device):
    """tools/config_fuzz.py: 150 random option combinations (all three kernel geometries, every input format, ragged and
    empty chained blocks, up to 69 streams) against the oracle, loop state included."""
    import subprocess, sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "config_fuzz.py"), "150", str(seed)], capture_output=True, text=True,
                       cwd=str(ROOT), timeout=500)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "failures: 0" in r.stdout, r.stdout[-3000:]


@pytest.mark.timeout(600)
def test_entry_point_fuzz(gpu_device):
    """tools/api_fuzz.py: ragged launches with unaligned offsets, state/history hand-off between contexts and the pipelined
    host path, 120 random option combinations."""
    import subprocess, sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "api_fuzz.py"), "120", "5"], capture_output=True, text=True,
                       cwd=str(ROOT), timeout=500)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "failures: 0" in r.stdout, r.stdout[-3000:]


@pytest.mark.parametrize("idx", range(14))
def test_random_option_combinations_match_oracle(idx, gpu_device):
    """-f/-O/-r/-s/-b/-d/-m/--bps drawn at random (both kernels get selected: > 65 taps or > 3.6 samples per
    firing go to the ring kernel): 3 streams x 2 chained blocks each, byte-identical to the oracle."""
    torch = _torch()
    cfg = _random_cfgs(14)[idx]
    rms = {8: 50.0, 16: 5000.0, 32: 0.7}[cfg.bps]
    n1, n2 = 9000, 5003
    streams = [synth.make_stream(900 + 10 * idx + i, cfg.samplerate, cfg.symrate, f0_hz=(i - 1) * 500.0, esn0_db=16.0,
                                 rms=rms, dc=(rms / 200, -rms / 300), oqpsk=cfg.oqpsk, fmt=cfg.bps) for i in range(3)]
    iqs = [synth.generate_host(s, n1 + n2) for s in streams]
    with Demodulator(cfg, 3) as d:
        got = [[], [], []]
        for lo, hi in ((0, n1), (n1, n1 + n2)):
            soft = d.process(torch.from_numpy(np.stack([a[lo:hi] for a in iqs])).cuda())
            torch.cuda.synchronize()
            st = d.status()
            for i in range(3):
                got[i].append(soft[i, : st[i].symbols_this_call].cpu().numpy())
        for i in range(3):
            ost = O.OracleStream(cfg)
            want = ost.run(iqs[i])[0]
            assert np.array_equal(np.concatenate(got[i]), want), (idx, i, cfg)
            assert np.float32(st[i].pll_freq) == np.float32(ost.state.pll_freq) and st[i].locked == ost.state.locked


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref (the reference built from /root/reference) did not travel with this checkout")
@pytest.mark.parametrize("seed", [21, 22])
def test_hip_equals_the_live_reference(seed, gpu_device, kernel_variant):
    """Not through the restatement: the HIP path against the REFERENCE ITSELF (oracle/_ref/ref_harness = the reference's own objects,
    built in the survey container and carried along) on random settings - every kernel variant, byte for byte."""
    torch = _torch()
    rng = np.random.default_rng(seed)
    done = 0
    while done < 8:
        oqpsk = bool(rng.random() < 0.4)
        symrate = int(rng.choice([72000, 80000]))
        cfg = DemodConfig(samplerate=int(symrate * float(rng.uniform(1.4, 16.0))), symrate=symrate, oqpsk=oqpsk, interp_factor=int(rng.integers(1, 9)),
                          rrc_order=int(rng.integers(8, 66)), pll_bw=float(rng.choice([0.5, 1.0, 2.0])), bps=int(rng.choice([8, 16, 32])))
        if not np.isfinite(O.OracleStream(cfg).rrc_table()).all():
            continue
        st = synth.make_stream(int(rng.integers(1, 1 << 30)), cfg.samplerate, cfg.symrate, oqpsk=oqpsk, f0_hz=float(rng.uniform(-900, 900)), esn0_db=14.0,
                               fmt=cfg.bps, **({"rms": 40.0} if cfg.bps == 8 else {"rms": 0.3} if cfg.bps == 32 else {}))
        iq = synth.generate_host(st, int(2500 * cfg.samplerate / cfg.symrate))
        want, _ = O.ref_demod(cfg, iq)
        with Demodulator(cfg, 3) as d:
            soft = d.process(torch.from_numpy(np.stack([iq, iq, iq])).cuda())
            torch.cuda.synchronize()
            m = d.status(2, 1)[0].symbols_this_call
            assert m == want.shape[0] and np.array_equal(soft[2, :m].cpu().numpy(), want), (cfg, d.kernel_name)
        done += 1
