"""GPU (-m gpu): one recording as overlapped tiles (meteor_demod_amd/recording.py) on the HIP path.

Every piece of the scheme is an ordinary bit-exact stream, so the stitched output of the HIP bank
must equal, byte for byte, the stitched output of the same scheme driven by the oracle
(tests/oracle_bank.py); agreement with the UNTILED serial reference is statistical."""
from __future__ import annotations

import numpy as np
import pytest

import oracle_py as O
from oracle_bank import OracleBank
from meteor_demod_amd import DemodConfig, synth
from meteor_demod_amd.recording import RecordingDemodulator, agreement

pytestmark = pytest.mark.gpu

C1 = DemodConfig(samplerate=230000)


@pytest.mark.parametrize("refine", [True, False])
def test_hip_stitched_recording_equals_oracle_stitched_recording(refine, gpu_device):
    import torch
    st = synth.make_stream(4242, 230000, 72000, f0_hz=300.0, clock_ppm=11.0, esn0_db=12.0)
    iq = synth.generate_host(st, 1_500_000)
    kw = dict(tile_samples=32768, pre_samples=8192, refine=refine, pilot_block=65536, pilot_margin_symbols=60000)
    want = RecordingDemodulator(C1, bank_factory=lambda c, k: OracleBank(c, k), **kw).demodulate(torch.from_numpy(iq))
    got = RecordingDemodulator(C1, **kw).demodulate(torch.from_numpy(iq).cuda())
    assert got.report.n_tiles == want.report.n_tiles > 20
    assert got.report.rotations == want.report.rotations and got.report.seam_shifts == want.report.seam_shifts
    assert got.report.first_lock_symbol == want.report.first_lock_symbol == 12775
    assert np.array_equal(got.soft.cpu().numpy(), want.soft.numpy())
    assert np.array_equal(got.tile_first_symbol, want.tile_first_symbol)


def test_hip_stitched_oqpsk_recording_equals_oracle_stitched_recording(gpu_device):
    """OQPSK: per-rail rotation matching, state rotation with the half-symbol clock move (mdemod_rotate_carrier in OQPSK
    mode), look-ahead seams: the HIP bank reproduces the oracle bank byte for byte and agrees with the serial run."""
    import torch
    cfg = DemodConfig(samplerate=230000, symrate=80000, oqpsk=True)
    st = synth.make_stream(4242, 230000, 80000, f0_hz=300.0, clock_ppm=11.0, esn0_db=14.0, oqpsk=True)
    iq = synth.generate_host(st, 2_000_000)
    kw = dict(tile_samples=32768, pre_samples=8192, pilot_block=65536, pilot_margin_symbols=80000)
    want = RecordingDemodulator(cfg, bank_factory=lambda c, k: OracleBank(c, k), **kw).demodulate(torch.from_numpy(iq))
    got = RecordingDemodulator(cfg, **kw).demodulate(torch.from_numpy(iq).cuda())
    assert got.report.n_tiles == want.report.n_tiles > 30 and got.report.rotations == want.report.rotations
    assert len(set(got.report.rotations)) == 4 and got.report.seam_shifts == want.report.seam_shifts
    assert np.array_equal(got.soft.cpu().numpy(), want.soft.numpy())
    a = agreement(got.soft.cpu().numpy(), O.oracle_demod(cfg, iq)[0])
    assert a["len_stitched"] == a["len_serial"] and a["hard_decisions_equal"] > 0.9999


def test_long_recording_on_many_lanes_agrees_with_the_serial_reference(gpu_device):
    """16 M samples (70 s of signal) as 237 tiles of 65536: pilot bytes are the reference's, tiles within the
    loops' noise of the serial run, symbol count preserved."""
    import torch
    st = synth.make_stream(99, 230000, 72000, f0_hz=-700.0, clock_ppm=-20.0, esn0_db=12.0)
    iq = synth.generate_device([st], 16_000_000)[0]
    serial = O.oracle_demod(C1, iq.cpu().numpy())[0]
    res = RecordingDemodulator(C1).demodulate(iq)
    out = res.soft.cpu().numpy()
    r = res.report
    assert r.pilot_locked and r.n_tiles > 200 and r.weak_seams == 0
    assert np.array_equal(out[: r.pilot_symbols], serial[: r.pilot_symbols])
    a = agreement(out, serial)
    assert a["len_stitched"] == a["len_serial"]
    assert a["hard_decisions_equal"] > 0.9999 and a["within_1lsb"] > 0.93
    assert all(x == 0 for x in r.refine_rotations)


def test_state_broadcast_and_carrier_rotation_primitives(gpu_device):
    """mdemod_set_state_all / mdemod_rotate_carrier against their definitions."""
    import math
    import torch
    from meteor_demod_amd import Demodulator
    st = synth.make_stream(5, 230000, 72000, f0_hz=100.0)
    iq = synth.generate_host(st, 20000)
    with Demodulator(C1, 1) as one, Demodulator(C1, 6) as bank:
        one.process(torch.from_numpy(iq[None, :12000]).cuda())
        seed = one.get_state(0)
        bank.process(torch.from_numpy(np.stack([iq[:3000]] * 6)).cuda())          # dirty the bank first
        bank.set_state_all(seed)
        for s in range(6):
            g = bank.get_state(s)
            for f, _ in seed._fields_:
                assert getattr(g, f) == getattr(seed, f), (s, f)
            assert not bank.get_history(s).any()
        q = torch.tensor([0, 1, 2, 3, 5, -1], dtype=torch.int32, device="cuda")
        bank.rotate_carrier(q)
        for s, k in enumerate([0, 1, 2, 3, 1, 3]):
            want = np.float32(math.fmod(float(seed.pll_phase) + k * (math.pi / 2), 2 * math.pi)) if k else np.float32(seed.pll_phase)
            assert np.float32(bank.get_state(s).pll_phase) == want, (s, k)
        # a stream seeded this way + the history of the donor continues the donor's stream exactly
        bank.set_state_all(seed)
        bank.set_history(2, one.get_history(0))
        soft = bank.process(torch.from_numpy(np.stack([iq[12000:]] * 6)).cuda())
        m = int(bank.symbol_counts()[2])
        want_all = O.oracle_demod(C1, iq)[0]
        tail = soft[2, :m].cpu().numpy()
        assert np.array_equal(tail, want_all[len(want_all) - m:])


@pytest.mark.parametrize("refine", [True, False])
def test_native_stitcher_equals_python_stitcher(refine, gpu_device):
    """mdemod_demodulate_recording (csrc/recording.hip) makes the same decisions as recording.py: same bytes."""
    import torch
    from meteor_demod_amd.recording import demodulate_recording_native
    st = synth.make_stream(77, 230000, 72000, f0_hz=450.0, clock_ppm=-30.0, esn0_db=10.0)
    iq = synth.generate_device([st], 6_000_000)[0]
    kw = dict(tile_samples=32768, pre_samples=8192, refine=refine, pilot_block=65536, pilot_margin_symbols=80000)
    want = RecordingDemodulator(C1, **kw).demodulate(iq)
    soft, rep = demodulate_recording_native(C1, iq, **kw)
    assert rep.n_tiles == want.report.n_tiles > 100 and rep.pilot_symbols == want.report.pilot_symbols
    assert rep.first_lock_symbol == want.report.first_lock_symbol and rep.weak_seams == want.report.weak_seams
    assert rep.seam_fixes == sum(1 for s in want.report.seam_shifts if s)
    assert rep.samples_demodulated == want.report.samples_demodulated
    assert np.array_equal(soft.cpu().numpy(), want.soft.cpu().numpy())


def test_native_stitcher_equals_python_stitcher_oqpsk(gpu_device):
    import torch
    from meteor_demod_amd.recording import demodulate_recording_native
    cfg = DemodConfig(samplerate=230000, symrate=80000, oqpsk=True)
    st = synth.make_stream(78, 230000, 80000, f0_hz=-350.0, clock_ppm=25.0, esn0_db=12.0, oqpsk=True)
    iq = synth.generate_device([st], 5_000_000)[0]
    kw = dict(tile_samples=32768, pre_samples=8192, refine=True, pilot_block=65536, pilot_margin_symbols=80000)
    want = RecordingDemodulator(cfg, **kw).demodulate(iq)
    soft, rep = demodulate_recording_native(cfg, iq, **kw)
    assert rep.n_tiles == want.report.n_tiles > 50 and rep.weak_seams == want.report.weak_seams
    assert rep.seam_fixes == sum(1 for s in want.report.seam_shifts if s)
    assert rep.samples_demodulated == want.report.samples_demodulated
    assert np.array_equal(soft.cpu().numpy(), want.soft.cpu().numpy())


def test_native_stitcher_short_recording_and_errors(gpu_device):
    import torch
    from meteor_demod_amd._capi import MdemodError
    from meteor_demod_amd.recording import demodulate_recording_native
    st = synth.make_stream(3, 230000, 72000, f0_hz=0.0)
    iq = synth.generate_device([st], 150_000)[0]
    soft, rep = demodulate_recording_native(C1, iq, pilot_margin_symbols=160000)      # the pilot never gets that far: all serial
    assert rep.n_tiles == 0 and np.array_equal(soft.cpu().numpy(), O.oracle_demod(C1, iq.cpu().numpy())[0])
    with pytest.raises(MdemodError):      # OQPSK needs the state rotation pass
        demodulate_recording_native(DemodConfig(samplerate=230000, symrate=80000, oqpsk=True), iq, refine=False)


@pytest.mark.parametrize("seed", range(4))
def test_native_and_python_stitchers_agree_on_random_settings_oqpsk(seed, gpu_device):
    """The same for OQPSK (rail matching, state rotation, look-ahead seams), with gaps and bursts."""
    import torch
    from meteor_demod_amd.recording import demodulate_recording_native
    rng = np.random.default_rng(300 + seed)
    cfg = DemodConfig(samplerate=int(rng.choice([230000, 250000, 460000])), symrate=80000, oqpsk=True)
    st = synth.make_stream(700 + seed, cfg.samplerate, 80000, f0_hz=float(rng.uniform(-500, 700)), clock_ppm=float(rng.uniform(-40, 40)),
                           esn0_db=float(rng.choice([6.0, 10.0, 14.0, 20.0])), oqpsk=True, rms=3000.0)
    n = int(rng.integers(2_000_000, 4_000_000))
    iq = synth.generate_device([st], n)[0]
    if seed >= 2:
        g0 = int(rng.integers(n // 3, n // 2))
        iq[g0: g0 + 200_000] = 0
        iq[g0 + 200_000: g0 + 215_000] = -32768
    kw = dict(tile_samples=int(rng.choice([0, 16448, 40000])), pre_samples=int(rng.choice([-1, 4096, 20000])), refine=True,
              pilot_block=int(rng.choice([16384, 65536])), pilot_margin_symbols=int(rng.choice([0, 5000, 60000])),
              match_symbols=int(rng.choice([64, 192])))
    want = RecordingDemodulator(cfg, **kw).demodulate(iq)
    soft, rep = demodulate_recording_native(cfg, iq, **kw)
    assert rep.n_tiles == want.report.n_tiles and rep.pilot_symbols == want.report.pilot_symbols, kw
    assert rep.weak_seams == want.report.weak_seams and rep.seam_fixes == sum(1 for s in want.report.seam_shifts if s), kw
    assert np.array_equal(soft.cpu().numpy(), want.soft.cpu().numpy()), kw


@pytest.mark.parametrize("seed", range(6))
def test_native_and_python_stitchers_agree_on_random_settings(seed, gpu_device):
    """Random tile / warm-up / margin / match settings, offsets of both signs, noise levels down to 4 dB (weak seams,
    unlocked pilots): the two implementations must make identical decisions."""
    import torch
    from meteor_demod_amd.recording import demodulate_recording_native
    rng = np.random.default_rng(100 + seed)
    st = synth.make_stream(500 + seed, 230000, 72000, f0_hz=float(rng.uniform(-600, 900)), clock_ppm=float(rng.uniform(-40, 40)),
                           esn0_db=float(rng.choice([4.0, 8.0, 12.0, 20.0])))
    n = int(rng.integers(2_000_000, 5_000_000))
    iq = synth.generate_device([st], n)[0]
    if seed >= 4:                          # a silent gap and a full-scale burst in the middle: unlocked tiles, weak seams
        g0 = int(rng.integers(n // 3, n // 2))
        iq[g0: g0 + 300_000] = 0
        iq[g0 + 300_000: g0 + 320_000] = 32767
    kw = dict(tile_samples=int(rng.choice([8200, 16448, 40000, 65600])), pre_samples=int(rng.choice([0, 2048, 8192, 20000])),
              refine=bool(rng.random() < 0.6), pilot_block=int(rng.choice([16384, 65536])),
              pilot_margin_symbols=int(rng.choice([0, 5000, 60000])), max_pilot_samples=int(rng.choice([300_000, 1 << 22])),
              match_symbols=int(rng.choice([32, 192, 400])))
    want = RecordingDemodulator(C1, **kw).demodulate(iq)
    soft, rep = demodulate_recording_native(C1, iq, **kw)
    assert rep.n_tiles == want.report.n_tiles and rep.pilot_symbols == want.report.pilot_symbols, kw
    assert rep.weak_seams == want.report.weak_seams and rep.seam_fixes == sum(1 for s in want.report.seam_shifts if s), kw
    assert np.array_equal(soft.cpu().numpy(), want.soft.cpu().numpy()), kw


def test_recording_at_one_megasample_uses_scaled_tiles(gpu_device):
    """A 1.024 MS/s recording (default filter: mid geometry): the default tile and warm-up lengths scale with the samples
    per symbol (292 k / 73 k samples), native == python, and the result agrees with the serial oracle."""
    import torch
    from meteor_demod_amd.recording import default_tiling, demodulate_recording_native
    cfg = DemodConfig(samplerate=1024000)
    assert default_tiling(C1) == (65600, 16384) and default_tiling(cfg)[0] > 290_000
    # rms 1500: with 14 samples per symbol a 6000-LSB signal drives the reference's AGC into its 0 <-> 0.019 limit cycle
    # (gain += 1e-4 * (190 - |y|) overshoots below zero), in the serial run as much as in the tiles
    st = synth.make_stream(12, 1024000, 72000, f0_hz=500.0, clock_ppm=-15.0, esn0_db=14.0, rms=1500.0)
    iq = synth.generate_device([st], 24_000_000)[0]
    want = RecordingDemodulator(cfg).demodulate(iq)
    soft, rep = demodulate_recording_native(cfg, iq)
    assert rep.n_tiles == want.report.n_tiles > 50 and rep.weak_seams == 0
    assert np.array_equal(soft.cpu().numpy(), want.soft.cpu().numpy())
    serial = O.oracle_demod(cfg, iq.cpu().numpy())[0]
    a = agreement(soft.cpu().numpy(), serial)
    assert a["len_stitched"] == a["len_serial"] and a["hard_decisions_equal"] > 0.9999 and a["within_1lsb"] > 0.95
    assert np.array_equal(soft[: rep.pilot_symbols].cpu().numpy(), serial[: rep.pilot_symbols])


def test_doppler_recording_with_spectral_carrier_seeds(gpu_device):
    """40 Hz/s of Doppler over 52 s on the GPU (mdemod_set_carrier_seeds + torch.fft estimates): the stitched stream
    keeps the serial run's symbol count and decisions; the per-stream seed op is checked against its definition."""
    import torch
    from meteor_demod_amd import Demodulator
    st = synth.make_stream(78, 230000, 72000, f0_hz=-300.0, clock_ppm=-8.0, esn0_db=12.0, doppler_hz_per_s=40.0)
    iq = synth.generate_device([st], 12_000_000)[0]
    serial = O.oracle_demod(C1, iq.cpu().numpy())[0]
    res = RecordingDemodulator(C1, carrier_seed="spectrum").demodulate(iq)
    a = agreement(res.soft.cpu().numpy(), serial)
    assert res.report.weak_seams == 0 and a["len_stitched"] == a["len_serial"]
    assert a["hard_decisions_equal"] > 0.9999 and a["within_1lsb"] > 0.97
    with Demodulator(C1, 5) as d:
        f = torch.tensor([0.1, -0.2, 0.0, 0.05, 0.3], dtype=torch.float32, device="cuda")
        u = torch.tensor([1, -1, 1, -1, 1], dtype=torch.int32, device="cuda")
        d.set_carrier_seeds(f, u)
        for i in range(5):
            g = d.get_state(i)
            assert np.float32(g.pll_freq) == np.float32(f[i].item()) and g.pll_updown == int(u[i]) and g.agc_gain == 1.0


@pytest.mark.gpu
@pytest.mark.parametrize("bps,oqpsk", [(16, False), (8, False), (16, True)])
def test_native_stitcher_follows_doppler(gpu_device, bps, oqpsk):
    """mdemod_demodulate_recording with carrier_seed=spectrum (carrier_line_kernel in csrc/recording.hip): same bar as the
    Python stitcher on a 40 Hz/s ramp; with pilot seeds the same recording loses tiles (that is what the option is for)."""
    import dataclasses
    from meteor_demod_amd.recording import demodulate_recording_native
    amp = dict(rms=6000.0) if bps == 16 else dict(rms=40.0, dc=(1.5, -1.0))
    symrate = 80000 if oqpsk else 72000
    # OQPSK: start above the carrier the sweep meets first.  With f0 = -300 Hz the REFERENCE declares lock at +575 Hz while
    # sweeping up (false lock, 28 k symbols in) and only reaches the carrier 160 k symbols later; tiles seeded from the
    # spectrum are on the carrier at once, so there is no serial stream to compare them with.
    f0, ramp = (500.0, -40.0) if oqpsk else (-300.0, 40.0)
    st = synth.make_stream(80 if oqpsk else 79, 230000, symrate, f0_hz=f0, clock_ppm=5.0, esn0_db=13.0 if oqpsk else 12.0,
                           doppler_hz_per_s=ramp, fmt=bps, oqpsk=oqpsk, **amp)
    iq = synth.generate_device([st], 12_000_000)[0]
    cfg = dataclasses.replace(C1, bps=bps, symrate=symrate, oqpsk=oqpsk)
    serial = O.oracle_demod(cfg, iq.cpu().numpy())[0]
    soft, rep = demodulate_recording_native(cfg, iq, carrier_seed="spectrum")
    a = agreement(soft.cpu().numpy(), serial)
    assert rep.weak_seams == 0 and a["len_stitched"] == a["len_serial"]
    assert a["hard_decisions_equal"] > 0.9999 and a["within_1lsb"] > 0.97
    soft0, rep0 = demodulate_recording_native(cfg, iq, carrier_seed="pilot")
    a0 = agreement(soft0.cpu().numpy(), serial)
    assert a0["hard_decisions_equal"] < a["hard_decisions_equal"] or rep0.weak_seams > 0


@pytest.mark.gpu
def test_native_stitcher_bridges_a_fade_with_neighbouring_carrier_estimates(gpu_device):
    """Three tiles' worth of the recording replaced by noise while the carrier ramps 40 Hz/s: those tiles have no spectral
    line and take their neighbours' carrier (mdemod_recording_report.weak_carrier_tiles); before the fade the output is
    the serial run's, after it the two agree again up to the lock's own quarter-turn ambiguity (the serial run re-locks on
    its own after the fade, so rotation and the symbol count across the fade are not comparable)."""
    import torch
    from meteor_demod_amd.recording import demodulate_recording_native, rotate_symbols
    st = synth.make_stream(91, 230000, 72000, f0_hz=300.0, clock_ppm=5.0, esn0_db=12.0, doppler_hz_per_s=40.0)
    iq = synth.generate_device([st], 8_000_000)[0]
    a, b = 3_000_000, 3_000_000 + 3 * 65600
    g = torch.Generator(device="cuda").manual_seed(3)
    iq[a:b] = (torch.randn((b - a, 2), device="cuda", generator=g) * 4200).to(torch.int16)
    serial, tr, ev = O.oracle_demod(C1, iq.cpu().numpy(), True)
    soft, rep = demodulate_recording_native(C1, iq, carrier_seed="spectrum")
    out = soft.cpu().numpy()
    assert 2 <= rep.weak_carrier_tiles <= 5
    n_before = int(a * 72000 / 230000) - 2000
    assert ((out[:n_before] >= 0) == (serial[:n_before] >= 0)).all(axis=1).mean() > 0.9999
    L = 1_000_000                                                   # the last 3.2 s, long after both have re-locked
    best = 0.0
    for r in range(4):
        rot = rotate_symbols(torch.from_numpy(out[-L - 4:]).unsqueeze(0), torch.tensor([r]))[0].numpy()
        for s in range(-2, 3):
            seg = rot[4 + s: 4 + s + L - 8]
            best = max(best, ((seg >= 0) == (serial[-L: -8] >= 0)).all(axis=1).mean())
    assert best > 0.9999
    assert tr["locked"][-L:].all()


@pytest.mark.gpu
def test_native_stitcher_seeds_the_agc_on_float_input(gpu_device):
    """Float input around +-1 with the amplitude swinging 0.4..1.0 over 12 s: mdemod_demodulate_recording waits for the
    reference's AGC to settle before handing over, then gives every tile the gain of the closed-form AGC recursion
    (mdemod_set_gain_seeds).  Without the seeds (carrier_seed=pilot) under a third of the soft symbols are within an LSB."""
    import dataclasses
    import torch
    from meteor_demod_amd import Demodulator
    from meteor_demod_amd.recording import demodulate_recording_native
    cfg = dataclasses.replace(C1, bps=32)
    st = synth.make_stream(7, 230000, 72000, f0_hz=300.0, clock_ppm=5.0, esn0_db=12.0, fmt=32, rms=0.25, dc=(0.001, -0.002))
    n = 6_000_000
    iq = synth.generate_device([st], n)[0]
    t = torch.arange(n, device="cuda", dtype=torch.float32) / 230000
    iq *= (0.7 - 0.3 * torch.cos(2 * np.pi * t / 12.0))[:, None]
    serial = O.oracle_demod(cfg, iq.cpu().numpy())[0]
    soft, rep = demodulate_recording_native(cfg, iq, carrier_seed="spectrum")
    a = agreement(soft.cpu().numpy(), serial)
    assert rep.pilot_symbols > 150_000 and rep.weak_seams == 0
    assert a["len_stitched"] == a["len_serial"] and a["hard_decisions_equal"] > 0.99999 and a["within_1lsb"] > 0.98
    soft0, _ = demodulate_recording_native(cfg, iq, carrier_seed="pilot")
    assert agreement(soft0.cpu().numpy(), serial)["within_1lsb"] < 0.6
    with Demodulator(C1, 3) as d:                                        # the seed op against its definition
        d.set_gain_seeds(torch.tensor([0.5, -2.0, 700.0], dtype=torch.float32, device="cuda"))
        assert [d.get_state(i).agc_gain for i in range(3)] == [0.5, 0.0, 700.0]


@pytest.mark.gpu
def test_native_stitcher_opens_the_lock_gate_from_the_tiles_when_the_pilot_never_locks(gpu_device):
    """A recording that starts before the signal does: the serial head gives up after max_pilot_samples without a lock, so
    the lock gate (main.c:308-315) has to come from the tiles - the first one whose stream reports a first lock.  The
    symbols after that point are the transmitted ones (checked against the serial run where that one is locked too)."""
    import torch
    from meteor_demod_amd.recording import demodulate_recording_native, rotate_symbols
    st = synth.make_stream(93, 230000, 72000, f0_hz=500.0, clock_ppm=5.0, esn0_db=12.0, rms=1500.0)
    n_noise, n_sig = 1_500_000, 5_000_000
    sig = synth.generate_device([st], n_sig)[0]
    g = torch.Generator(device="cuda").manual_seed(5)
    noise = (torch.randn((n_noise, 2), device="cuda", generator=g) * 500).to(torch.int16)
    iq = torch.cat((noise, sig)).contiguous()
    soft, rep = demodulate_recording_native(C1, iq, carrier_seed="spectrum", max_pilot_samples=600_000)
    assert not rep.pilot_locked and rep.pilot_samples < 700_000
    sym_at_signal = int(n_noise * 72000 / 230000)
    tile_sym = 20536
    assert sym_at_signal - tile_sym <= rep.first_lock_symbol <= sym_at_signal + 2 * tile_sym, rep.first_lock_symbol
    assert rep.weak_carrier_tiles >= 10                                  # the noise tiles have no carrier line
    # the serial run needs its sweep to find the carrier; compare where it is locked too, up to the quarter-turn ambiguity
    serial, tr, ev = O.oracle_demod(C1, iq.cpu().numpy(), True)
    out = soft.cpu().numpy()
    L = 600_000
    assert tr["locked"][-L:].all()
    best = 0.0
    for r in range(4):
        rot = rotate_symbols(torch.from_numpy(out[-L - 4:]).unsqueeze(0), torch.tensor([r]))[0].numpy()
        for s in range(-2, 3):
            best = max(best, ((rot[4 + s: 4 + s + L - 8] >= 0) == (serial[-L: -8] >= 0)).all(axis=1).mean())
    assert best > 0.9999
    # the stitched stream is on the data well before the serial run is: right after the gate opens
    k = rep.first_lock_symbol + 2 * tile_sym
    assert np.abs(out[k: k + 20000].astype(int)).mean() > 45             # locked constellation (~60), not noise (~30)


@pytest.mark.gpu
def test_native_stitcher_needs_no_more_output_room_than_it_writes(gpu_device):
    """The CLI sizes the output for the nominal symbol rate + 2 % + 4096 (host/meteor_demod_amd.c).  A recording shorter
    than one pilot block must fit in that (the pilot used to insist on room for one symbol per sample of its block), a
    buffer that is really too small must be refused, and the byte-exact prefix must be what pilot_symbols says."""
    from meteor_demod_amd._capi import MdemodError
    from meteor_demod_amd.recording import demodulate_recording_native
    st = synth.make_stream(31, 230000, 72000, f0_hz=100.0, esn0_db=12.0)
    for n in (1, 1000, 100_000, 400_000):
        iq = synth.generate_device([st], n)[0]
        serial = O.oracle_demod(C1, iq.cpu().numpy())[0]
        cap = int(n * 72000 / 230000 * 1.02) + 4096
        soft, rep = demodulate_recording_native(C1, iq, carrier_seed="spectrum", soft_capacity=cap)
        out = soft.cpu().numpy()
        assert abs(len(out) - len(serial)) <= 1 and np.array_equal(out[: rep.pilot_symbols], serial[: rep.pilot_symbols])
        if rep.n_tiles == 0:
            assert np.array_equal(out, serial)
    with pytest.raises(MdemodError):
        demodulate_recording_native(C1, iq, carrier_seed="spectrum", soft_capacity=1000)
    # first seam against the pilot may drop the pilot's last symbol (no second pass): the exact prefix shrinks with it
    iq = synth.generate_device([st], 1_200_000)[0]
    serial = O.oracle_demod(C1, iq.cpu().numpy())[0]
    for pre in (0, 1000, 16384):
        soft, rep = demodulate_recording_native(C1, iq, refine=False, pre_samples=pre, pilot_margin_symbols=2000, tile_samples=20032)
        out = soft.cpu().numpy()
        assert np.array_equal(out[: rep.pilot_symbols], serial[: rep.pilot_symbols])


@pytest.mark.gpu
@pytest.mark.parametrize("oqpsk,bps", [(False, 16), (True, 32), (False, 8)])
def test_estimate_carrier_entry(gpu_device, oqpsk, bps):
    """mdemod_estimate_carrier (z^4 -> boxcar decimation -> FFT in LDS -> peak, one kernel): the estimate is the synthetic
    carrier, it agrees with the full-resolution torch.fft estimator of recording.py, the quality figure separates signal
    from noise, windows past the end of the recording are served, and the window length is rounded as documented."""
    import dataclasses
    import torch
    from meteor_demod_amd.recording import carrier_estimates, estimate_carrier_native
    symrate = 80000 if oqpsk else 72000
    cfg = dataclasses.replace(C1, symrate=symrate, oqpsk=oqpsk, bps=bps)
    amp = {8: dict(rms=40.0, dc=(1.5, -1.0)), 16: dict(rms=3000.0), 32: dict(rms=0.25, dc=(0.001, -0.002))}[bps]
    f0 = -437.0
    st = synth.make_stream(61, 230000, symrate, f0_hz=f0, clock_ppm=3.0, esn0_db=12.0, oqpsk=oqpsk, fmt=bps, **amp)
    n = 600_000
    iq = synth.generate_device([st], n)[0]
    noise = (torch.randn((200_000, 2), device="cuda") * (amp["rms"] / 1.4))
    noise = noise.to(iq.dtype) if bps != 8 else (noise + 128).clamp(0, 255).to(torch.uint8)
    both = torch.cat((iq, noise)).contiguous()
    starts = np.array([0, 100_000, 333_333, n - 65536, n + 50_000, n + 190_000])     # signal x4, noise, window past the end
    freq, qual, used = estimate_carrier_native(cfg, both, starts, 100_000)
    assert used == 65536
    steps = 2 if oqpsk else 1
    want = 2 * np.pi * f0 / (symrate * steps)
    f, q = freq.cpu().numpy(), qual.cpu().numpy()
    assert np.abs(f[:4] - want).max() < 2 * np.pi * 1.0 / (symrate * steps)          # within 1 Hz
    assert q[:4].min() > 20 and q[4] < 7
    ft, qt = carrier_estimates(both, starts[:4], 65536, 230000, symrate, nco_steps_per_symbol=steps)
    assert np.abs(ft.cpu().numpy() - f[:4]).max() < 2 * np.pi * 0.5 / (symrate * steps)
    assert np.isfinite(f).all() and np.isfinite(q).all()
    f2, q2, used2 = estimate_carrier_native(cfg, both, starts[:2], 5000)              # short windows: 4096 samples
    assert used2 == 4096 and np.abs(f2.cpu().numpy() - want).max() < 2 * np.pi * 12.0 / (symrate * steps)
