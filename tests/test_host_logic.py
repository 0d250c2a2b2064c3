"""CPU tests of the product's host side: the C-ABI library loads and exports every
symbol the header declares, init-time tables equal the reference's, and the
library fails loudly (no CPU fallback) when no GPU is present."""
from __future__ import annotations

import ctypes as C
import re
import sys
from pathlib import Path

import numpy as np
import pytest

from conftest import ROOT, load_npz
from golden_cases import CASES, sha
from meteor_demod_amd import DemodConfig, _capi, derive_tables, scale_freq_max, synth
from meteor_demod_amd._capi import MdemodParams


def _declared_symbols() -> list[str]:
    return _symbols_of(ROOT / "include" / "meteor_demod_amd.h")


def _symbols_of(path):
    text = re.sub(r"/\*.*?\*/", "", path.read_text(), flags=re.S)
    return sorted(set(re.findall(r"\b(mdemod_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_are_all_exported():
    lib = _capi.lib()
    names = _declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/meteor_demod_amd.h but not exported"
    # the drop-in boundary stays small: what replaces demod.h:29-50 and the five getters, the batch calls, one recording entry
    assert len(names) <= 28, names            # (r06: + mdemod_last_error, + mdemod_fanin_peer)
    # and the python binding table covers exactly the public header plus the internal one (stitcher primitives, self-tests)
    internal = _symbols_of(ROOT / "meteor_demod_amd" / "csrc" / "mdemod_internal_api.h")
    for n in internal:
        assert hasattr(lib, n), f"{n} declared in csrc/mdemod_internal_api.h but not exported"
    assert sorted(_capi.SIGNATURES) == sorted(set(names) | set(internal))


def test_abi_version_and_struct_layouts():
    lib = _capi.lib()
    assert lib.mdemod_abi_version() == 5
    assert C.sizeof(_capi.MdemodParams) == 48
    assert C.sizeof(_capi.MdemodStatus) == 56
    assert C.sizeof(_capi.MdemodLockEvent) == 16
    assert C.sizeof(_capi.MdemodStreamState) == 80
    assert C.sizeof(_capi.MdemodRecordingOpts) == 56 and C.sizeof(_capi.MdemodRecordingReport) == 112
    # the C compiler agrees with the ctypes mirrors
    import subprocess, tempfile
    from conftest import ROOT
    src = ('#include <stdio.h>\n#include "meteor_demod_amd.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(mdemod_params), '
           'sizeof(mdemod_status), sizeof(mdemod_lock_event), sizeof(mdemod_stream_state), sizeof(mdemod_recording_opts), '
           'sizeof(mdemod_recording_report)); return 0;}')
    with tempfile.TemporaryDirectory() as td:
        (Path(td) / "s.c").write_text(src)
        subprocess.run(["gcc", "-I", str(ROOT / "include"), str(Path(td) / "s.c"), "-o", str(Path(td) / "s")], check=True)
        out = subprocess.run([str(Path(td) / "s")], capture_output=True, text=True, check=True).stdout.split()
    assert [int(x) for x in out] == [48, 56, 16, 80, 56, 112]
    assert lib.mdemod_strerror(-3).decode().startswith("HIP")


def test_product_library_does_not_link_the_oracle():
    """The shipped path must not depend on oracle/ in any form."""
    import subprocess
    out = subprocess.run(["ldd", str(_capi.LIB_PATH)], capture_output=True, text=True).stdout
    assert "oracle" not in out and "lrpt_oracle" not in out
    syms = subprocess.run(["nm", "-D", str(_capi.LIB_PATH)], capture_output=True, text=True).stdout
    assert "orc_" not in syms
    forbidden = re.compile(r"oracle_py|lrpt_oracle|\borc_[a-z]|oracle/_ref|import\s+oracle|from\s+oracle")
    for src in list((ROOT / "meteor_demod_amd").rglob("*.py")) + list((ROOT / "meteor_demod_amd" / "csrc").iterdir()) \
            + list((ROOT / "host").glob("*.c")):
        assert not forbidden.search(src.read_text()), src


@pytest.mark.parametrize("tag", ["c1", "c3", "c4", "odd"])
def test_product_rrc_table_matches_reference(tag, manifest):
    """filter_init_rrc known-answer (filter.c:10-28,71-94) for the product's own host code."""
    cfg = DemodConfig(**manifest["tables"][f"rrc_{tag}"]["cfg"])
    rrc, consts, lut = derive_tables(cfg)
    g = load_npz("tables")
    assert np.array_equal(rrc, g[f"rrc_{tag}"])
    assert np.array_equal(lut, g["tanh_lut"])


def test_product_loop_constants_match_survey_a7():
    """SURVEY App. A.7 (measured from the reference): derived constants per config."""
    _, c, _ = derive_tables(DemodConfig(samplerate=230000))
    assert c["pll_alpha"] == np.float32(1.23405785e-04) and c["pll_beta"] == np.float32(7.61496555e-09)
    assert c["pll_fmax"] == np.float32(0.3)
    assert c["t_center"] == np.float32(0.393382043) and c["t_maxdev"] == np.float32(9.60405378e-05)
    assert c["t_alpha"] == np.float32(3.99991986e-05) and c["t_beta"] == np.float32(3.99991956e-10)
    _, c, _ = derive_tables(DemodConfig(samplerate=230000, symrate=80000, oqpsk=True))
    assert c["pll_alpha"] == np.float32(2.22119459e-04) and c["pll_fmax"] == np.float32(0.15)
    assert c["t_center"] == np.float32(0.437091142)
    _, c, _ = derive_tables(DemodConfig(samplerate=1000000, rrc_order=64, interp_factor=8))
    assert c["t_center"] == np.float32(0.0565486662) and c["t_alpha"] == np.float32(2.49996865e-05)


def test_product_constants_equal_oracle_constants():
    import oracle_py as O
    for case in CASES:
        _, c, lut = derive_tables(case.cfg)
        st = O.OracleStream(case.cfg)
        for k in ("pll_alpha", "pll_beta", "pll_fmax", "t_alpha", "t_beta", "t_center", "t_maxdev", "osf"):
            assert np.float32(getattr(st.consts, k)) == c[k], (case.name, k)


def test_bad_parameters_are_rejected():
    lib = _capi.lib()
    # samplerate 17000 at 72k symbols/s: less than a quarter of a sample per symbol (down to there the reference's
    # keep-the-last-symbol-of-a-sample rule is what the kernels do: goldens one_per_symbol, sub_sample*)
    for bad in (dict(interp_factor=0), dict(rrc_order=0), dict(samplerate=0), dict(symrate=-1), dict(bps=12), dict(samplerate=17000)):
        cfg = DemodConfig(samplerate=230000)
        for k, v in bad.items():
            setattr(cfg, k, v)
        p = cfg.to_c()
        assert lib.mdemod_derive_tables(C.byref(p), None, 0, None, None) == _capi.MDEMOD_ERR_PARAM
        ctx = C.c_void_p()
        assert lib.mdemod_create(C.byref(p), C.byref(ctx)) == _capi.MDEMOD_ERR_PARAM
    p = DemodConfig(samplerate=230000).to_c(n_streams=0)
    assert lib.mdemod_create(C.byref(p), C.byref(C.c_void_p())) == _capi.MDEMOD_ERR_PARAM


def test_no_gpu_means_loud_failure_not_cpu_fallback():
    """Without a HIP device mdemod_create must fail with MDEMOD_ERR_HIP."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from meteor_demod_amd import Demodulator
    with pytest.raises(_capi.MdemodError) as ei:
        Demodulator(DemodConfig(samplerate=230000), 1)
    assert ei.value.code == _capi.MDEMOD_ERR_HIP


def test_freq_max_scaling_follows_main_c():
    """main.c:136: -d in Hz -> rad/symbol in double then narrowed; default -1 stays negative."""
    assert scale_freq_max(-1, 72000) < 0
    assert scale_freq_max(3500.0, 72000.0) == float(np.float32(3500.0 * 2 * np.pi / 72000.0))


def test_synth_is_deterministic_and_addressable():
    st = synth.make_stream(42, 230000, 72000, f0_hz=777.0, clock_ppm=13.0)
    a = synth.generate_host(st, 5000)
    b = synth.generate_host(st, 3000, n0=2000)
    assert np.array_equal(a[2000:], b)
    assert sha(a) == "%s" % sha(synth.generate_host(st, 5000))
    # format variants share the same underlying signal
    st8 = synth.make_stream(42, 230000, 72000, f0_hz=777.0, clock_ppm=13.0, fmt=8, rms=60.0)
    u = synth.generate_host(st8, 1000)
    assert u.dtype == np.uint8 and 100 < u.mean() < 156


def test_synth_signal_is_what_it_claims():
    """Power, carrier offset and symbol rate of the synthetic recording."""
    fs, rs, f0 = 230000.0, 72000.0, 1200.0
    st = synth.make_stream(7, fs, rs, f0_hz=f0, esn0_db=30.0, dc=(0.0, 0.0))
    x = synth.generate_host(st, 1 << 16).astype(np.float64)
    z = x[:, 0] + 1j * x[:, 1]
    assert abs(np.sqrt(np.mean(np.abs(z) ** 2)) - 6000.0) < 300.0
    # 4th power removes QPSK modulation: line at 4*f0
    spec = np.abs(np.fft.fft(z ** 4))
    peak = np.fft.fftfreq(z.size, 1 / fs)[np.argmax(spec)]
    assert abs(peak - 4 * f0) < 20.0


def _run_proof(name, *flags, args=()):
    import subprocess
    import tempfile
    src = ROOT / "tools" / "proofs" / name
    with tempfile.TemporaryDirectory() as td:
        exe = Path(td) / "verify"
        subprocess.run(["g++", "-O2", "-ffp-contract=off", *flags, str(src), "-o", str(exe)], check=True)
        out = subprocess.run([str(exe), *args], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    return out.stdout


def test_turn_code_by_multiplication_proof_holds_on_the_host():
    """tools/proofs/verify_turncode_mul.cpp: fast_sin's turn code as ONE double multiplication (demod_device.h: md_turn_code)
    equals the reference's double division for every float |x| < 16 (exhaustive, ~10 CPU-seconds), with the constant 3 ulp
    either side as well (margin)."""
    out = _run_proof("verify_turncode_mul.cpp", "-pthread")
    assert out.count(": mismatches 0") == 7, out


def test_period_wrap_in_float_arithmetic_proof_holds_on_the_host():
    """tools/proofs/verify_wrap_f32.cpp: (float)((double)x -+ 2*M_PI) == (x -+ C_HI) -+ C_LO in float arithmetic for every float
    with 2pi <= |x| < 4pi (demod_device.h: md_nco_advance<true>, md_wrap_2pi)."""
    out = _run_proof("verify_wrap_f32.cpp")
    assert "checked 16777216 floats" in out and "mismatches 0" in out, out


def test_symbol_clock_closed_form_proof_holds_on_the_host():
    """tools/proofs/verify_clock_jump.cpp: the closed-form blind steps of configs[3]'s symbol clock (csrc/clock_jump.h, the very header
    the kernel compiles) against the reference's sequential rounded additions (timing.c:32-38): every clock word the instance can be
    launched with (153 009 floats), both ends of the launcher's bound, 64 starting phases each here (the full run behind DESIGN.md
    took 2 048: 627 M cases) - same final phase bit for bit, same step count, the firing always found by the checked additions."""
    out = _run_proof("verify_clock_jump.cpp", "-pthread", args=["1", "64"])
    assert "153009 clock words" in out and ": 0 mismatches" in out and "8368 words with a tie" in out, out


def test_symbol_clock_closed_form_at_any_rate_holds_on_the_host():
    """The same closed form with the schedule as numbers (cj_schedule / clock_jump_run: what the v3 kernels run when a firing's run of
    steps is long - sample rates from about 1.8 MS/s): 14 sample rates x QPSK / OQPSK x six -O, every run the host would schedule,
    clock words over the whole range the loop allows plus words that tie in some binade, starting phases over the window the kernel
    checks per lane."""
    out = _run_proof("verify_clock_jump.cpp", "-pthread", args=["any", "400", "64"])
    assert "168 configurations, 89 runs with a schedule" in out and ": 0 mismatches" in out, out


def test_cli_device_plan_is_round_robin():
    """host/meteor_demod_amd.c --devices a,b,c --plan: the sharding arithmetic of the C host without files or GPUs - file i on the
    (i mod G)-th device of the list, never more workers than files."""
    import subprocess
    cli = ROOT / "meteor_demod_amd" / "lib" / "meteor_demod_amd"
    r = subprocess.run([str(cli), "--devices", "0,2,5", "--plan", "a", "b", "c", "d", "e", "f", "g"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout.splitlines() == ["device 0: a d g", "device 2: b e", "device 5: c f"]
    r = subprocess.run([str(cli), "--devices", "3,1", "--plan", "only"], capture_output=True, text=True)
    assert r.stdout.splitlines() == ["device 3: only"]
    assert subprocess.run([str(cli), "--devices", "0,x", "--plan", "a"], capture_output=True).returncode == 1


def test_rotating_window_register_partition():
    """The v3 kernels hand the top of the VGPR file (and of the AccVGPR file) to hand-written assembly and keep hipcc below with
    `amdgpu_num_vgpr`, whose unit is an observed property of the compiler (two registers on gfx90a+), not a documented one: check
    the emitted assembly of EVERY assembly-owning file (demod_kernel_rot.hip: std + hybrid; demod_kernel_rotp.hip: wide / mid /
    far, s16 / u8, the configs[3] instance among them) - no compiler-generated instruction may touch a register of a window or of
    the coefficient buffers, no scratch inside the main loops, no scratch at all in the std kernels.  build() runs the same check on
    the assembly of the objects it links and fails the BUILD on a violation; here it runs on that assembly (or compiles afresh)."""
    from meteor_demod_amd import build as B
    files = {"demod_kernel_rot": B.LIB / "demod_kernel_rot.gfx950.s", "demod_kernel_rotp": B.LIB / "demod_kernel_rotp.gfx950.s"}
    assert B.check_rot_partition(files) == []
    # every kernel of both files is covered by a limit, and the limits are the generated headers' own
    limits = B._asm_limits()
    assert limits["demod_kernel_rot"][0] == 80 and len([k for k in limits if k.startswith("demod_kernel_rotp_")]) == 6


def test_register_partition_check_catches_a_violation():
    """A deliberately lowered limit must be reported (so that a compiler bump that moves a value into the assembly's registers
    fails the build): the compiler does use the registers just below each limit."""
    from meteor_demod_amd import build as B
    limits = B._asm_limits()
    for stem, key in (("demod_kernel_rot", "demod_kernel_rot"), ("demod_kernel_rotp", "demod_kernel_rotp_WIDE_16_")):
        path = B.LIB / (stem + ".gfx950.s")
        if not path.exists():
            B.build()
        text = path.read_text()
        assert B.scan_asm_partition(text, limits, B.SCRATCH_FREE) == []
        lowered = dict(limits)
        lowered[key] = (limits[key][0] - 24, limits[key][1])
        bad = B.scan_asm_partition(text, lowered, B.SCRATCH_FREE)
        assert bad and all(key.rstrip("_") in b for b in bad), (key, bad[:3])
    # a kernel that must be scratch-free and is not: the first std kernel's metadata says it spills
    import re
    text = (B.LIB / "demod_kernel_rot.gfx950.s").read_text()
    at = re.search(r"^_ZN\w*demod_kernel_rotILi\w+:", text, re.M).start()
    k = text.index("ScratchSize: 0", at)
    fake = text[:k] + "ScratchSize: 16" + text[k + len("ScratchSize: 0"):]
    assert any("scratch" in b for b in B.scan_asm_partition(fake, limits, B.SCRATCH_FREE))


def test_carrier_window_rounding_is_host_logic():
    """mdemod_carrier_window_samples needs no GPU: power of two in [4096, 2^18], and no more than 16384 points after the
    decimation the z^4 band allows (at 2.5 samples per symbol only 4x decimation keeps +-4*0.33 rad/symbol inside 80 % of
    the band; at one sample per symbol none does, so the window itself shrinks to the 16384 points that fit in LDS)."""
    import ctypes as C
    from meteor_demod_amd import DemodConfig, _capi
    lib = _capi.lib()
    def used(samplerate, symrate, want):
        p = DemodConfig(samplerate=samplerate, symrate=symrate).to_c(1, 0)
        return int(lib.mdemod_carrier_window_samples(C.byref(p), want))
    assert used(230000, 72000, 100_000) == 65536 and used(230000, 72000, 5000) == 4096 and used(230000, 72000, 100) == 4096
    assert used(230000, 72000, 1 << 20) == 65536            # 16384 points x 4 (decimation by 8 would fold the band)
    assert used(1_022_400, 72000, 1 << 20) == 262144        # 16x decimation available: 16384 points x 16
    assert used(80000, 72000, 100_000) == 32768             # 1.1 samples per symbol: decimation by 2 still keeps the band
    assert used(73000, 72000, 100_000) == 16384             # barely oversampled: no decimation, 16384 points
    assert lib.mdemod_carrier_window_samples(None, 4096) == 0


# ---- the C host's full-screen display (host/tui.c; the reference's tui.c) ----------------------------------------------

# what the reference's own humanize() / seconds_to_str() (utils.c:22-57) print for these values (recorded from a scratch build of
# utils.c in this container)
TUI_FORMATS = """size 0 -> [0  B]
size 999 -> [999  B]
size 1000 -> [1000  B]
size 1001 -> [1.00 kB]
size 12345 -> [12.3 kB]
size 123456 -> [123 kB]
size 23456789 -> [23.5 MB]
size 4000000000 -> [4.00 GB]
clock 0 -> 00:00:00
clock 59 -> 00:00:59
clock 3725 -> 01:02:05
clock 356400 -> 99:00:00
clock 360000 -> 00:00:00
"""


def _cli():
    from meteor_demod_amd import build
    exe = build.LIB / "meteor_demod_amd"
    if not exe.exists():
        pytest.skip("C host not built")
    return exe


def test_tui_formats_are_the_references():
    import subprocess
    r = subprocess.run([str(_cli()), "--tui-selftest"], stdin=subprocess.DEVNULL, capture_output=True, text=True)
    if "built without ncurses" in r.stderr:
        pytest.skip("no ncurses in this image")
    assert r.returncode == 0
    assert r.stdout.startswith(TUI_FORMATS), r.stdout
    plot = [l[6:-1] for l in r.stdout.splitlines() if l.startswith("plot |")]
    assert len(plot) == 9 and all(len(l) == 19 for l in plot)
    # four clusters at (+-90, +-90): columns 9 +- 90*19/255 = 3 / 15, rows 4 -+ 90*9/255 = 1 / 7 (and their neighbours: +-10 of noise)
    for r_, c_ in ((1, 3), (1, 15), (7, 3), (7, 15)):
        assert plot[r_][c_] == "#", plot
    assert all(ch == " " for ch in plot[4]) and all(l[9] == " " for l in plot)     # nothing on the axes


@pytest.mark.parametrize("how", ["terminal", "pipes"])
def test_tui_draws_its_panes(how):
    """No GPU call: --tui-selftest draws one frame of made-up values when stdin and stdout are a terminal, or anywhere with --tui."""
    from ptyrun import run_in_pty, run_on_pipes
    if how == "terminal":
        try:
            rc, screen, err = run_in_pty([str(_cli()), "--tui-selftest"])
        except OSError as e:
            pytest.skip(f"no pseudo-terminals here: {e}")
    else:
        rc, screen, err = run_on_pipes([str(_cli()), "--tui", "--tui-selftest"])
    if "built without ncurses" in err:
        pytest.skip("no ncurses in this image")
    assert rc == 0, err
    for piece in ("Input: selftest.wav, output: selftest.s", "Demodulator initialized", "PLL status: Locked", "Carrier freq", "+1234.5 Hz",
                  "72000.1 Hz", "0.031", "Data in", "00:01:08/00:02:05 (54.5%)", "Data out", "23.5 MB", "Demodulation complete",
                  "Press any key to exit..."):
        assert piece in screen, (piece, screen)


# ---- which kernel a configuration gets (host-only planning: mdemod_plan_kernel) ------------------------------------------

PLAN_CASES = {
    # (DemodConfig kwargs, forced generation flag) -> piece of the kernel's name
    "configs[1]": (dict(samplerate=230000), 0, "v3 rotating register window"),
    "configs[2] oqpsk": (dict(samplerate=230000, symrate=80000, oqpsk=True), 0, "v3 rotating register window"),
    "configs[3] wide": (dict(samplerate=1000000, rrc_order=64, interp_factor=8), 0, "v3 rotating packed window, wide"),
    "1.024 MS/s mid": (dict(samplerate=1024000), 0, "v3 rotating packed window, mid"),
    "1.8 MS/s far": (dict(samplerate=1800000), 0, "v3 rotating packed window, far"),
    "3.2 MS/s far (44 samples per firing)": (dict(samplerate=3200000), 0, "v3 rotating packed window, far"),
    "3.4 MS/s: past the far window -> gather": (dict(samplerate=3400000), 0, "v3 gather: sample rates beyond every window, 65 taps"),
    "10 MS/s -> gather": (dict(samplerate=10000000), 0, "v3 gather"),
    "4 MS/s long filter -> gather": (dict(samplerate=4000000, rrc_order=64, interp_factor=4), 0, "v3 gather: sample rates beyond every window, 129 taps"),
    "6 MS/s u8 -> gather": (dict(samplerate=6000000, bps=8), 0, "v3 gather"),
    "6 MS/s float -> gather": (dict(samplerate=6000000, bps=32), 0, "v3 gather: sample rates beyond every window, 65 taps"),
    "6 MS/s float, long filter: v1": (dict(samplerate=6000000, rrc_order=64, interp_factor=4, bps=32), 0, "v1 LDS ring"),
    "2.048 MS/s long filter: wide": (dict(samplerate=2048000, rrc_order=64, interp_factor=4), 0, "v3 rotating packed window, wide"),
    "float std": (dict(samplerate=230000, bps=32), 0, "v3 rotating register window"),
    "float mid -> hybrid": (dict(samplerate=1024000, bps=32), 0, "v3 hybrid window, mid"),
    "float 2.048 MS/s -> hybrid mid": (dict(samplerate=2048000, bps=32), 0, "v3 hybrid window, mid"),
    "float long filter -> hybrid": (dict(samplerate=1000000, rrc_order=64, interp_factor=8, bps=32), 0, "v3 hybrid window: float input, 129 taps"),
    "float long filter at 2.048 MS/s": (dict(samplerate=2048000, rrc_order=64, interp_factor=4, bps=32), 0, "v3 hybrid window: float input, 129 taps"),
    "float 3.2 MS/s -> hybrid far": (dict(samplerate=3200000, bps=32), 0, "v3 hybrid window, far"),
    "float 4 MS/s -> gather (55.6 samples per firing)": (dict(samplerate=4000000, bps=32), 0, "v3 gather"),
    "-O 32: compact table keeps v3": (dict(samplerate=230000, interp_factor=32), 0, "v3 rotating register window"),
    "161 taps: v1": (dict(samplerate=230000, rrc_order=80), 0, "v1 LDS ring"),
    "161 taps x 64 banks: table in global memory": (dict(samplerate=1000000, rrc_order=80, interp_factor=64), 0, "[table in global memory]"),
    "v1 forced": (dict(samplerate=230000), 1, "v1 LDS ring"),
}


@pytest.mark.parametrize("case", list(PLAN_CASES))
def test_kernel_plan_of_a_configuration(case):
    """Geometry selection is host arithmetic (csrc/demod_host.cpp, plan_context in csrc/demod_api.cpp): no device needed."""
    import ctypes as C
    from meteor_demod_amd import DemodConfig, _capi
    kw, flag, want = PLAN_CASES[case]
    lib = _capi.lib()
    p = DemodConfig(**kw).to_c(100000, 0)
    p.reserved = flag | 0x4                     # generation + MDEMOD_FLAG_LAT_OFF: the lane kernel's plan, whatever the stream count
    name = C.create_string_buffer(256)
    lds, block = C.c_uint32(), C.c_uint32()
    rc = lib.mdemod_plan_kernel(C.byref(p), name, 256, C.byref(lds), C.byref(block))
    assert rc == 0, rc
    assert want in name.value.decode(), name.value
    assert 0 < lds.value <= 160 * 1024 and block.value in (64, 128, 192, 256, 512), (lds.value, block.value)


def test_the_retired_kernel_generation_is_refused():
    """MDEMOD_FLAG_KERNEL_MASK = 2 named round 2's moving register window; it was retired in round 4 (every geometry it had is a
    v3 kernel now): asking for it is a parameter error, not a silent substitution."""
    import ctypes as C
    from meteor_demod_amd import DemodConfig, _capi
    p = DemodConfig(samplerate=230000).to_c(1000, 0)
    p.reserved = 2
    name = C.create_string_buffer(64)
    assert _capi.lib().mdemod_plan_kernel(C.byref(p), name, 64, None, None) == -1


def test_bench_bound_block_reads_the_tracked_profile():
    """bench.py's per-configuration bound / ceiling block from profiles/hbm_traffic.json, without a GPU: every BASELINE
    configuration has its counters, the ceiling lies above what is achieved and far below BASELINE.json's 0.40 for configs[1] / [2]."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", ROOT / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    ms = {"c1": 24.2, "c3": 48.8, "c4": 14.6}
    for tag in ("c1", "c3", "c4"):
        cfg, _ = bench.demod_config(tag)
        b = bench.bound_block(cfg, tag, 393216, 16448, ms[tag])
        v = b["valu"]
        assert b["traffic"] and 1.0 < b["traffic_ratio"] < 1.5, (tag, b["traffic_ratio"])
        assert v["valu_instructions_per_wave_firing"] and 0.7 < v["simd_valu_busy_frac"] < 1.0
        assert v["fir_packed_instructions_per_firing"]["floor"] == 2 * cfg.taps
        assert b["hbm_frac"] < v["pipe_busy_ceiling_hbm_frac"] <= v["ceiling_hbm_frac"] < 0.40, (tag, b["hbm_frac"], v)
        assert 0.7 < v["frac_of_measured"] < 1.0
    assert abs(bench.flops_per_sample_of(bench.demod_config("c1")[0]) - 122.7) < 0.5


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", ROOT / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    return bench


LINE_REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                 "data", "config", "roofline", "cpu_baseline")


def _assert_line_ok(line: str, bench, n1: bool = True):
    """The contract of the ONE stdout line (a 23 KB line was cut by the driver's 8 KB tail and the round went
    unmeasured): under the hard bound, required keys present and FIRST, the roofline / cpu_baseline objects complete, no long strings."""
    import json
    assert len(line) < bench.LINE_HARD_BYTES, len(line)
    d = json.loads(line)
    need = LINE_REQUIRED if n1 else LINE_REQUIRED[:-1]
    assert list(d)[: len(need)] == list(need), list(d)
    assert set(("workload", "tiles_per_gpu", "tile_samples", "samples_per_step")) <= set(d["config"])
    assert "model" not in d["config"]
    r = d["roofline"]
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic", "kernel", "kernel_ms",
                "algorithmic_bytes_per_sample", "ceiling_hbm_frac", "valu")) <= set(r), sorted(r)
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    if n1:
        assert set(("value", "unit", "cores", "kind", "sample")) <= set(d["cpu_baseline"]), d["cpu_baseline"]

    def walk(o, path=""):
        if isinstance(o, dict):
            for k, v in o.items():
                assert len(k) <= 64, (path, k)
                walk(v, path + "/" + k)
        elif isinstance(o, list):
            for v in o:
                walk(v, path)
        elif isinstance(o, str):
            assert len(o) <= bench.LINE_MAX_STRING, (path, len(o))
    walk(d)
    return d


def test_bench_line_is_compact_and_complete():
    """bench.compact_line on the tracked full records of rounds 5 and 6 (profiles/r0N_bench_line.json: what bench.py gathered on
    the GPU box) and on a bloated one: the stdout line stays under 6 KB (target 4 KB) and keeps every key the driver parses."""
    import copy
    import json
    bench = _bench_module()
    seen = 0
    for name in ("r05_bench_line.json", "r06_bench_full.json"):
        f = ROOT / "profiles" / name
        if not f.exists():
            continue
        full = json.loads(f.read_text())
        if full["roofline"]["bound"] == "valu":                  # (round 5 wrote the binding resource here; the contract's values are hbm | mfma)
            full["roofline"]["bound"] = "hbm"
        line = json.dumps(bench.compact_line(full))
        d = _assert_line_ok(line, bench)
        assert len(line) <= bench.LINE_TARGET_BYTES, len(line)
        assert d["value"] == full["value"] and d["roofline"]["kernel_ms"] == full["roofline"]["kernel_ms"]
        assert d["useful"]["whole_buffer_msps"] == full["single_recording"]["configs[1] whole buffer"]["msamples_per_s"]
        assert d["useful"]["c2_2p28_msps"] == full["single_recording"]["configs[1] 2^28 samples"]["msamples_per_s"]
        assert "single_recording" not in d and "cli_wall_times" not in d                      # the long blocks stay in the extras
        # a record that grew (more extras, long notes everywhere) must not grow the line
        fat = copy.deepcopy(full)
        fat["single_recording"]["more"] = {"x" * 40 + str(i): "y" * 500 for i in range(200)}
        fat["config"]["workload"] = fat["config"]["workload"] * 5
        fat["check"] = "z" * 5000
        fat["cpu_baseline"]["sample"] = "s" * 3000
        fat["something_new"] = {"k": "v" * 10000}
        _assert_line_ok(json.dumps(bench.compact_line(fat)), bench)
        seen += 1
    assert seen


def test_bench_emitter_prints_once_even_from_the_watchdog(tmp_path):
    """bench.Emitter: one line whatever happens after the timed region.  A child process builds a record,
    then (a) finishes normally, (b) hangs in a "collective" until the watchdog prints and ends the process with rc 0, (c) is sent
    SIGTERM while hung, as a launcher does when another rank dies.  Exactly one parseable line each time."""
    import json
    import signal
    import subprocess
    import time
    full = json.loads((ROOT / "profiles" / "r05_bench_line.json").read_text())
    full["roofline"]["bound"] = "hbm"
    rec = tmp_path / "full.json"
    rec.write_text(json.dumps(full))
    code = ("import sys, json, time, importlib.util\n"
            f"spec = importlib.util.spec_from_file_location('bench_mod', {str(ROOT / 'bench.py')!r}); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)\n"
            f"b.EXTRAS_FILE = __import__('pathlib').Path({str(tmp_path / 'extras.json')!r})\n"
            f"full = json.load(open({str(rec)!r}))\n"
            "em = b.Emitter(0, full, float(sys.argv[2]))\n"
            "em.stage = 'fanin'\n"
            "print('READY', file=sys.stderr, flush=True)\n"
            "if sys.argv[1] == 'hang': time.sleep(600)\n"
            "em.emit(); em.emit()\n")
    bench = _bench_module()
    for mode, deadline, sig in (("ok", 30, None), ("hang", 1.0, None), ("hang", 300, signal.SIGTERM)):
        p = subprocess.Popen([sys.executable, "-c", code, mode, str(deadline)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        if sig:
            assert p.stderr.readline().strip() == "READY"
            time.sleep(0.3)
            p.send_signal(sig)
        out, err = p.communicate(timeout=60)
        assert p.returncode == 0, (mode, sig, p.returncode, err[-500:])
        lines = [l for l in out.splitlines() if l.startswith("{")]
        assert len(lines) == 1, (mode, sig, out[-500:])
        d = _assert_line_ok(lines[0], bench)
        assert d["value"] == full["value"]
        if mode == "hang":
            assert "fanin" in d["errors"]["post_region"] and ("watchdog" if sig is None else "SIGTERM") in d["errors"]["post_region"]
        else:
            assert "errors" not in d
        assert "bench.py full record: {" in err                                          # the long record went to stderr ...
        assert json.loads((tmp_path / "extras.json").read_text())["value"] == full["value"]   # ... and to the extras file


def test_no_exception_crosses_the_c_boundary():
    """Every int-returning extern "C" entry of the library is a function-try-block ending in MDEMOD_API_CATCH (mdemod_create has
    its own, which also gives the half-built context back): a std::bad_alloc inside the library must come back as
    MDEMOD_ERR_NOMEM, not as abort() in a caller written in C."""
    import os
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "meteor_demod_amd", "csrc")
    api = open(os.path.join(root, "demod_api.cpp")).read()
    block = api[api.index('extern "C" {'):api.index('} /* extern "C" */')]
    entries = re.findall(r"\nint\n(mdemod_\w+)\(([^{]*?)\)\n(try )?\{", block)
    assert len(entries) >= 30
    unguarded = [name for name, _args, guard in entries if not guard and name != "mdemod_create"]
    assert not unguarded, unguarded
    assert "mdemod_destroy(ctx);\n\t\treturn MDEMOD_ERR_NOMEM;" in block              # mdemod_create's own handler
    rec = open(os.path.join(root, "recording.hip")).read()
    for name, guard in re.findall(r'extern "C" int\n(mdemod_\w+)\([^{]*?\)\n(try )?\{', rec):
        assert guard, name
    assert block.count("} MDEMOD_API_CATCH") == len(entries) - 1


@pytest.mark.parametrize("kw", [dict(samplerate=230400, symrate=80000, oqpsk=True, interp_factor=5), dict(samplerate=1080000, interp_factor=4),
                                dict(samplerate=230400, symrate=80000, oqpsk=True, interp_factor=5, rrc_order=48)],
                         ids=["oqpsk80k-230.4k-O5", "1.08M-O4", "order48"])
def test_a_filter_with_a_tap_that_is_not_finite_is_refused(kw, capfd):
    """filter.c:86-93 divides by zero where samples-per-symbol x -O / 2.4 lands on a tap; the reference then runs inf / NaN through its
    loops and indexes its tanh table with (int)NaN - its oracle restatement segfaults there as the reference does.  The library refuses
    such a table (MDEMOD_ERR_PARAM, with the way out in mdemod_last_error() - the library itself prints nothing since ABI 5, VERDICT r05
    item 6); one -O further the same rates are fine, and a call that succeeds leaves no text behind."""
    import ctypes as C
    import dataclasses
    from meteor_demod_amd import DemodConfig, _capi
    lib = _capi.lib()
    cfg = DemodConfig(**kw)
    name = C.create_string_buffer(200)
    p = cfg.to_c(1, 0)
    assert lib.mdemod_plan_kernel(C.byref(p), name, 200, None, None) == _capi.MDEMOD_ERR_PARAM
    said = _capi.last_error()
    assert "choose another -O" in said and "filter.c:86-93" in said and str(cfg.samplerate) in said
    rrc = (C.c_float * 8192)()
    assert lib.mdemod_derive_tables(C.byref(p), rrc, 8192, None, None) == _capi.MDEMOD_ERR_PARAM
    assert "choose another -O" in _capi.last_error()
    ctx = C.c_void_p()
    assert lib.mdemod_create(C.byref(p), C.byref(ctx)) == _capi.MDEMOD_ERR_PARAM and not ctx.value       # refused before any device is asked for
    assert "choose another -O" in _capi.last_error()
    p2 = dataclasses.replace(cfg, interp_factor=cfg.interp_factor + 1).to_c(1, 0)
    assert lib.mdemod_plan_kernel(C.byref(p2), name, 200, None, None) == 0
    assert _capi.last_error() == ""                                   # the text belongs to the failing call only
    out = capfd.readouterr()
    assert out.err == "" and out.out == "", out                       # nothing was printed by anybody


def test_last_error_says_which_setting_and_debug_prints_it():
    """mdemod_last_error() for the refusals of mdemod_host_derive, one text per reason, thread-local; with MDEMOD_DEBUG in the
    environment (a child process) the same text also goes to stderr, prefixed with the library's name."""
    import ctypes as C
    import dataclasses
    import subprocess
    import threading
    from meteor_demod_amd import DemodConfig, _capi
    lib = _capi.lib()
    name = C.create_string_buffer(200)
    base = DemodConfig(samplerate=230000)
    for change, needle in ((dict(interp_factor=0), "-O 0"), (dict(interp_factor=65), "-O 65"), (dict(rrc_order=0), "-f 0"), (dict(rrc_order=257), "-f 257"),
                           (dict(bps=12), "12 bits per sample"), (dict(samplerate=0), "must be positive"), (dict(samplerate=1000, symrate=72000), "quarter of an input sample"),
                           (dict(samplerate=2000000000, interp_factor=4), "overflows")):
        p = dataclasses.replace(base, **change).to_c(1, 0)
        assert lib.mdemod_plan_kernel(C.byref(p), name, 200, None, None) == _capi.MDEMOD_ERR_PARAM, change
        assert needle in _capi.last_error(), (change, _capi.last_error())
    # another thread has its own text (none)
    seen = []
    t = threading.Thread(target=lambda: seen.append(_capi.last_error()))
    t.start(); t.join()
    assert seen == [""] and "overflows" in _capi.last_error()
    # MdemodError carries it
    with pytest.raises(_capi.MdemodError) as e:
        _capi.check(lib.mdemod_plan_kernel(C.byref(dataclasses.replace(base, bps=12).to_c(1, 0)), name, 200, None, None), "mdemod_plan_kernel")
    assert "12 bits per sample" in str(e.value) and e.value.detail
    code = ("import ctypes as C, dataclasses\nfrom meteor_demod_amd import DemodConfig, _capi\nlib = _capi.lib()\n"
            "p = DemodConfig(samplerate=230000, bps=12).to_c(1, 0)\nprint(lib.mdemod_plan_kernel(C.byref(p), C.create_string_buffer(64), 64, None, None))\n")
    import os
    for dbg in (False, True):
        env = {k: v for k, v in os.environ.items() if k != "MDEMOD_DEBUG"}
        if dbg:
            env["MDEMOD_DEBUG"] = "1"
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=str(ROOT), env=env, timeout=120)
        assert r.returncode == 0 and r.stdout.strip() == "-1", r.stderr[-500:]
        assert ("meteor_demod_amd: 12 bits per sample" in r.stderr) == dbg, r.stderr[-500:]


@pytest.mark.parametrize("kw,most", [(dict(samplerate=230000), 4096), (dict(samplerate=230000, symrate=80000, oqpsk=True), 4096),
                                     (dict(samplerate=230000, rrc_order=48, bps=32), 4096),
                                     (dict(samplerate=1000000, rrc_order=64, interp_factor=8), 2048), (dict(samplerate=1000000, rrc_order=64, interp_factor=8, bps=32), 2048),
                                     (dict(samplerate=1024000), 1024), (dict(samplerate=1024000, bps=32), 1024), (dict(samplerate=2048000), 1024)],
                         ids=["configs1", "configs2", "97taps-f32-230k", "configs3", "configs3-f32", "mid-1.024M", "mid-1.024M-f32", "far-2.048M"])
def test_where_a_context_hands_over_from_the_wave_kernel_to_the_lane_kernels(kw, most):
    """wants_latency_kernel (demod_api.cpp, measured r05 with tools/lat_bench.py): one stream per WAVE up to `most` streams, one per lane
    of the configuration's v3 kernel from there on - decided without a device (mdemod_plan_kernel)."""
    import ctypes as C
    from meteor_demod_amd import DemodConfig, _capi
    lib = _capi.lib()
    cfg = DemodConfig(**kw)
    name = C.create_string_buffer(200)
    for n, wave in ((1, True), (most, True), (most + 1, False), (65536, False)):
        p = cfg.to_c(n, 0)
        assert lib.mdemod_plan_kernel(C.byref(p), name, 200, None, None) == 0
        assert (b"demod_kernel_lat" in name.value) == wave, (n, name.value)
