#!/usr/bin/env python3
"""bench.py — IQ Msamples/s demodulated on MI355X (BASELINE.json metric).

Workload at N=1 = BASELINE.json configs[1]: QPSK 72k, 230 kS/s int16, RRC order
32, oversampling 5, ONE device-resident synthetic IQ buffer cut into T tiles of
L samples; each tile is an independent stream demodulated bit-exactly by one GPU
lane (tiles of one recording are independent streams).  A "step"
is one pass of the fused demod kernel over the whole buffer (T*L samples), input
already in HBM.  For N>1 each rank owns its own buffer of the same shape (weak
scaling), no collective on the data path; at N > 1 the soft symbols are then
gathered on rank 0 over RCCL (outside the timed region; `fanin` in the JSON
line; `--no-fanin` skips it).

One JSON line on rank 0, with `roofline` (HBM, algorithmic bytes / measured
kernel time) and `cpu_baseline` (the reference's own code from oracle/_ref, or
the oracle port, timed on the host cores).  The same CPU leg also verifies sampled
tiles of the GPU output byte-for-byte against the oracle ("check"); with
--no-cpu-baseline nothing under oracle/ is touched.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


VALU_PEAK_TOPS = 78.6         # MI355X: 157.3 TFLOP/s FP32 vector counts FMAs; the reference's unfused mul/add get half of it


def flops_per_sample_of(cfg) -> float:
    """unfused flops per input sample = 2*interp + (symrate/fs) * (4*taps*F + ~100*F'), F = firings per symbol."""
    F = 2 if cfg.oqpsk else 1
    return 2 * cfg.interp_factor + (cfg.symrate / cfg.samplerate) * (4 * (2 * cfg.rrc_order + 1) * F + 100 * (1.7 if cfg.oqpsk else 1.0))


def bound_block(cfg, tag: str, T: int, L: int, kernel_ms: float) -> dict:
    """What binds the kernel of one BASELINE configuration and where its ceiling is, for the JSON line.

    Counters cannot be collected inside a timed run: traffic and the instruction mix are READ from the tracked rocprofv3 profile of
    this exact command (profiles/hbm_traffic.json, written by tools/make_profile_md.py), keyed by configuration and shape; the
    kernel time is this run's.  Returns {"traffic", "traffic_source", "valu": {...}} with

      frac_of_measured       achieved / the rate 1024 SIMDs at 2.4 GHz could issue THIS kernel's VALU mix back to back
      ceiling_hbm_frac       the HBM fraction with the pipe 100 % busy AND the FIR at its instruction floor - 2 packed instructions
                             per tap and firing (v_pk_mul_f32 + v_pk_add_f32: filter.c:55-62's unfused complex x real MAC on both
                             rails), no window padding - everything else as measured.  The number to read `frac` against."""
    bytes_per_sample = cfg.bps / 4 + 2 * cfg.symrate / cfg.samplerate
    achieved = T * L * bytes_per_sample / (kernel_ms * 1e-3) / 1e9
    hbm_frac = achieved / HBM_PEAK_GBS
    flops = flops_per_sample_of(cfg)
    valu_tops = flops * (T * L) / (kernel_ms * 1e-3) / 1e12
    rec = {}
    tfile = ROOT / "profiles" / "hbm_traffic.json"
    if tfile.exists():
        try:
            rec = json.loads(tfile.read_text()).get(f"{tag}:{T}x{L}", {})
        except Exception:
            rec = {}
    traffic = rec.get("hbm_bytes_per_launch")
    prof_round = rec.get("round")
    valu_per_firing, simd_busy = rec.get("valu_per_wave_firing"), rec.get("simd_valu_busy_frac")
    pipe_cycles, mean_cost, samples_per_wf = rec.get("valu_pipe_cycles_per_wave_firing"), rec.get("valu_mean_simd_cycles_per_instruction"), rec.get("samples_per_wave_firing")
    peak_measured = flops * samples_per_wf * (1024 * 2.4e9 / pipe_cycles) / 1e12 if pipe_cycles and samples_per_wf else None
    valu = {"achieved_top_s": round(valu_tops, 2), "peak_top_s": VALU_PEAK_TOPS, "frac": round(valu_tops / VALU_PEAK_TOPS, 4),
            "peak_measured_top_s": round(peak_measured, 2) if peak_measured else None,
            "frac_of_measured": round(valu_tops / peak_measured, 4) if peak_measured else None,
            "algorithmic_unfused_flops_per_sample": round(flops, 1),
            "valu_instructions_per_wave_firing": valu_per_firing,
            # share of a SIMD's 4-cycle issue quanta that carry a VALU instruction (2 waves x SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES)
            "simd_valu_busy_frac": simd_busy,
            "valu_instructions_source": f"profiles/hbm_traffic.json (round {prof_round}, SQ_INSTS_VALU)" if valu_per_firing else None}
    mix = rec.get("valu_mix_static_main_loop") or {}
    fir = mix.get("packed f32 (FIR taps)")
    if fir and pipe_cycles and simd_busy:
        taps = 2 * cfg.rrc_order + 1
        fir_floor = 2 * taps                                   # per firing: one packed multiply and one packed add per tap
        fir_static, fir_cycles = fir
        # what a wave-firing really issues (the wave-agreed padding skips jump over half-chunks): measured in round 6 with two --pmc passes
        # (profiles/r06_fir_padding.json, tools/fir_dynamic.py) for the std-window kernels; the static count of the code elsewhere
        fir_dyn = None
        try:
            fir_dyn = json.loads((ROOT / "profiles" / "r06_fir_padding.json").read_text()).get(tag, {}).get("fir_packed_dynamic")
        except Exception:
            pass
        fir_now = fir_dyn or fir_static
        floor_cycles = pipe_cycles - max(0.0, fir_now - fir_floor) * (fir_cycles / fir_static)
        valu["fir_packed_instructions_per_firing"] = {"static": fir_static, **({"dynamic": fir_dyn} if fir_dyn else {}), "floor": fir_floor}
        valu["pipe_busy_ceiling_hbm_frac"] = round(hbm_frac / simd_busy, 4)
        valu["ceiling_hbm_frac"] = round(hbm_frac * (pipe_cycles / simd_busy) / floor_cycles, 4)
        valu["ceiling_definition"] = ("pipe 100 % busy and the FIR at 2 packed instructions per tap and firing (no window padding: not reachable, the lanes of a wave "
                                      "sit ~3 slots apart and share one window), the other "
                                      f"{round(valu_per_firing - fir_now, 1)} VALU instructions per wave-firing as measured: {round(floor_cycles)} SIMD cycles of VALU pipe per "
                                      f"wave-firing against the {round(pipe_cycles / simd_busy)} one takes now; scaled from THIS run's kernel time, so it moves "
                                      "with the box's sustained clock like `frac` does (250-269 GS/s across boxes: 0.183-0.197)")
        conv = mix.get("conversion (SDWA)")
        if conv and conv[0] > 100:
            valu["conversions_note"] = (f"{conv[0]:.0f} of the instructions are v_cvt_f32_i32_sdwa: the packed s16 window is converted at use, every sample "
                                        f"{conv[0] / (2.0 * cfg.samplerate / cfg.symrate):.1f}x; converting once needs the window as floats (320 registers, one wave per "
                                        "SIMD: measured slower, and float windows are what the hybrid kernels are) - this is the floor for a packed window, "
                                        "so the ceiling above keeps them")
    return {"bytes_per_sample": bytes_per_sample, "achieved": achieved, "hbm_frac": hbm_frac, "traffic": traffic, "round": prof_round,
            "traffic_ratio": round(traffic / (T * L * bytes_per_sample), 3) if traffic else None,
            "valu_tops": valu_tops, "valu": valu, "valu_per_firing": valu_per_firing, "mean_cost": mean_cost, "peak_measured": peak_measured}



def parse() -> argparse.Namespace:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--tiles", type=int, default=393216,
                    help="tiles (= streams = lanes) per GPU; default = 2 residency rounds of the register-window kernels "
                         "(256 CUs x 12 waves x 64 lanes = 196608 lanes resident)")
    ap.add_argument("--tile-samples", type=int, default=16448,
                    help="IQ samples per tile.  NOT a power of two: lane l reads at base + l*tile_bytes, and a 64 KiB stride "
                         "puts all 64 lanes of a wave on the same L2 channel/sets (measured: 3.4x over-fetch, -3 %% throughput)")
    ap.add_argument("--config", default="c1", choices=["c1", "c3", "c4"], help="c1 = the headline config")
    ap.add_argument("--fanin", action="store_true", help="(the default at N > 1 since round 5: kept so that older command lines still parse)")
    ap.add_argument("--no-fanin", action="store_true",
                    help="N > 1: skip the gather of the soft symbols on rank 0 (RCCL over xGMI, after the timed region) and its `fanin` report")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="dry run of the N > 1 path on ONE GPU: every rank on device 0, gloo for the barrier / the reductions / the "
                         "fan-in (rows staged through host memory).  Exercises the whole world > 1 control flow where only one GPU can be "
                         "reached; what it cannot show is the RCCL transport itself")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    return ap.parse_args()


def demod_config(tag: str):
    from meteor_demod_amd import DemodConfig
    if tag == "c1":
        return DemodConfig(samplerate=230000), "configs[1]: QPSK 72k, 230 kS/s s16, RRC order 32, oversamp 5"
    if tag == "c3":
        return DemodConfig(samplerate=230000, symrate=80000, oqpsk=True), "configs[2]: OQPSK 80k, 230 kS/s s16"
    return (DemodConfig(samplerate=1000000, rrc_order=64, interp_factor=8),
            "configs[3]: QPSK 72k, 1 MS/s s16, RRC order 64, oversamp 8")


def cpu_baseline(cfg) -> dict:
    """Reference CPU path on this host: one single-threaded process per core (the reference has
    exactly one demod thread, main.c:218), each on the same 2^24-sample recording (about 10 s of CPU
    work per build: the contract's bounded sample)."""
    sys.path.insert(0, str(ROOT / "tests"))
    import numpy as np
    import oracle_py as O
    from meteor_demod_amd import synth

    n = 1 << 24
    st = synth.make_stream(424242, cfg.samplerate, cfg.symrate, oqpsk=cfg.oqpsk, f0_hz=1200.0)
    try:
        iq = synth.generate_device([st], n).cpu().numpy()[0]
    except Exception:
        iq = synth.generate_host(st, n)
    host = host_cpu_facts()
    # one single-threaded process per PHYSICAL core this process may run on, and no more than the cgroup's CPU quota
    # lets run at once: 256 processes under a 16-CPU quota measure the throttle, not the cores (r02: 2.06 MS/s per "core")
    procs = max(1, min(host["physical_cores"] or len(os.sched_getaffinity(0)), host["quota_cpus"] or 1 << 30))
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as td:
        path = Path(td) / "in.raw"
        iq.tofile(path)
        if O.have_ref():
            kind = "reference"

            def run_all(harness):
                cmd = [str(harness), "time", *O._ref_args(cfg), str(path)]
                t0 = time.time()
                ps = [subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for _ in range(procs)]
                outs = [p.communicate()[0] for p in ps]
                if any(p.returncode for p in ps):
                    return None, None               # e.g. SIGILL: built for a newer ISA than this host
                return [float(o.split()[0]) for o in outs], time.time() - t0

            # one process on the otherwise idle host first: what ONE core does with the reference when nothing competes
            host["one_process_idle_msps"] = one_core_msps(O.REF_HARNESS, O._ref_args(cfg), path, n)
            per, wall = run_all(O.REF_HARNESS)
            builds = {"strict -O2 -ffp-contract=off (the parity build)": round(procs * n / max(per) / 1e6, 2)}
            if O.REF_HARNESS_SHIPPED.exists():
                per_s, wall_s = run_all(O.REF_HARNESS_SHIPPED)
                if per_s:
                    builds["as shipped -O3 -march=x86-64-v3 -ftree-vectorize (CMakeLists.txt:17-18; not bit-reproducible)"] = \
                        round(procs * n / max(per_s) / 1e6, 2)
                    if max(per_s) < max(per):
                        per, wall = per_s, wall_s
        else:
            kind = "port"
            code = ("import sys,time,numpy as np;sys.path.insert(0,%r);import oracle_py as O;"
                    "from meteor_demod_amd import DemodConfig;cfg=DemodConfig(**%r);"
                    "iq=np.fromfile(%r,dtype=np.int16).reshape(-1,2);t=time.time();O.oracle_demod(cfg,iq);print(time.time()-t)"
                    % (str(ROOT / "tests"), cfg.__dict__, str(path)))
            t0 = time.time()
            ps = [subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, text=True, cwd=str(ROOT))
                  for _ in range(procs)]
            outs = [p.communicate()[0] for p in ps]
            wall = time.time() - t0
            per = [float(o.split()[-1]) for o in outs]
    agg = procs * n / max(per) / 1e6
    extra = {"builds_msps": builds} if kind == "reference" else {}
    return {"value": round(agg, 2), "unit": "Msamples/s", "cores": procs, "nproc": os.cpu_count(), "kind": kind, **extra, "host": host,
            "per_core_msps": round(n / (sum(per) / len(per)) / 1e6, 2),
            "sample": f"{procs} one-thread processes x 2^24-sample {cfg.symrate // 1000}k recording ({procs * n / 1e6:.0f} M samples, {sum(per):.1f} s CPU, "
                      f"{wall:.1f} s wall); faster of the builds",
            "sample_note": "one process per physical core, capped by the cgroup CPU quota; value = the faster of the builds listed"}


def one_core_msps(harness, ref_args, path, n, repeats: int = 2):
    """One single-threaded process of the reference's own loop (oracle/_ref/ref_harness time) on the otherwise idle host: what ONE
    core does when nothing competes.  None when the harness cannot run here (missing, or built for a newer ISA: SIGILL)."""
    best = None
    for _ in range(repeats):
        r = subprocess.run([str(harness), "time", *ref_args, str(path)], capture_output=True, text=True)
        if r.returncode != 0 or not r.stdout.split():
            return None
        t = float(r.stdout.split()[0])
        best = t if best is None else min(best, t)
    return round(n / best / 1e6, 2)


def one_core_all_configs() -> dict:
    """The reference's own per-sample loop (oracle/_ref/ref_harness time: demod.c + dsp/*.c compiled from /root/reference) as ONE
    process on the idle host for every single-GPU configuration of BASELINE.json, strict and as-shipped builds: Msamples/s."""
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle_py as O
    from meteor_demod_amd import synth
    if not O.have_ref():
        return {"error": "oracle/_ref not built on this host"}
    res = {}
    n = 1 << 22
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as td:
        for tag in ("c1", "c3", "c4"):
            cfg, workload = demod_config(tag)
            st = synth.make_stream(424242, cfg.samplerate, cfg.symrate, oqpsk=cfg.oqpsk, f0_hz=1200.0, rms=2000.0 if tag == "c4" else 6000.0)
            path = Path(td) / f"{tag}.raw"
            synth.generate_host(st, n).tofile(path)
            res[workload.split(":")[0]] = {"strict_msps": one_core_msps(O.REF_HARNESS, O._ref_args(cfg), path, n),
                                           "as_shipped_msps": one_core_msps(O.REF_HARNESS_SHIPPED, O._ref_args(cfg), path, n) if O.REF_HARNESS_SHIPPED.exists() else None}
    res["sample"] = f"2^22 samples per configuration, one process at a time, best of 2"
    return res


def cli_wall_times(local: int) -> dict:
    """One 2^26-sample configs[1] WAV file end to end (process start, file read, demodulation, .s written), wall seconds: this
    repository's C host in exact mode (one wavefront: the reference's own bytes) and with --tiled, and the reference's own binary
    (oracle/_ref/meteor_demod_ref = /root/reference's main.c and all, strict build) on one host core."""
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle_py as O
    from golden_cases import wav_header
    from meteor_demod_amd import synth
    n = 1 << 26
    st = synth.make_stream(1000, 230000, 72000, f0_hz=1200.0, clock_ppm=-3.5)
    iq = synth.generate_device([st], n, device=local)[0].cpu().numpy()
    cli = ROOT / "meteor_demod_amd" / "lib" / "meteor_demod_amd"
    res = {"file": f"2^26 samples s16 WAV ({n * 4 / 1e6:.0f} MB in /dev/shm), configs[1]"}
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as td:
        wav = Path(td) / "rec.wav"
        with open(wav, "wb") as f:
            f.write(wav_header(230000, 16, iq.nbytes))
            f.write(iq.tobytes())
        del iq
        outs = {}

        def run(name, cmd, repeat=1):
            best = None
            for _ in range(repeat):
                t0 = time.perf_counter()
                r = subprocess.run(cmd, capture_output=True, text=True)
                dt = time.perf_counter() - t0
                if r.returncode:
                    res[name] = {"error": (r.stderr or r.stdout)[-200:]}
                    return
                best = dt if best is None else min(best, dt)
            o = Path(cmd[cmd.index("-o") + 1])
            outs[name] = o.read_bytes()
            res[name] = {"seconds": round(best, 3), "msamples_per_s": round(n / best / 1e6, 1), "output_bytes": len(outs[name])}

        run("this_host_tiled", [str(cli), "-q", "--tiled", "-o", str(Path(td) / "t.s"), str(wav)], repeat=2)
        run("this_host_exact", [str(cli), "-q", "-B", "-o", str(Path(td) / "e.s"), str(wav)])
        if O.REF_BINARY.exists():
            run("reference_binary_one_core", [str(O.REF_BINARY), "-q", "-B", "-o", str(Path(td) / "r.s"), str(wav)])
        if "this_host_exact" in outs and "reference_binary_one_core" in outs:
            res["exact_output_equals_the_reference_binary"] = outs["this_host_exact"] == outs["reference_binary_one_core"]
    return res


def host_cpu_facts() -> dict:
    """What `cores` means on this host: logical CPUs in the affinity mask, distinct physical cores behind them (sysfs topology),
    the cgroup CPU quota if there is one.  256 busy processes on 128 physical cores with SMT are not 256 cores."""
    cpus = sorted(os.sched_getaffinity(0))
    phys = set()
    for c in cpus:
        try:
            base = Path(f"/sys/devices/system/cpu/cpu{c}/topology")
            phys.add(((base / "physical_package_id").read_text().strip(), (base / "core_id").read_text().strip()))
        except OSError:
            pass
    quota = None
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            quota = Path(f).read_text().strip()
            break
        except OSError:
            continue
    model = None
    try:
        model = next(l.split(":", 1)[1].strip() for l in Path("/proc/cpuinfo").read_text().splitlines() if l.startswith("model name"))
    except (OSError, StopIteration):
        pass
    quota_cpus = None
    try:
        a, b = quota.split()[:2] if quota and " " in quota else (quota, "100000")
        if a not in (None, "max", "-1"):
            quota_cpus = max(1, int(int(a) / int(b)))
    except (ValueError, AttributeError):
        pass
    return {"logical_cpus_in_affinity_mask": len(cpus), "physical_cores": len(phys) or None, "cgroup_cpu_max": quota, "quota_cpus": quota_cpus,
            "cpu_model": model}


def single_recording(cfg, iq, check: bool = True, serial=None, label_unchecked: bool = False, stream=None, twins_prefix: int = 0) -> dict:
    """The north star's overlapped tiling of ONE recording (NOTEBOOK.md 3.1): end-to-end latency of
    mdemod_demodulate_recording on the device tensor `iq` [n, 2] and, with `check`, agreement with the untiled serial
    oracle (symbol count, hard decisions, +-1 LSB, the exact prefix byte for byte).  Part of the CPU leg when checked."""
    sys.path.insert(0, str(ROOT / "tests"))
    import torch
    from meteor_demod_amd.recording import agreement, demodulate_recording_native
    n = int(iq.shape[0])
    demodulate_recording_native(cfg, iq[: 1 << 21])                 # warm-up (allocations, kernel images)
    torch.cuda.synchronize()
    t0 = time.time()
    soft, rep = demodulate_recording_native(cfg, iq)
    torch.cuda.synchronize()
    dt = time.time() - t0
    out = {"samples": n, "seconds": round(dt, 4), "msamples_per_s": round(n / dt / 1e6, 1),
           "pilot_seconds": round(rep.pilot_seconds, 4), "tiles_seconds": round(rep.tiles_seconds, 4),
           "pilot_samples": int(rep.pilot_samples), "pilot_msamples_per_s": round(rep.pilot_samples / max(rep.pilot_seconds, 1e-9) / 1e6, 2),
           "tiles": int(rep.n_tiles), "tile_samples": int(rep.tile_samples),
           "work_per_sample": round(rep.samples_demodulated / n, 2), "symbols": int(rep.n_symbols),
           "seam_fixes": int(rep.seam_fixes), "weak_seams": int(rep.weak_seams), "frame_misses": int(rep.frame_misses),
           "repaired_tiles": int(rep.repaired_tiles), "rotation_jumps": int(rep.rotation_jumps), "pilot_locked": int(rep.pilot_locked),
           "weak_carrier_tiles": int(rep.weak_carrier_tiles), "weak_clock_tiles": int(rep.weak_clock_tiles),
           "dead_reckoning_residual_rms_rad": round(float(rep.frame_residual_rms), 3)}
    if twins_prefix:
        # the yardstick that sees the same signal (round 5): 23 converged twins of the serial run - bit-exact streams of the library,
        # perturbed at instants spread over the first `twins_prefix` samples - window by window next to the tiled output
        from meteor_demod_amd.recording import tiled_vs_twins
        out["vs_twins_same_windows"] = tiled_vs_twins(cfg, iq[: twins_prefix].contiguous(), soft, int(rep.exact_symbols), copies=23)
        out["vs_twins_same_windows"]["samples"] = int(min(twins_prefix, n))
    if stream is not None:
        # EVERY symbol of the output against the symbols the generator transmitted (a device kernel regenerates them: milliseconds
        # for 2 G symbols, no serial run needed): a rotation jump or a cycle slip anywhere in the recording shows as a pairing change
        from meteor_demod_amd import synth
        t = synth.truth_check(stream, soft.contiguous(), first_symbol=int(max(rep.first_lock_symbol, 0)) + 20000, device=iq.device.index or 0)
        t["rail_error_rate"] = None if t["rail_error_rate"] is None else float(f"{t['rail_error_rate']:.3e}")
        t["note"] = ("hard decisions of ALL symbols from 20 000 after the first lock on, against the TRANSMITTED symbols; Es/N0 = 12 dB gives "
                     "a rail error rate of Q(sqrt(Es/N0)) = 3.4e-5 to any demodulator; pairing_changes = rotation jumps or symbol slips")
        out["truth"] = t
    if check or serial is not None:
        import numpy as np
        import oracle_py as O
        t_cpu = None
        if serial is None:
            t0 = time.time()
            serial = O.oracle_demod(cfg, iq.cpu().numpy())[0]
            t_cpu = time.time() - t0
            compared = "the whole recording"
        else:
            compared = (f"the first {len(serial)} symbols (the serial run of the first samples) for +-1 LSB; the rest only through `truth` "
                        "(every hard decision against the transmitted symbols)")
        m = len(serial)
        got = soft[: m + 64].cpu().numpy() if label_unchecked else soft.cpu().numpy()
        if label_unchecked:
            got = got[:m]                      # a serial run of a prefix IS the prefix of the serial run (the path is causal)
        a = agreement(got, serial)
        ex = min(int(rep.exact_symbols), m)
        # agreement by position in a tile's body: do tiles start badly?  (symbol -> input sample by proportion)
        mm = min(len(got), m)
        ok = np.abs(got[:mm].astype(np.int16) - serial[:mm].astype(np.int16)).max(axis=1) <= 1
        sps = int(rep.n_symbols) / n
        idx = np.arange(mm)
        pos = ((idx / sps - rep.pilot_samples) % rep.tile_samples) * sps
        tiled = idx >= int(rep.exact_symbols)
        head, rest = ok[tiled & (pos < 4096)], ok[tiled & (pos >= 4096)]
        out.update({"checked_against_serial_oracle": compared,
                    "symbols": [int(rep.n_symbols) if label_unchecked else a["len_stitched"], a["len_serial"]],
                    "exact_prefix_symbols": ex, "exact_prefix_bytes_equal": bool((got[:ex] == serial[:ex]).all()),
                    "within_1lsb": round(a["within_1lsb"], 5), "hard_decisions_equal": round(a["hard_decisions_equal"], 6),
                    "worst_window_4096": round(a["worst_window"], 4),
                    **{k: v for k, v in _window_stats(ok).items() if k in ("windows_below_0.99", "windows", "share_below_0.99", "window_p01")},
                    "within_1lsb_first_4096_of_a_tile_body": round(float(head.mean()), 5) if len(head) else None,
                    "within_1lsb_rest_of_the_tile_bodies": round(float(rest.mean()), 5) if len(rest) else None})
        if t_cpu is not None:
            out["serial_oracle_one_core_seconds"] = round(t_cpu, 2)
        out["_serial"] = serial
    elif label_unchecked:
        out["checked_against_serial_oracle"] = "no serial run; see `truth`"
    return out


def _window_stats(ok, W: int = 4096) -> dict:
    import numpy as np
    wins = np.array([float(ok[i:i + W].mean()) for i in range(0, len(ok) - W + 1, W)]) if len(ok) >= W else np.array([])
    return {"within_1lsb": round(float(ok.mean()), 5) if len(ok) else None, "worst_window_4096": round(float(wins.min()), 4) if len(wins) else None,
            "windows_below_0.99": int((wins < 0.99).sum()), "windows": int(len(wins)), "share_below_0.99": round(float((wins < 0.99).mean()), 5) if len(wins) else None,
            "window_p01": round(float(np.quantile(wins, 0.01)), 4) if len(wins) else None, "symbols_compared": int(len(ok))}


def perturbation_floor(cfg, iq) -> dict:
    """The yardstick for `within_1lsb` (oracle only, CPU leg), two ways.

    `one_lsb`: the reference against ITSELF with one input sample changed by 1 LSB.  The loops are chaotic at the ulp level (a
    symbol-clock word one ulp apart sustains a 3e-4 rad timing offset): ~0.2 % of the symbols are more than 1 LSB apart from then on -
    UNTIL the two runs meet again, which they do (every float of the state coincides by chance after 1e5..1e7 symbols) and are then
    identical for good.  A long comparison therefore mixes stretches at the floor with stretches of exact equality (r03's
    "floor 0.9992 / 0.9998" were that).
    `converged_pair_while_apart`: two CONVERGED runs of the reference on the same samples - the serial run, and the serial run's own
    state at a quarter of the recording with its symbol-clock word moved by 1 ppm (pulled in within a few loop time constants) -
    compared from 60 000 symbols after the perturbation up to the last symbol on which they differ.  That is what an independent
    demodulation of later samples (a tile) can expect against the serial run: a tile is emitted for ~2e4 symbols and has no time
    to meet it."""
    sys.path.insert(0, str(ROOT / "tests"))
    import numpy as np
    import oracle_py as O
    x = iq.cpu().numpy()
    K = len(x) // 4
    a = O.OracleStream(cfg)
    head = a.run(x[:K])[0]
    b = O.OracleStream(cfg)
    b.run(x[:K])
    b.state.t_freq = np.float32(b.state.t_freq * (1.0 + 1e-6))
    sa, sb = a.run(x[K:])[0], b.run(x[K:])[0]
    m = min(len(sa), len(sb))
    d = np.abs(sa[:m].astype(np.int16) - sb[:m].astype(np.int16)).max(axis=1)
    skip = 60000
    last = int(np.flatnonzero(d > 0)[-1]) + 1 if (d > 0).any() else 0
    apart = _window_stats(d[skip:last] <= 1) if last > skip + 4096 else {"within_1lsb": None}
    apart["met_again_after_symbols"] = last if last < m - 4096 else None
    y = x.copy()
    y[len(y) // 8, 0] += 1
    c = O.oracle_demod(cfg, y)[0]
    full = np.concatenate([head, sa])
    mm = min(len(full), len(c))
    dd = np.abs(full[:mm].astype(np.int16) - c[:mm].astype(np.int16)).max(axis=1)
    first = int(np.argmax(dd > 0))
    one = _window_stats(dd[first:] <= 1)
    lastc = int(np.flatnonzero(dd > 0)[-1]) + 1 if (dd > 0).any() else 0
    one["met_again_after_symbols"] = lastc - first if lastc < mm - 4096 else None
    return {"what": "the serial reference against itself: see bench.py perturbation_floor", "samples": int(len(x)),
            "within_1lsb": one["within_1lsb"], "worst_window_4096": one["worst_window_4096"],       # (r03's keys: the 1-LSB run, whole)
            "one_lsb": one, "converged_pair_while_apart": apart}


def recordings_leg(cfg_tag: str, buf, local: int, buf_stream=None) -> dict:
    """One long buffer as ONE recording on every BASELINE single-GPU configuration (2^26 samples each), then configs[1] end to
    end at SURVEY C2's 2^28 samples and on the whole bench buffer.  Everything that is timed is checked against the serial oracle
    (symbol count, hard decisions, +-1 LSB overall / at the start of a tile body / in the worst 4096-symbol window, the exact
    prefix byte for byte); of the whole buffer the first 2^28 samples' worth of symbols are, the rest is labelled UNCHECKED.  The
    reference's own 1-LSB-perturbation floor is measured for every configuration."""
    import torch
    from meteor_demod_amd import synth
    res = {}
    n = 1 << 26
    for tag in ("c1", "c3", "c4"):
        cfg, workload = demod_config(tag)
        if tag == cfg_tag and buf is not None and buf.shape[0] >= n:
            iq, rec = buf[:n], buf_stream
        else:
            # configs[3]: amplitude at which the reference's own AGC is stable (at 14 samples per symbol a 6000-LSB signal makes
            # gain += 1e-4 * (190 - |y|) overshoot through zero: the serial run itself is unlocked 42 % of the time)
            rec = synth.make_stream(2000, cfg.samplerate, cfg.symrate, oqpsk=cfg.oqpsk, f0_hz=1200.0, rms=2000.0 if tag == "c4" else 6000.0)
            iq = synth.generate_device([rec], n, device=local)[0]
        key = workload.split(":")[0]
        res[key] = single_recording(cfg, iq.contiguous(), stream=rec, twins_prefix=n)
        res[key].pop("_serial", None)
        res["perturbation_floor_" + key] = perturbation_floor(cfg, iq[: 1 << 25])
        del iq
        torch.cuda.empty_cache()
    res["configs[3] at SURVEY 8(d)'s 6000 LSB"] = c4_at_full_amplitude(local)
    if cfg_tag == "c1" and buf is not None:
        cfg, _ = demod_config("c1")
        serial28 = None
        if buf.shape[0] >= (1 << 28):
            r = single_recording(cfg, buf[: 1 << 28], stream=buf_stream, twins_prefix=1 << 27)
            serial28 = r.pop("_serial", None)
            res["configs[1] 2^28 samples"] = r
        if buf.shape[0] > (1 << 28) and serial28 is not None:
            # the last symbols of the prefix run depend on samples past 2^28 only through nothing (the path is causal), but keep
            # clear of the very end
            r = single_recording(cfg, buf, check=False, serial=serial28[: len(serial28) - 64], label_unchecked=True, stream=buf_stream)
            r.pop("_serial", None)
            res["configs[1] whole buffer"] = r
        elif buf.shape[0] > (1 << 28):
            res["configs[1] whole buffer"] = single_recording(cfg, buf, check=False, label_unchecked=True, stream=buf_stream)
    return res


def host_fed(local: int) -> dict:
    """"report host-fed throughput separately".  mdemod_process_host (what the C host calls: host buffers in, host
    buffers out, PCIe both ways, csrc/host_pipe.cpp) on 16 384 streams x 32 768 samples of configs[1] - never `value`.  The caller's
    buffers are allocated and touched beforehand (fresh pages would be timed as page faults), the call itself is timed; beside it
    the rate of the same bytes through plain pinned copies in and out (what the link gives with nothing else to do)."""
    import ctypes as C
    import numpy as np
    import torch
    from meteor_demod_amd import DemodConfig, Demodulator, synth
    cfg = DemodConfig(samplerate=230000)
    ns, n = 16384, 32768
    one = synth.generate_host(synth.make_stream(1, 230000, 72000, f0_hz=300.0), n)
    iq = np.empty((ns, n, 2), dtype=np.int16)
    iq[:] = one
    with Demodulator(cfg, ns, device=local) as d:
        cap = d.max_symbols(n)
        soft = np.zeros((ns, cap, 2), dtype=np.int8)
        iq_ptrs = (C.c_void_p * ns)(*[iq.ctypes.data + i * n * 4 for i in range(ns)])
        counts = (C.c_uint32 * ns)(*([n] * ns))
        soft_ptrs = (C.c_void_p * ns)(*[soft.ctypes.data + i * cap * 2 for i in range(ns)])
        caps = (C.c_uint32 * ns)(*([cap] * ns))
        produced = (C.c_uint32 * ns)()
        def timed_calls():
            times = []
            for _ in range(4):
                d.reset()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                rc = d._lib.mdemod_process_host(d._ctx, iq_ptrs, counts, soft_ptrs, caps, produced)
                times.append(time.perf_counter() - t0)
                if rc:
                    raise RuntimeError(f"mdemod_process_host: {rc}")
            return min(times[1:])
        # as any caller's buffers: staged through the library's pinned ring by the CPU ...
        dt_staged = timed_calls()
        first_rows = soft[:8].copy()
        # ... and as the C host (and INTEGRATION.md's binding) does it: the read buffer pinned once, rows copied where they are
        t0 = time.perf_counter()
        d.pin_host(iq)
        pin_ms = (time.perf_counter() - t0) * 1e3
        dt = timed_calls()
        same = bool(np.array_equal(first_rows, soft[:8]))
        out_bytes = 2 * int(sum(produced))
    # the link alone: the same bytes, pinned, one copy each way
    h_in = torch.empty(ns * n * 2, dtype=torch.int16).pin_memory()
    d_in = torch.empty_like(h_in, device=f"cuda:{local}")
    h_out = torch.empty(out_bytes, dtype=torch.int8).pin_memory()
    d_out = torch.empty_like(h_out, device=f"cuda:{local}")
    # (round 5: the two copies on streams of their own, as the pipeline issues them - r04 queued both on one stream, so the copy out
    #  waited for the copy in and the "link" read 49.8 GB/s where it gives 57)
    link = []
    s_in, s_out = torch.cuda.Stream(device=local), torch.cuda.Stream(device=local)
    for _ in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(s_in):
            d_in.copy_(h_in, non_blocking=True)
        with torch.cuda.stream(s_out):
            h_out.copy_(d_out, non_blocking=True)
        torch.cuda.synchronize()
        link.append(time.perf_counter() - t0)
    link = link[1:]
    in_bytes = ns * n * 4
    return {"workload": f"{ns} streams x {n} samples of configs[1] through mdemod_process_host (host buffers both ways)",
            "seconds": round(dt, 4), "msamples_per_s": round(ns * n / dt / 1e6, 1),
            "gbytes_per_s_in": round(in_bytes / dt / 1e9, 1), "gbytes_per_s_both_ways": round((in_bytes + out_bytes) / dt / 1e9, 1),
            "input": "the caller's read buffer pinned once with mdemod_pin_host_buffer (as host/meteor_demod_amd.c does), rows copied from where they are",
            "pin_ms_once": round(pin_ms, 1), "same_bytes_as_staged": same,
            "staged": {"seconds": round(dt_staged, 4), "gbytes_per_s_in": round(in_bytes / dt_staged / 1e9, 1),
                       "what": "the same call on buffers the library was not told about: packed into its own pinned ring by 12 host threads"},
            "input_bytes": in_bytes, "output_bytes": out_bytes,
            "link_alone_gbytes_per_s_in": round(in_bytes / min(link) / 1e9, 1),
            "frac_of_link": round(min(link) / dt, 3),
            "note": "PCIe-inclusive; `value` above is device-resident.  link_alone = the same bytes as one pinned hipMemcpyAsync each way, "
                    "the two on streams of their own (concurrent, as the pipeline issues them)"}


def c4_at_full_amplitude(local: int) -> dict:
    """configs[3] (1 MS/s, -f 64 -O 8) as ONE recording at the 6 000 LSB rms SURVEY 8(d) specifies - the amplitude the other legs
    avoid (rms 2000), because there the REFERENCE does not demodulate: at 14 samples per symbol its filter output is so large that
    gain += 1e-4 * (190 - |y|) (agc.c:13-25) overshoots through zero every few symbols.  Reported: how long the serial run is
    locked at all, how many of ITS hard decisions are the transmitted symbols, the same for the tiled run, and their agreement."""
    sys.path.insert(0, str(ROOT / "tests"))
    import numpy as np
    import torch
    import oracle_py as O
    from meteor_demod_amd import synth
    from meteor_demod_amd.recording import agreement, demodulate_recording_native
    cfg, _ = demod_config("c4")
    out = {}
    for rms in (6000.0, 3000.0):
        st = synth.make_stream(2000, cfg.samplerate, cfg.symrate, f0_hz=1200.0, rms=rms)
        n = 1 << 25
        iq = synth.generate_device([st], n, device=local)[0].contiguous()
        soft, rep = demodulate_recording_native(cfg, iq, device=local)
        serial, trace, ev = O.oracle_demod(cfg, iq.cpu().numpy(), True)
        a = agreement(soft.cpu().numpy(), serial)
        m0, cnt = min(len(serial), int(rep.n_symbols)) // 2, 262144
        ser_dev = torch.from_numpy(serial).to(soft.device).contiguous()
        out[f"rms {int(rms)}"] = {
            "serial_run_locked_fraction": round(float(trace["locked"].mean()), 4), "serial_run_lock_events": len(ev),
            "serial_decisions_that_are_the_transmitted_symbols": round(synth.best_pairing_agreement(st, ser_dev, m0, cnt, device=local), 4),
            "tiled_decisions_that_are_the_transmitted_symbols": round(synth.best_pairing_agreement(st, soft.contiguous(), m0, cnt, device=local), 4),
            "tiled_vs_serial": {"symbols": [a["len_stitched"], a["len_serial"]], "within_1lsb": round(a["within_1lsb"], 4),
                                "hard_decisions_equal": round(a["hard_decisions_equal"], 4)},
            "pilot_locked": int(rep.pilot_locked), "rotation_jumps": int(rep.rotation_jumps)}
        del iq, soft, ser_dev
        torch.cuda.empty_cache()
    out["note"] = ("0.5 = no relation to the signal.  At 6000 LSB the reference's own output is not a demodulation of the signal (its AGC is unstable at "
                   "14 samples per symbol), so there is nothing for a tiled run to agree with; at 3000 LSB both runs make the same decisions")
    return out


def other_configs(skip: str, T: int, L: int, local: int) -> dict:
    """Untimed extra (N=1 only): the other single-GPU configurations of BASELINE.json on the same tiling, 3 passes each,
    so that the round's BENCH file also carries the OQPSK and the 1 MS/s / 129-tap numbers.  Not `value`."""
    import torch
    from meteor_demod_amd import Demodulator, synth
    res = {}
    from meteor_demod_amd import DemodConfig
    extra = {"x1": (DemodConfig(samplerate=1024000), "not in BASELINE.json: QPSK 72k, 1.024 MS/s s16, default RRC order 32, oversamp 5"),
             "x2": (DemodConfig(samplerate=1800000), "not in BASELINE.json: QPSK 72k, 1.8 MS/s s16, default RRC order 32, oversamp 5"),
             "x3": (DemodConfig(samplerate=1024000, bps=32), "not in BASELINE.json: QPSK 72k, 1.024 MS/s f32, default RRC order 32, oversamp 5"),
             "x4": (DemodConfig(samplerate=1000000, rrc_order=64, interp_factor=8, bps=32),
                    "not in BASELINE.json: QPSK 72k, 1 MS/s f32, RRC order 64, oversamp 8 (v3 hybrid window since r03; v1 ring kernel before)"),
             "x5": (DemodConfig(samplerate=2048000, bps=32), "not in BASELINE.json: QPSK 72k, 2.048 MS/s f32, default RRC order 32, oversamp 5"),
             "x6": (DemodConfig(samplerate=3200000), "not in BASELINE.json: QPSK 72k, 3.2 MS/s s16 (an RTL-SDR's top rate), default RRC order 32, oversamp 5"),
             "x7": (DemodConfig(samplerate=10000000), "not in BASELINE.json: QPSK 72k, 10 MS/s s16 (an Airspy's), default RRC order 32, oversamp 5: gather geometry, "
                                                      "which loads 68 of a symbol's 139 samples - hbm_frac counts them all, as SURVEY 8(d) does")}
    for tag in ("c3", "c4", "x1", "x2", "x3", "x4", "x5", "x6", "x7"):
        if tag == skip:
            continue
        cfg, workload = extra[tag] if tag in extra else demod_config(tag)
        Tc = T // 2 if cfg.bps == 32 else T        # 8 bytes per sample: half the tiles
        rec = synth.make_stream(2000, cfg.samplerate, cfg.symrate, oqpsk=cfg.oqpsk, f0_hz=1200.0, fmt=cfg.bps,
                                **(dict(rms=0.25, dc=(0.001, -0.002)) if cfg.bps == 32 else {}))
        buf = torch.empty((Tc * L, 2), dtype=torch.float32 if cfg.bps == 32 else torch.int16, device=f"cuda:{local}")
        synth.generate_device([rec], Tc * L, out=buf.view(1, Tc * L, 2), device=local)
        x = buf.view(Tc, L, 2)
        with Demodulator(cfg, Tc, device=local) as d:
            soft = torch.empty((Tc, d.max_symbols(L), 2), dtype=torch.int8, device=f"cuda:{local}")
            d.process(x, soft=soft)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(3):
                d.process(x, soft=soft)
            b.record()
            torch.cuda.synchronize()
            ms = a.elapsed_time(b) / 3
            bps = cfg.bps / 4 + 2 * cfg.symrate / cfg.samplerate
            entry = {"msamples_per_s": round(Tc * L / ms / 1e3, 1), "kernel_ms": round(ms, 3), "tiles": Tc,
                     "hbm_frac": round(Tc * L * bps / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "kernel": d.kernel_name}
            if tag in ("c1", "c3", "c4"):
                # the BASELINE configurations carry what binds them and where their ceiling is, like the headline (profiles/hbm_traffic.json)
                Bc = bound_block(cfg, tag, Tc, L, ms)
                entry.update({"bound": "valu" if Bc["valu_tops"] / VALU_PEAK_TOPS > Bc["hbm_frac"] else "hbm", "traffic": Bc["traffic"],
                              "traffic_over_algorithmic": Bc["traffic_ratio"], "profile_round": Bc["round"], "valu": Bc["valu"]})
            res[workload.split(";")[0]] = entry
        del buf, x, soft
        torch.cuda.empty_cache()
    return res


def spot_check(cfg, d, x, tiles, L, n_check=12) -> str:
    """Untimed: reset, one pass, compare sampled tiles (always including the first and the last,
    i.e. blocks of the first and of the last residency round) byte-for-byte with the oracle."""
    sys.path.insert(0, str(ROOT / "tests"))
    import numpy as np
    import torch
    import oracle_py as O
    d.reset()
    soft = d.process(x)
    torch.cuda.synchronize()
    picks = {0, tiles - 1, tiles // 2} | set(int(t) for t in np.random.default_rng(0).choice(tiles, n_check, replace=False))
    for t in sorted(picks):
        st = d.status(int(t), 1)[0]
        want = O.oracle_demod(cfg, x[int(t)].cpu().numpy())[0]
        got = soft[int(t), : st.symbols_this_call].cpu().numpy()
        if got.shape != want.shape or not np.array_equal(got, want):
            return f"MISMATCH tile {int(t)}"
    return f"{len(picks)} sampled tiles byte-identical to oracle"


def spawn_ranks(args) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks (one process per GPU, RCCL rendezvous on 127.0.0.1) through
    torch.distributed.run and hand its exit code back.  Runs BEFORE anything touches the GPU in this process
    (torch.cuda.device_count() does not initialise it)."""
    import socket
    import torch
    have = torch.cuda.device_count()
    if args.oversubscribe:
        have = args.gpus if have >= 1 else 0        # every rank on device 0
    if have < args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but this node exposes {have} GPU(s); refusing to print an N={have} line "
                         f"labelled as {args.gpus}\n")
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


LINE_TARGET_BYTES = 4096      # the driver keeps the last ~8 KB of stdout (r05's 23 KB line was cut: BENCH_r05.parsed = null)
LINE_HARD_BYTES = 6000        # tests/test_host_logic.py and tests/test_gpu_multi.py hold the line under this
LINE_MAX_STRING = 160
EXTRAS_FILE = ROOT / "bench_extras.json"


def _clip(s, n: int = LINE_MAX_STRING):
    return s if not isinstance(s, str) or len(s) <= n else s[: n - 3] + "..."


def _pick(d, *keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def compact_line(full: dict) -> dict:
    """The ONE stdout line, from the full record: the contract's keys first, then the few figures a reader needs beside `value`
    (VERDICT r05 item 1).  Everything else - `single_recording`, `other_configs`, `cli_wall_times`, `host_fed`, notes and
    definitions - stays in the full record, which goes to bench_extras.json next to this script and to stderr.  Pure function of
    `full` (tests feed it tracked records)."""
    out = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                    "vs_baseline", "dtype", "data")}
    cfgd = full.get("config", {})
    out["config"] = {"workload": _clip(cfgd.get("workload")), **_pick(cfgd, "tiles_per_gpu", "tile_samples", "samples_per_step")}
    r = full.get("roofline", {})
    v = r.get("valu") or {}
    fir = v.get("fir_packed_instructions_per_firing")
    out["roofline"] = {"bound": r.get("bound"), **_pick(r, "limited_by"), "achieved": r.get("achieved"), "peak": r.get("peak"), "unit": r.get("unit"),
                       "frac": r.get("frac"), "traffic": r.get("traffic"), "traffic_over_algorithmic": r.get("traffic_over_algorithmic"),
                       "kernel": _clip(r.get("kernel"), 80), "kernel_ms": r.get("kernel_ms"), **_pick(r, "kernel_ms_over_ranks"),
                       "algorithmic_bytes_per_sample": r.get("algorithmic_bytes_per_sample"),
                       "ceiling_hbm_frac": v.get("ceiling_hbm_frac"), **_pick(r, "profile_round"),
                       # (scalars at this level too: a reader that keeps only the flat keys of `roofline` still sees what binds)
                       "simd_valu_busy_frac": v.get("simd_valu_busy_frac"), "valu_per_wave_firing": v.get("valu_instructions_per_wave_firing"),
                       **({"fir_packed_dynamic_per_firing": fir.get("dynamic"), "fir_packed_floor_per_firing": fir.get("floor")} if isinstance(fir, dict) and fir.get("dynamic") else {}),
                       "valu": {**_pick(v, "frac", "frac_of_measured", "simd_valu_busy_frac", "valu_instructions_per_wave_firing"),
                                **({"fir_packed_per_firing": fir} if fir else {})}}
    c = full.get("cpu_baseline")
    if isinstance(c, dict):
        out["cpu_baseline"] = {**_pick(c, "value", "unit", "cores", "nproc", "kind", "per_core_msps"), "sample": _clip(c.get("sample"))}
        one = c.get("one_core_msps")
        if isinstance(one, dict) and "error" not in one:
            out["cpu_baseline"]["one_core_strict_msps"] = {k: e.get("strict_msps") for k, e in one.items() if isinstance(e, dict)}
    if full.get("check") is not None:
        out["check"] = _clip(full["check"])
    sr = full.get("single_recording")
    if isinstance(sr, dict):
        whole, c2 = sr.get("configs[1] whole buffer") or {}, sr.get("configs[1] 2^28 samples") or {}
        tw = c2.get("vs_twins_same_windows") or {}
        out["useful"] = {"what": "ONE recording in, ONE locked symbol stream out (overlapped tiles, +-1 LSB statistical): `value` is cold-start tiles",
                         "whole_buffer_msps": whole.get("msamples_per_s"), "whole_buffer_work_per_sample": whole.get("work_per_sample"),
                         "c2_2p28_msps": c2.get("msamples_per_s"), "c2_2p28_seconds": c2.get("seconds"),
                         "c2_within_1lsb_of_serial": c2.get("within_1lsb"), "c2_hard_decisions_equal": c2.get("hard_decisions_equal"),
                         **({"c2_twins_within_1lsb_same_windows": (tw.get("twins_same_windows") or {}).get("within_1lsb"),
                             "c2_worst_window_tiled_vs_twins": [(tw.get("tiled") or {}).get("worst_window"), (tw.get("twins_same_windows") or {}).get("worst_window")]}
                            if tw else {})}
        if "error" in sr:
            out["useful"]["error"] = _clip(sr["error"])
    oc = full.get("other_configs")
    if isinstance(oc, dict):
        out["other_configs"] = {k.split(":")[0]: _pick(e, "msamples_per_s", "hbm_frac", "kernel_ms") for k, e in oc.items()
                                if k.startswith("configs[") and isinstance(e, dict)}
    hf = full.get("host_fed")
    if isinstance(hf, dict):
        out["host_fed"] = _pick(hf, "gbytes_per_s_in", "link_alone_gbytes_per_s_in", "frac_of_link") or {"error": _clip(hf.get("error"))}
    cw = full.get("cli_wall_times")
    if isinstance(cw, dict):
        out["cli_2p26_wav_seconds"] = {"exact": (cw.get("this_host_exact") or {}).get("seconds"), "tiled": (cw.get("this_host_tiled") or {}).get("seconds"),
                                       "reference_one_core": (cw.get("reference_binary_one_core") or {}).get("seconds"),
                                       "exact_equals_reference": cw.get("exact_output_equals_the_reference_binary")}
    for k in ("rccl", "fanin"):
        if isinstance(full.get(k), dict):
            out[k] = {kk: (_clip(vv) if isinstance(vv, str) else vv) for kk, vv in full[k].items() if not kk.endswith("_means") and kk != "note"}
    if full.get("oversubscribed"):
        out["oversubscribed"] = _clip(full["oversubscribed"])
    if full.get("errors"):
        out["errors"] = {k: _clip(v) for k, v in full["errors"].items()}
    out["extras"] = full.get("extras", EXTRAS_FILE.name)
    return out


class Emitter:
    """Prints the line exactly once, whatever happens after the timed region: at the end of main(), or from
    the watchdog when the post-region work (fan-in, checker, extras) exceeds its deadline, or when a SIGTERM arrives (the launcher
    tearing the job down because ANOTHER rank died).  The timed result is never lost to code that runs after it."""

    def __init__(self, rank: int, full: dict, deadline_s: float):
        import signal
        import socket
        import threading
        self.rank, self.full, self.done = rank, full, False
        self.lock = threading.Lock()
        self.deadline = time.monotonic() + deadline_s
        self.stage = "start"
        threading.Thread(target=self._watchdog, daemon=True).start()
        # SIGTERM: the C-level handler writes the signal number to a socket at once, from whichever thread it lands on; a thread
        # of our own waits on the other end (the main thread may be inside a collective and never get back to the interpreter)
        try:
            self._rd, self._wr = socket.socketpair()
            self._wr.setblocking(False)
            signal.signal(signal.SIGTERM, lambda *_: None)
            signal.set_wakeup_fd(self._wr.fileno(), warn_on_full_buffer=False)
            threading.Thread(target=self._on_signal, daemon=True).start()
        except (ValueError, OSError):
            pass                                   # (not the main thread: no handler, the watchdog still stands)

    def _finish_from_thread(self, why: str):
        if self.rank == 0:
            self.full.setdefault("errors", {})["post_region"] = why
            self.emit()
        sys.stdout.flush()
        os._exit(0)

    def _watchdog(self):
        while time.monotonic() < self.deadline:
            time.sleep(0.25)
            if self.done:
                return
        if not self.done:
            self._finish_from_thread(f"work after the timed region exceeded its deadline in stage '{self.stage}': line printed by the watchdog, process ended")

    def _on_signal(self):
        import signal
        while True:
            b = self._rd.recv(1)
            if b and b[0] == signal.SIGTERM and not self.done:
                self._finish_from_thread(f"SIGTERM in stage '{self.stage}' (the launcher ending the job: another rank failed?): line printed by the signal thread")

    def emit(self):
        with self.lock:
            if self.done or self.rank != 0:
                self.done = True
                return
            # a snapshot: from the watchdog or the signal thread the main thread may still be adding to the record
            full = None
            for _ in range(20):
                try:
                    full = json.loads(json.dumps(self.full, default=str))
                    break
                except RuntimeError:                 # "dictionary changed size during iteration"
                    time.sleep(0.02)
            if full is None:
                full = {k: self.full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                                      "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "errors")}
            try:
                EXTRAS_FILE.write_text(json.dumps(full, indent=1, default=str))
                full["extras"] = EXTRAS_FILE.name + " (next to bench.py; also on stderr)"
            except OSError as e:
                full["extras"] = f"stderr only ({type(e).__name__}: bench_extras.json not writable)"
            sys.stderr.write("bench.py full record: " + json.dumps(full, default=str) + "\n")
            sys.stderr.flush()
            line = json.dumps(compact_line(full), default=str)
            if len(line) > LINE_HARD_BYTES:        # never again an unparseable record: drop the optional blocks, largest first
                c = compact_line(full)
                for k in sorted((k for k in c if k not in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                                          "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")), key=lambda k: -len(json.dumps(c[k], default=str))):
                    c.pop(k)
                    line = json.dumps(c, default=str)
                    if len(line) <= LINE_TARGET_BYTES:
                        break
            sys.stdout.write(line + "\n")
            sys.stdout.flush()
            self.done = True


def _fault(stage: str, rank: int) -> None:
    """Test-only fault injection (tests/test_gpu_multi.py): MDEMOD_BENCH_FAULT=<stage>_raise[@rank] | <stage>_hang[@rank]."""
    spec = os.environ.get("MDEMOD_BENCH_FAULT", "")
    if not spec:
        return
    what, _, who = spec.partition("@")
    if who and int(who) != rank:
        return
    if what == stage + "_raise":
        raise RuntimeError(f"injected fault in {stage} on rank {rank}")
    if what == stage + "_hang":
        time.sleep(3600)


def device_identity(local: int) -> str:
    """uuid + PCI address of the device a rank computes on (short: eight of them ride in the line)."""
    import torch
    p = torch.cuda.get_device_properties(local)
    parts = []
    try:
        parts.append(str(p.uuid))
    except Exception:
        parts.append(p.name)
    try:
        parts.append(f"pci {int(p.pci_bus_id):02x}:{int(p.pci_device_id):02x}")
    except Exception:
        pass
    return " ".join(parts)[:64]


def main() -> None:
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args))
    import torch
    from meteor_demod_amd import Demodulator, synth
    from meteor_demod_amd.sharding import fanin_bytes_on_the_wire, fanin_soft, init_from_env

    post_deadline = float(os.environ.get("MDEMOD_BENCH_POST_DEADLINE_S", "0") or 0)
    rank, local, world = init_from_env("gloo" if args.oversubscribe else None, timeout_s=300.0)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the demodulator has no CPU path")
    if args.oversubscribe:
        local = 0
    torch.cuda.set_device(local)
    dist = torch.distributed if world > 1 else None
    # device tensors for RCCL, host tensors for the oversubscribed dry run's gloo
    coll_dev = "cpu" if args.oversubscribe else f"cuda:{local}"

    cfg, workload = demod_config(args.config)
    T, L = args.tiles, args.tile_samples

    # ---- one long synthetic recording, generated on the device, cut into T tiles -------------
    rec = synth.make_stream(1000 * 1 + rank, cfg.samplerate, cfg.symrate, oqpsk=cfg.oqpsk, f0_hz=1200.0,
                            clock_ppm=7.0 * (rank - world / 2))
    buf = torch.empty((T * L, 2), dtype=torch.int16, device=f"cuda:{local}")
    synth.generate_device([rec], T * L, out=buf.view(1, T * L, 2), device=local)
    x = buf.view(T, L, 2)

    d = Demodulator(cfg, T, device=local)
    cap = d.max_symbols(L)
    soft = torch.empty((T, cap, 2), dtype=torch.int8, device=f"cuda:{local}")

    for _ in range(args.warmup):
        d.process(x, soft=soft)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()
        d.process(x, soft=soft)
        b.record()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = sum(a.elapsed_time(b) for a, b in ev) / max(1, len(ev))
    elapsed_here, kernel_ms_here = elapsed, kernel_ms

    # ---- the timed region is over: from here on nothing may lose it -------------------------------------------------------------
    # Rank 0 holds the record of THIS rank's clock already; the max over ranks replaces it if the reduction below succeeds.
    full: dict = {}
    em = Emitter(rank, full, post_deadline or (240.0 if world > 1 else 1500.0))
    errors: dict = {}

    def assemble(elapsed: float, kernel_ms: float, kernel_ms_ranks=None) -> None:
        samples_per_step = T * L * world
        value = samples_per_step * args.steps / elapsed / 1e6
        bytes_per_sample = cfg.bps / 4 + 2 * cfg.symrate / cfg.samplerate
        achieved = T * L * bytes_per_sample / (kernel_ms * 1e-3) / 1e9         # algorithmic bytes per launch, per GPU / kernel time
        B = bound_block(cfg, args.config, T, L, kernel_ms)
        if B["peak_measured"]:
            B["valu"]["peak_measured_note"] = ("the rate at which 1024 SIMDs at 2.4 GHz could issue THIS kernel's VALU mix back to back: "
                                               f"{B['valu_per_firing']} VALU instructions per wave-firing (rocprofv3 SQ_INSTS_VALU) x {B['mean_cost']} SIMD cycles each "
                                               "(mix-weighted, per-instruction costs measured at two waves per SIMD: tools/ubench/valu_mix.hip, "
                                               "tools/valu_cost.py); the 78.6 Top/s above is the data-sheet unfused-FP32 figure, which no mix of "
                                               "conversions, selects, f64 and packed instructions can reach")
        ceiling = B["valu"].get("ceiling_hbm_frac")
        ceiling_note = None
        if ceiling:
            ceiling_note = (f"BASELINE.json's >= 0.40 of HBM is out of reach for this configuration under the reference's bit-exact arithmetic: the kernel is "
                            f"bound by FP32 VALU issue, not by traffic ({B['traffic_ratio']}x the algorithmic bytes at {achieved / 1e3:.2f} of 8 TB/s); with the issue "
                            f"pipe 100 % busy and the FIR at its floor of 2 unfused packed instructions per tap it tops out at {ceiling} of HBM "
                            f"(now {round(achieved / HBM_PEAK_GBS, 4)} = {round(achieved / HBM_PEAK_GBS / ceiling * 100)} % of that ceiling)")
        valu_binds = B["valu_tops"] / VALU_PEAK_TOPS > achieved / HBM_PEAK_GBS
        full.update({
            "metric": "IQ Msamples/s demodulated (whole node)", "value": round(value, 1), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload + f"; {T} cold-start tiles x {L} samples/GPU, 1 per lane, unlocked, bit-exact per tile",
                       "workload_note": "one device-resident IQ buffer per GPU cut into independent tiles; every tile starts from power-on state, and at "
                                        "+1.2 kHz none locks within 16448 samples (lock needs ~80 k symbols): `value` is the rate of the reference's recurrence, "
                                        "`useful` the rate at which ONE recording becomes ONE locked symbol stream",
                       "tiles_per_gpu": T, "tile_samples": L, "samples_per_step": samples_per_step,
                       "input_bytes_per_gpu": T * L * 4},
            "roofline": {"bound": "hbm",
                         "limited_by": "fp32 valu issue" if valu_binds else "hbm",
                         "bound_note": "achieved/peak/frac are the HBM figures BASELINE.json asks for (the path has no MFMA work); the resource that binds is "
                                       "FP32 VALU issue (SURVEY H3: the reference's unfused arithmetic caps configs[1] at 0.37 of the HBM peak): see `valu`",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": B["traffic"],
                         "traffic_source": (f"profiles/hbm_traffic.json[{args.config}:{T}x{L}] (round {B['round']}): rocprofv3 FETCH_SIZE x2 + WRITE_SIZE of "
                                            "this command, separate --pmc passes; not measured by this run") if B["traffic"] else None,
                         "profile_round": B["round"],
                         "valu": B["valu"],
                         **({"ceiling_note": ceiling_note} if ceiling_note else {}),
                         "traffic_over_algorithmic": B["traffic_ratio"],
                         "kernel": d.kernel_name,
                         "kernel_ms": round(kernel_ms, 3),
                         **({"kernel_ms_over_ranks": kernel_ms_ranks} if kernel_ms_ranks else {}),
                         "algorithmic_bytes_per_sample": round(bytes_per_sample, 4)},
        })
        if errors:
            full["errors"] = errors

    assemble(elapsed, kernel_ms)
    if args.oversubscribe:
        full["oversubscribed"] = (f"DRY RUN: {world} ranks on ONE GPU (device 0), gloo instead of RCCL - the world > 1 control flow, not a "
                                  "scaling measurement; `value` is what one GPU gives when it is shared")

    def stage(name: str, fn):
        """One step of the post-region work: an exception is recorded under errors[name] and the run goes on."""
        em.stage = name
        try:
            _fault(name, rank)
            return fn()
        except BaseException as e:                                   # (SystemExit / KeyboardInterrupt included: the line still has to appear)
            errors[name] = f"{type(e).__name__}: {e}"[:300]
            full["errors"] = errors
            return None

    if dist:
        def reduce_times():
            tt = torch.tensor([elapsed_here, kernel_ms_here, -kernel_ms_here], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            assemble(float(tt[0]), float(tt[1]),
                     {"min": round(-float(tt[2]), 3), "max": round(float(tt[1]), 3), "this_rank": round(kernel_ms_here, 3)})   # a straggler GPU shows here
            return True
        if stage("reduce_times", reduce_times) is None and rank == 0:
            full["value_note"] = "rank 0's own clock: the max-over-ranks reduction failed (see errors)"

        def rccl_facts():
            # did the collective layer see every rank, and which device is each on?
            ones = torch.ones(1, dtype=torch.int32, device=coll_dev)
            dist.all_reduce(ones, op=dist.ReduceOp.SUM)
            ident = device_identity(local).encode()[:64].ljust(64, b" ")
            mine = torch.tensor(list(ident), dtype=torch.uint8, device=coll_dev)
            every = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(every, mine)
            devs = [bytes(t.cpu().tolist()).decode(errors="replace").strip() for t in every]
            ver = None
            try:
                ver = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception:
                pass
            full["rccl"] = {"backend": dist.get_backend(), "world": world, "ranks_seen": int(ones[0]), "device_name": torch.cuda.get_device_properties(local).name, "devices": devs,
                            "distinct_devices": len(set(devs)), "rccl_version": ver}
        stage("rccl", rccl_facts)

    if dist and not args.no_fanin:
        def fanin():
            # fan-in of the soft symbols to rank 0 over RCCL (outside the timed region): rows compacted on the device to the
            # nominal symbol pitch first (0.63 B per input sample instead of the hard-bound 2 B), then exact-size send/recv
            counts = torch.from_numpy(d.status_array()["symbols_this_call"].astype("int32")).to(coll_dev)
            pitch = d.nominal_pitch(L)
            torch.cuda.synchronize(); dist.barrier()
            t1 = time.perf_counter()
            packed = d.compact(soft, pitch)
            if args.oversubscribe:
                packed = packed.cpu()                    # gloo: the rows go through host memory
            got, got_counts = fanin_soft(packed, counts, T * world, dst=0)
            torch.cuda.synchronize(); dist.barrier()
            fanin_ms = (time.perf_counter() - t1) * 1e3
            fanin_bytes = fanin_bytes_on_the_wire([(T, pitch)] * world)
            full["fanin"] = {"ms": round(fanin_ms, 2), "rows_intact": None, "bytes_over_xgmi": fanin_bytes,
                             "gbytes_per_s": round(fanin_bytes / (fanin_ms * 1e-3) / 1e9, 1), "row_pitch_symbols": pitch,
                             "transport": "exact-size send/recv per rank (batch_isend_irecv), rows land in place on rank 0",
                             "rows_intact_means": "the rows of EVERY rank arrived on rank 0 with the hash their sender computed, and every row has its symbol count",
                             "note": "compact to nominal pitch + point-to-point to rank 0, outside the timed region"}
            # rows intact: every rank hashes the rows it sent (a 64-bit sum of its bytes weighted by position, on the device), rank 0
            # hashes what arrived for each rank, the sums travel by all_gather - no second copy of the soft symbols over the links
            def row_hash(t, chunk=1 << 24):
                flat, acc = t.reshape(-1), 0
                w = (torch.arange(chunk, device=flat.device, dtype=torch.int64) % 65521) + 1
                for at in range(0, flat.numel(), chunk):             # (wrapping int64 sums: the chunking does not change the value)
                    v = flat[at:at + chunk].to(torch.int64)
                    acc = (acc + int((v * w[: v.numel()] * (1 + (at // chunk) % 8191)).sum().item())) & 0xFFFFFFFFFFFFFFFF
                return acc - (1 << 64) if acc >= (1 << 63) else acc
            mine = torch.tensor([row_hash(packed)], dtype=torch.int64, device=coll_dev)
            sums = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(sums, mine)
            if rank == 0:
                per = int(packed.numel())
                flat = got.reshape(-1)
                arrived = [row_hash(flat[r * per:(r + 1) * per]) for r in range(world)]
                full["fanin"]["rows_intact"] = bool(arrived == [int(x.item()) for x in sums] and int(got_counts.numel()) == T * world
                                                    and int(got_counts.min()) > 0)
        if stage("fanin", fanin) is None and "fanin" in errors:
            full.setdefault("fanin", {})["error"] = errors["fanin"]
        torch.cuda.empty_cache()

    # every rank checks sampled tiles of ITS buffer against the oracle (the checker, after the timed region); rank 0 reports
    if dist and not args.no_check:
        def check_ranks():
            try:
                verdict = spot_check(cfg, d, x, T, L, n_check=2)
            except Exception as e:                       # (no oracle on this box: report it, never fail the measurement over the checker)
                verdict = f"not checked: {type(e).__name__}: {e}"
            full["check"] = f"rank 0: {verdict} (the reduction over ranks did not complete)"
            ok = torch.tensor([1 if "identical" in verdict else 0], dtype=torch.int32, device=coll_dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            full["check"] = (f"every rank: {verdict}" if int(ok[0]) else f"NOT byte-identical on every rank (rank {rank}: {verdict})")
        stage("check", check_ranks)

    if rank != 0:
        em.stage = "exit"
        em.emit()                                        # (marks this rank done: its watchdog stands down)
        if dist:
            try:
                dist.barrier()
                dist.destroy_process_group()
            except Exception:
                pass
        return

    if world == 1 and not args.no_cpu_baseline:
        # The CPU leg: the only place in this file that touches oracle/ — it times the reference's own
        # code on the host cores and (unless --no-check) uses the oracle as CHECKER on sampled tiles.
        def put(key, fn):
            r = stage(key, fn)
            full[key] = r if r is not None else {"error": errors.get(key, "no result")}
        put("cpu_baseline", lambda: cpu_baseline(cfg))
        if not args.no_check:
            full["check"] = stage("check", lambda: spot_check(cfg, d, x, T, L)) or f"not checked: {errors.get('check')}"
        del soft
        stage("close", d.close)
        torch.cuda.empty_cache()
        if not args.no_check:
            put("single_recording", lambda: recordings_leg(args.config, buf, local, buf_stream=rec))
        del buf, x
        torch.cuda.empty_cache()
        put("other_configs", lambda: other_configs(args.config, T, L, local))
        one = stage("one_core_all_configs", one_core_all_configs)
        if isinstance(full.get("cpu_baseline"), dict) and one is not None:
            full["cpu_baseline"]["one_core_msps"] = one
        put("cli_wall_times", lambda: cli_wall_times(local))
        put("host_fed", lambda: host_fed(local))
    em.stage = "emit"
    em.emit()
    if dist:
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception:
            pass


if __name__ == "__main__":
    main()
