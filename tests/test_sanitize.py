"""Sanitizer + fuzz gate of the HIP-free host code (CPU suite).

Nothing in the product was ever built with a sanitizer before round 5, and round 4's `cj_schedule` looped forever on parameter sets
`mdemod_host_derive` accepts.  This file builds, with gcc's sanitizers, everything on the host side that does not need the HIP runtime:

  fuzz_derive   csrc/demod_host.cpp + csrc/clock_jump.h   ASan + UBSan   >= 1e5 random / edge mdemod_params, wall-clock bound per call,
                                                                         every schedule checked against sequential stepping
  fuzz_hostc    host/meteor_demod_amd.c helpers           ASan + UBSan   write_gated vs main.c:305-315 per symbol, WAV headers, numbers
  pool_test     csrc/pack_pool.h                          TSan / ASan    the packing threads, shared by several callers; fork + exit
  the C host    host/meteor_demod_amd.c (whole program)   TSan and ASan  against tests/sanitize/stub_backend.c: worker threads per
                                                                         device, --tiled job threads, outputs == a model of main.c:303-322

GPU sanitizers are not available on this pool and are not asked for.  The binaries land in a temporary directory; they are test
infrastructure and never part of the product."""
from __future__ import annotations

import json
import os
import subprocess
from pathlib import Path

import numpy as np
import pytest

from conftest import ROOT
from golden_cases import wav_header

SAN = ROOT / "tests" / "sanitize"
CSRC = ROOT / "meteor_demod_amd" / "csrc"
ASAN = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer"]
TSAN = ["-O1", "-g", "-fsanitize=thread", "-fno-omit-frame-pointer"]
ENV = dict(os.environ, ASAN_OPTIONS="abort_on_error=0:detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
           TSAN_OPTIONS="halt_on_error=1:second_deadlock_stack=1")


def _cc(out: Path, cmd: list[str]) -> Path:
    proc = subprocess.run(cmd + ["-o", str(out)], capture_output=True, text=True)
    assert proc.returncode == 0, proc.stderr[-3000:]
    return out


@pytest.fixture(scope="module")
def bins(tmp_path_factory):
    d = tmp_path_factory.mktemp("sanitize")
    inc = ["-I", str(ROOT / "include")]
    host_c = str(ROOT / "host" / "meteor_demod_amd.c")
    stub = str(SAN / "stub_backend.c")
    return {
        "derive": _cc(d / "fuzz_derive", ["g++", "-std=c++17", *ASAN, "-ffp-contract=off", *inc, str(SAN / "fuzz_derive.cpp"),
                                           str(CSRC / "demod_host.cpp"), "-pthread"]),
        "hostc": _cc(d / "fuzz_hostc", ["gcc", "-std=gnu11", *ASAN, *inc, str(SAN / "fuzz_hostc.c"), stub, "-pthread", "-lm"]),
        "pool_tsan": _cc(d / "pool_tsan", ["g++", "-std=c++17", *TSAN, str(SAN / "pool_test.cpp"), "-pthread"]),
        "pool_asan": _cc(d / "pool_asan", ["g++", "-std=c++17", *ASAN, str(SAN / "pool_test.cpp"), "-pthread"]),
        "cli_tsan": _cc(d / "cli_tsan", ["gcc", "-std=gnu11", *TSAN, *inc, host_c, stub, "-pthread", "-lm"]),
        "cli_asan": _cc(d / "cli_asan", ["gcc", "-std=gnu11", *ASAN, *inc, host_c, stub, "-pthread", "-lm"]),
    }


def _run(cmd, timeout, env=ENV, **kw):
    cap = dict(stderr=subprocess.PIPE) if "stdout" in kw else dict(capture_output=True)       # (a caller's own stdout: only stderr is captured)
    proc = subprocess.run([str(c) for c in cmd], timeout=timeout, env=env, **cap, **kw)
    err = proc.stderr.decode(errors="replace")
    assert "Sanitizer" not in err and "runtime error" not in err, err[-4000:]
    return proc


@pytest.mark.timeout(600)
def test_derive_fuzz_under_asan_ubsan(bins):
    """>= 1e5 random mdemod_params through mdemod_host_derive: returns within the bound, never trips ASan / UBSan, and every clock
    schedule it hands to the kernels reproduces timing.c:32-38's additions.  The seed changes nothing about what must hold."""
    proc = _run([bins["derive"], 120000, 5, 8, 5.0], timeout=500)
    assert proc.returncode == 0, proc.stderr.decode()[-3000:]
    rep = json.loads(proc.stdout.decode().strip().splitlines()[-1])
    assert rep["ok"] and rep["cases"] == 120000
    # the draw really covers the accepted region, the schedules, and the region that hung in round 4
    assert rep["accepted"] > 40000 and rep["with_clock_schedule"] > 5000 and rep["below_one_sample_per_firing"] > 5000, rep


def test_round4_hang_parameters_return(bins):
    """The three command lines VERDICT r04 quotes (symrate >= 2 fs O for OQPSK, 4 fs O for QPSK): mdemod_derive_tables of the product
    library itself returns at once."""
    import ctypes as C
    from meteor_demod_amd import DemodConfig, _capi
    lib = _capi.lib()
    for kw in (dict(samplerate=36000, symrate=72000, interp_factor=1, oqpsk=True), dict(samplerate=18000, symrate=72000, interp_factor=1),
               dict(samplerate=20000, symrate=80000, interp_factor=2, oqpsk=True), dict(samplerate=9000, symrate=72000, interp_factor=1, oqpsk=True)):
        p = DemodConfig(**kw).to_c()
        assert lib.mdemod_derive_tables(C.byref(p), None, 0, None, None) == 65 * kw["interp_factor"], kw          # (the table's length: accepted)


def test_host_helpers_fuzz_under_asan_ubsan(bins):
    proc = _run([bins["hostc"], 3000, 11], timeout=300)
    assert proc.returncode == 0, (proc.stdout + proc.stderr).decode()[-3000:]


def test_pack_pool_under_tsan(bins):
    proc = _run([bins["pool_tsan"], "race", 6, 300], timeout=300)
    assert proc.returncode == 0, (proc.stdout + proc.stderr).decode()[-3000:]
    assert b" 0 bad" in proc.stdout


def test_pack_pool_is_sized_by_the_cpus_it_may_use(bins):
    """the pool was min(12, hardware_concurrency()), which ignores a container's cpuset and the cgroup CPU quota (12 pack
    threads time-slicing on 2 CPUs are slower than 2).  PackPool::usable_cpus() = affinity mask and cpu.max, whichever is smaller."""
    import os
    import shutil

    def quota_cpus():
        try:
            q, per = Path("/sys/fs/cgroup/cpu.max").read_text().split()[:2]
            return None if q == "max" else max(1, -(-int(q) // int(per)))
        except (OSError, ValueError):
            return None
    want = min(x for x in (len(os.sched_getaffinity(0)), quota_cpus(), os.cpu_count()) if x)
    out = _run([bins["pool_asan"], "cpus"], timeout=60).stdout.split()
    assert int(out[1]) == want and int(out[3]) == min(12, want), (out, want)
    assert int(out[5]) == max(1, min(8, want // min(12, want)))            # pools: as many as the usable CPUs have room for
    out = _run([bins["pool_asan"], "cpus"], timeout=60, env=dict(ENV, MDEMOD_PACK_THREADS="3")).stdout.split()
    assert int(out[3]) == min(3, want)
    if shutil.which("taskset") and len(os.sched_getaffinity(0)) >= 2:
        two = sorted(os.sched_getaffinity(0))[:2]
        out = _run(["taskset", "-c", ",".join(map(str, two)), bins["pool_asan"], "cpus"], timeout=60).stdout.split()
        assert int(out[1]) == min(2, want) and int(out[3]) == min(2, want), out


def test_pack_pool_gives_concurrent_callers_pools_of_their_own(bins):
    """ADVICE r05, second half: the pool was process-wide and one job at a time - one library context per GPU is one host thread,
    so eight GPUs shared a single pack.  Callers that find a pool busy now get another one (up to usable CPUs / threads per pool;
    here forced: 3 pools of 2 threads).  `overlap` needs three callers INSIDE their jobs at the same moment; under TSan, and the
    race test on three pools."""
    env = dict(ENV, MDEMOD_PACK_POOLS="3", MDEMOD_PACK_THREADS="2")
    proc = _run([bins["pool_tsan"], "overlap", 3], timeout=120, env=env)
    assert proc.returncode == 0 and b"on 3 pools, 0 bad" in proc.stdout, proc.stdout
    proc = _run([bins["pool_tsan"], "race", 6, 150], timeout=300, env=env)
    assert proc.returncode == 0 and b"up to 3 pools of 2, 0 bad" in proc.stdout, proc.stdout
    proc = _run([bins["pool_asan"], "fork"], timeout=60, env=dict(env, ASAN_OPTIONS="detect_leaks=0"))     # (the child leaks on purpose: see the next test)
    assert proc.returncode == 0, proc.stdout


def test_pack_pool_forked_child_exits(bins):
    """~PackPool in a forked child joined threads that do not exist there.  (The child leaks the pool's state on purpose,
    so LeakSanitizer is off for this one run; ASan and UBSan stay on.)"""
    env = dict(ENV, ASAN_OPTIONS="detect_leaks=0")
    proc = _run([bins["pool_asan"], "fork"], timeout=60, env=env)
    assert proc.returncode == 0, (proc.stdout + proc.stderr).decode()[-3000:]
    proc = _run([bins["pool_asan"], "race", 4, 100], timeout=120)
    assert proc.returncode == 0


def test_streaming_copy_equals_memcpy_under_asan(bins):
    """csrc/pack_pool.h: stream_copy (the pack into the pinned ring and the unpack into the caller's rows: non-temporal stores behind an
    aligning head and a tail) for every alignment and the lengths around its steps, guard bytes checked."""
    proc = _run([bins["pool_asan"], "copy"], timeout=120)
    assert proc.returncode == 0 and b"equal memcpy" in proc.stdout, (proc.stdout + proc.stderr).decode()[-2000:]


# ---- the whole C host against the stub backend -------------------------------------------------------------------------------

def _model_output(data: bytes, bps_opt: int, lock: int, decim: int = 3) -> bytes:
    """main.c:303-322 + wavfile.c:16-69 for the stub's "demodulator" (one symbol per `decim` samples: the sample's first two bytes)."""
    bps, off = bps_opt, 0
    if len(data) >= 44 and data[:4] == b"RIFF" and data[8:12] == b"WAVE" and int.from_bytes(data[22:24], "little") == 2:
        bits = int.from_bytes(data[34:36], "little")
        bps = bits
        if bits:
            off = 44
    if bps == 0:
        bps = 16
    body = data[off:]
    body = body[: len(body) // 32768 * 32768]                    # whole reads only (wavfile.c:55)
    sb = 2 * bps // 8
    raw = np.frombuffer(body, dtype=np.uint8).reshape(-1, sb)[::decim, :2].astype(np.int8) if len(body) else np.zeros((0, 2), np.int8)
    n = len(raw)
    out = bytearray()
    ring = np.zeros((512, 2), np.int8)
    full = n // 512
    for c in range(full):
        ring[:] = raw[512 * c: 512 * c + 512]
        if lock >= 0 and lock < n and lock <= 512 * c + 511:
            out += ring.tobytes()
    rest = n - 512 * full
    ring[:rest] = raw[512 * full:]
    tail = min(4 * rest, 1024)                                   # fwrite(ring, ring_idx, 2, f): ring_idx = 2 * rest int8 entries
    out += ring.tobytes()[:tail]
    return bytes(out)


def _make_inputs(tmp_path, rng):
    files = []
    for i, (kind, nbytes) in enumerate((("wav16", 10 * 32768 + 1234), ("wav16", 3 * 32768), ("wav16", 32768 * 7 + 44 + 5), ("wav16", 500),
                                        ("wav16", 32768 * 21 + 17), ("wav16", 32768 * 2 + 44))):
        body = rng.integers(0, 256, nbytes, dtype=np.uint8).tobytes()
        data = wav_header(200000, 16, nbytes) + body
        f = tmp_path / f"in{i}.wav"
        f.write_bytes(data)
        files.append((f, data))
    return files


@pytest.mark.parametrize("san", ["tsan", "asan"])
@pytest.mark.parametrize("mode", ["exact-1dev", "exact-3dev", "tiled-jobs4", "tiled-2dev"])
def test_c_host_against_stub_backend(bins, san, mode, tmp_path):
    """The host program with its threads (one worker per device; --tiled: up to --jobs files of a worker in flight) against the stub
    backend: every output file equals the model of the reference's file handling, and the sanitizer has nothing to say."""
    rng = np.random.default_rng(5)
    files = _make_inputs(tmp_path, rng)
    lock = 1000
    env = dict(ENV, STUB_LOCK=str(lock), STUB_DEVICES="3")
    args = ["-q", "-B", "-r", "72000"]
    if mode == "exact-3dev":
        args += ["--devices", "0,1,2"]
    elif mode == "exact-1dev":
        args += ["--device", "0"]
    elif mode == "tiled-jobs4":
        args += ["--tiled", "--device", "0", "--jobs", "4"]
    else:
        args += ["--tiled", "--devices", "0,1", "--jobs", "2"]
    proc = _run([bins[f"cli_{san}"], *args, *[f for f, _ in files]], timeout=300, env=env, cwd=tmp_path)
    assert proc.returncode == 0, (proc.stdout + proc.stderr).decode()[-3000:]
    for f, data in files:
        got = Path(str(f) + ".s").read_bytes()
        assert got == _model_output(data, 0, lock), f.name


@pytest.mark.parametrize("san", ["asan"])
def test_c_host_option_edges_against_stub_backend(bins, san, tmp_path):
    """Option values and headers a user can hand over (main.c:82-152 takes them all): no crash, no sanitizer report, the reference's
    exit behaviour where it has one."""
    rng = np.random.default_rng(6)
    raw = tmp_path / "x.raw"
    raw.write_bytes(rng.integers(0, 256, 5 * 32768 + 99, dtype=np.uint8).tobytes())
    cli = bins[f"cli_{san}"]
    env = dict(ENV, STUB_LOCK="100")

    def run(*a, **kw):
        return _run([cli, *a], timeout=120, env=env, cwd=tmp_path, **kw)

    # raw input needs -s (main.c:167-171)
    assert run("-q", "-B", "-o", "o.s", raw).returncode == 1
    assert run("-q", "-B", "-s", "200k", "-o", "o.s", raw).returncode == 0
    assert (tmp_path / "o.s").read_bytes() == _model_output(raw.read_bytes(), 16, 100)
    # u8 and f32 raw
    for bps in (8, 32):
        assert run("-q", "-B", "-s", "200k", "--bps", str(bps), "-o", "o.s", raw).returncode == 0
        assert (tmp_path / "o.s").read_bytes() == _model_output(raw.read_bytes(), bps, 100)
    # a sample size the reader does not know: an empty output (wavfile.c:71-73), exit 0
    assert run("-q", "-B", "-s", "200k", "--bps", "24", "-o", "o.s", raw).returncode == 0
    assert (tmp_path / "o.s").read_bytes() == b""
    # parameters the library refuses: exit 2, in both modes, no allocation sized from them
    for bad in (["-O", "0"], ["-O", "65"], ["-f", "0"], ["-f", "300"], ["-r", "0"], ["-r", "3M"], ["-s", "0"]):
        for tiled in ([], ["--tiled"]):
            assert run("-q", "-B", "-s", "200k", *bad, *tiled, "-o", "o.s", raw).returncode == 2, (bad, tiled)
    # a WAV header with a zero sample rate / zero bits / one channel
    for rate, bits, ch, want in ((0, 16, 2, 2), (200000, 0, 2, 1), (200000, 16, 1, 1)):
        h = bytearray(wav_header(200000, 16, 1000))
        h[24:28] = int(rate).to_bytes(4, "little"); h[34:36] = int(bits).to_bytes(2, "little"); h[22:24] = int(ch).to_bytes(2, "little")
        w = tmp_path / "h.wav"
        w.write_bytes(bytes(h) + raw.read_bytes())
        for tiled in ([], ["--tiled"]):
            assert run("-q", "-B", *tiled, "-o", "o.s", w).returncode == want, (rate, bits, ch, tiled)
    # stdin / stdout plumbing
    proc = run("-q", "-s", "200k", "--stdout", "-", stdin=open(raw, "rb"))
    assert proc.returncode == 0 and proc.stdout == _model_output(raw.read_bytes(), 16, 100)
    # an output that cannot be written (a full disk): the reference ignores fwrite / fclose (main.c:314,321,274) and exits 0 with a short
    # file; so does this host - but it says so on stderr (r06), in both modes and on stdout
    if os.path.exists("/dev/full"):
        for tiled in ([], ["--tiled"]):
            proc = run("-q", "-B", "-s", "200k", *tiled, "-o", "/dev/full", raw)
            assert proc.returncode == 0 and b"/dev/full: writing the soft symbols failed" in proc.stderr, (tiled, proc.stderr)
        proc = run("-q", "-s", "200k", "--stdout", raw, stdout=open("/dev/full", "wb"))
        assert proc.returncode == 0 and b"(stdout): writing the soft symbols failed" in proc.stderr
    proc = run("-q", "-B", "-s", "200k", "-o", "o.s", raw)
    assert proc.returncode == 0 and b"failed" not in proc.stderr
    # --plan prints the round robin and touches nothing
    proc = run("--devices", "0,1", "--plan", "a", "b", "c")
    assert proc.returncode == 0 and proc.stdout.decode().split("\n")[:2] == ["device 0: a c", "device 1: b"]
    assert run("--devices", "0,x", "--plan", "a").returncode == 1
    assert run("--jobs", "0", raw).returncode == 1


ANALYZER = Path("/opt/rocm/lib/llvm/bin/clang")


@pytest.mark.skipif(not ANALYZER.exists(), reason="no clang static analyzer in this image")
def test_static_analyzer_is_silent_on_the_host_code(tmp_path):
    """clang --analyze (core, unix, deadcode, cplusplus checkers) over everything that runs on the host: the C host with and without
    its display, the host side of every C++ / HIP file behind the C-ABI, the oracle.  Round 5 found a dead store and nothing worse;
    the test keeps it that way."""
    jobs = [(["-x", "c", "-std=gnu99"] + d, ROOT / "host" / f) for f in ("meteor_demod_amd.c", "tui.c") for d in ([], ["-DMDEMOD_TUI"])]
    jobs += [(["-x", "c", "-std=gnu99"], ROOT / "oracle" / "lrpt_oracle.c")]
    jobs += [(["-x", "hip", "--cuda-host-only", "--offload-arch=gfx950", "-std=c++17", "-I/opt/rocm/include"], CSRC / f)
             for f in ("demod_host.cpp", "demod_api.cpp", "host_pipe.cpp", "recording.hip")]
    findings = []
    for flags, src in jobs:
        r = subprocess.run([str(ANALYZER), "--analyze", f"-I{ROOT / 'include'}", "-Xclang", "-analyzer-output=text", "-o", str(tmp_path / "a.plist")] + flags + [str(src)],
                           capture_output=True, text=True, timeout=600)
        findings += [ln for ln in r.stderr.splitlines() if "warning:" in ln and "nodiscard" not in ln and "-Wunused" not in ln]
        assert "error:" not in r.stderr, r.stderr[-2000:]
    assert not findings, "\n".join(findings)
