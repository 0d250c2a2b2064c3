"""GPU tests (-m gpu): the kernels stay inside what they were given.  There is no address sanitizer for the device on this pool, so the
output rows sit in a buffer of canaries (before the first row, behind the last, between a row's capacity and the row stride, behind a
stream's own symbols) and the input streams sit in a buffer of full-scale garbage (a sample read from outside a stream's block, and
used, changes the bytes against the oracle's).  Every kernel family, ragged batches with empty and one-sample streams, stream counts
that are no multiple of a wave or a block."""
from __future__ import annotations

import ctypes as C
import zlib

import numpy as np
import pytest

import oracle_py as O
from meteor_demod_amd import DemodConfig, Demodulator, synth
from meteor_demod_amd.demod import check

pytestmark = pytest.mark.gpu

CANARY = 0x5A
# (name, configuration, environment, part of the kernel's name)
FAMILIES = [
    ("std_qpsk", DemodConfig(samplerate=230000), {"MDEMOD_LAT": "0"}, "demod_kernel_rot "),
    ("std_oqpsk_u8", DemodConfig(samplerate=230000, symrate=80000, oqpsk=True, bps=8), {"MDEMOD_LAT": "0"}, "demod_kernel_rot "),
    ("wide", DemodConfig(samplerate=1000000, rrc_order=64, interp_factor=8), {"MDEMOD_LAT": "0"}, "demod_kernel_rotp"),
    ("mid", DemodConfig(samplerate=1024000), {"MDEMOD_LAT": "0"}, "demod_kernel_rotp"),
    ("far_u8", DemodConfig(samplerate=2150000, rrc_order=20, interp_factor=3, bps=8), {"MDEMOD_LAT": "0"}, "demod_kernel_rotp"),
    ("hybrid_f32", DemodConfig(samplerate=1000000, rrc_order=64, interp_factor=8, bps=32), {"MDEMOD_LAT": "0"}, "demod_kernel_roth"),
    ("gather", DemodConfig(samplerate=10000000), {"MDEMOD_LAT": "0"}, "demod_kernel_gat"),
    ("v1_ring", DemodConfig(samplerate=230000), {"MDEMOD_LAT": "0", "MDEMOD_KERNEL": "v1"}, "demod_kernel "),
    ("v1_ring_f32", DemodConfig(samplerate=230000, bps=32), {"MDEMOD_LAT": "0", "MDEMOD_KERNEL": "v1"}, "demod_kernel "),
    ("latency", DemodConfig(samplerate=230000), {"MDEMOD_LAT": "1"}, "demod_kernel_lat"),
    ("latency_oqpsk_f32", DemodConfig(samplerate=640000, symrate=80000, oqpsk=True, rrc_order=24, interp_factor=4, bps=32), {"MDEMOD_LAT": "1"}, "demod_kernel_lat"),
]
NP_DTYPE = {8: np.uint8, 16: np.int16, 32: np.float32}


def _garbage(shape, bps, rng):
    if bps == 8:
        return rng.choice(np.array([0, 255], np.uint8), size=shape)
    if bps == 16:
        return rng.choice(np.array([-32768, 32767], np.int16), size=shape)
    return rng.choice(np.array([-3.0e4, 3.0e4], np.float32), size=shape)


@pytest.mark.parametrize("name,cfg,env,kernel", FAMILIES, ids=[f[0] for f in FAMILIES])
@pytest.mark.parametrize("tight", [False, True], ids=["roomy", "tight"])
def test_kernels_stay_inside_their_rows(name, cfg, env, kernel, tight, gpu_device, monkeypatch):
    import torch
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    per_sym = cfg.samplerate / cfg.symrate
    base = int(400 * per_sym)                                            # a few hundred symbols per stream: the oracle finishes in seconds
    lens = [0, 1, 2, base, base + 1, 63, 64, 65, 3 * base + 7, 0, base // 2, 5, 2 * base, base + 3, 17, base, 4 * base + 1]
    lens += [base + i for i in range(70 - len(lens))]                    # 70 streams: more than a wave, no multiple of anything
    ns = len(lens)
    streams = [synth.make_stream(9000 + i, cfg.samplerate, cfg.symrate, oqpsk=cfg.oqpsk, f0_hz=37.0 * i - 900.0, esn0_db=15.0) for i in range(ns)]
    dt = NP_DTYPE[cfg.bps]
    iqs = []
    for s, n in zip(streams, lens):
        a = synth.generate_host(s, n) if n else np.zeros((0, 2), np.int16)
        if cfg.bps == 8:
            a = (np.clip(a // 64, -128, 127) + 128).astype(np.uint8)
        elif cfg.bps == 32:
            a = a.astype(np.float32)
        iqs.append(a)
    # the streams inside full-scale garbage, at odd sample offsets (no 16-byte alignment)
    offsets, pos = [], 301
    for a in iqs:
        offsets.append(pos)
        pos += a.shape[0] + int(rng.integers(1, 9))
    flat = _garbage((pos + 500, 2), cfg.bps, rng).astype(dt)
    for o, a in zip(offsets, iqs):
        flat[o:o + a.shape[0]] = a
    with Demodulator(cfg, ns) as d:
        assert kernel in d.kernel_name + " ", d.kernel_name
        cap = d.max_symbols(max(lens)) if not tight else 96              # tight: most streams produce more than fits (overflow reported, nothing written behind cap)
        stride = cap + 24
        pad = 4096
        big = torch.full((pad + ns * stride * 2 + pad,), CANARY, dtype=torch.int8, device="cuda")
        soft = big[pad: pad + ns * stride * 2].view(ns, stride, 2)
        flat_d = torch.from_numpy(flat).cuda()
        off_d = torch.tensor(offsets, dtype=torch.int64).cuda()
        cnt_d = torch.tensor(lens, dtype=torch.int32).cuda()
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(d._lib.mdemod_process_device(d._ctx, C.c_void_p(flat_d.data_ptr()), C.c_void_p(off_d.data_ptr()), C.c_void_p(cnt_d.data_ptr()),
                                           C.c_void_p(soft.data_ptr()), stride, cap, stream), "mdemod_process_device")
        torch.cuda.synchronize()
        st = d.status()
        got = big.cpu().numpy()
    assert (got[:pad] == CANARY).all(), "bytes written in front of the first row"
    assert (got[pad + ns * stride * 2:] == CANARY).all(), "bytes written behind the last row"
    rows = got[pad: pad + ns * stride * 2].reshape(ns, stride, 2)
    assert (rows[:, cap:] == CANARY).all(), "bytes written between a row's capacity and the stride"
    for i, a in enumerate(iqs):
        want = O.oracle_demod(cfg, a)[0] if a.shape[0] else np.zeros((0, 2), np.int8)
        assert st[i].symbols_this_call == want.shape[0], (i, lens[i])
        assert st[i].overflow == (1 if want.shape[0] > cap else 0), i
        m = min(cap, want.shape[0])
        assert np.array_equal(rows[i, :m], want[:m]), (i, lens[i])      # full-scale garbage either side of the block changed nothing
        assert (rows[i, m:cap] == CANARY).all(), (i, "bytes written behind the stream's own symbols")


@pytest.mark.parametrize("log2n", [17, 21], ids=["one_block", "fourteen_sub_blocks"])
@pytest.mark.parametrize("pinned", [False, True], ids=["staged", "pinned_rows"])
def test_host_entry_stays_inside_the_callers_rows(pinned, log2n, gpu_device):
    """mdemod_process_host: output rows of the caller's inside canaries (the unpack's streaming stores: an aligning head, 64-byte
    bodies, a tail), input rows inside full-scale garbage; several sub-blocks (the pipeline), capacities that fit exactly."""
    cfg = DemodConfig(samplerate=230000)
    rng = np.random.default_rng(77)
    ns, n = 48, 1 << log2n                                                # 25 MB: one block; 400 MB: 12 sub-blocks + the short ones at either end, the last three out through the compaction kernel
    st = synth.make_stream(4242, cfg.samplerate, cfg.symrate, f0_hz=250.0, esn0_db=15.0)
    one = synth.generate_host(st, n + ns)
    gap = 37                                                              # samples of garbage between the rows
    buf = _garbage((ns, n + gap, 2), 16, rng).astype(np.int16)
    for s in range(ns):
        buf[s, :n] = one[s: s + n]                                        # every stream its own shift of the signal
    want = [O.oracle_demod(cfg, np.ascontiguousarray(buf[s, :n]))[0] for s in range(0, ns, 12)]
    with Demodulator(cfg, ns) as d:
        if pinned:
            d.pin_host(buf)
        caps = [d.max_symbols(n)] * ns
        caps[0] = want[0].shape[0]                                        # exactly what stream 0 produces: the last byte written is the row's last
        stride = max(caps) + 19
        pad = 1000
        out = np.full(pad + ns * stride * 2 + pad, CANARY, np.int8)
        rows = out[pad: pad + ns * stride * 2].reshape(ns, stride, 2)
        iq_ptrs = (C.c_void_p * ns)(*[buf[s].ctypes.data for s in range(ns)])
        counts = (C.c_uint32 * ns)(*([n] * ns))
        soft_ptrs = (C.c_void_p * ns)(*[rows[s].ctypes.data for s in range(ns)])
        soft_caps = (C.c_uint32 * ns)(*caps)
        produced = (C.c_uint32 * ns)()
        check(d._lib.mdemod_process_host(d._ctx, iq_ptrs, counts, soft_ptrs, soft_caps, produced), "mdemod_process_host")
        if pinned:
            d.unpin_host(buf)
    assert (out[:pad] == CANARY).all() and (out[pad + ns * stride * 2:] == CANARY).all()
    for s in range(ns):
        assert (rows[s, produced[s]:] == CANARY).all(), s
    for k, s in enumerate(range(0, ns, 12)):
        assert produced[s] == want[k].shape[0] and np.array_equal(rows[s, : produced[s]], want[k]), s


@pytest.mark.parametrize("tag,cfg,log2n", [("qpsk", DemodConfig(samplerate=230000), 23), ("oqpsk", DemodConfig(samplerate=230000, symrate=80000, oqpsk=True), 23)],
                         ids=["qpsk", "oqpsk"])
def test_recording_entry_stays_inside_its_output(tag, cfg, log2n, gpu_device):
    """mdemod_demodulate_recording (serial head + tile banks + stitching + compaction): the stitched output inside canaries, with
    room to spare and with exactly the capacity it needs; one symbol less is MDEMOD_ERR_OVERFLOW and still writes nothing outside."""
    import torch
    from meteor_demod_amd.recording import demodulate_recording_native
    n = 1 << log2n
    st = synth.make_stream(515, cfg.samplerate, cfg.symrate, oqpsk=cfg.oqpsk, f0_hz=800.0, esn0_db=14.0)
    iq = synth.generate_device([st], n)[0].contiguous()
    ref, rep0 = demodulate_recording_native(cfg, iq)
    m = int(rep0.n_symbols)
    assert m > 0.9 * n * cfg.symrate / cfg.samplerate and rep0.n_tiles > 1
    pad = 8192
    for cap, ok in ((m + 5000, True), (m, True), (m - 1, False)):
        big = torch.full((pad + 2 * cap + pad,), CANARY, dtype=torch.int8, device="cuda")
        soft = big[pad: pad + 2 * cap].view(cap, 2)
        if ok:
            out, rep = demodulate_recording_native(cfg, iq, soft=soft)
            assert int(rep.n_symbols) == m and torch.equal(out, ref)
        else:
            with pytest.raises(RuntimeError, match="-4|overflow|OVERFLOW"):
                demodulate_recording_native(cfg, iq, soft=soft)
        torch.cuda.synchronize()
        got = big.cpu().numpy()
        assert (got[:pad] == CANARY).all() and (got[pad + 2 * cap:] == CANARY).all(), (cap, ok)
        if ok:
            assert (got[pad + 2 * m: pad + 2 * cap] == CANARY).all(), "bytes written behind the recording's own symbols"


def test_failures_come_back_as_text_not_on_stderr(gpu_device, capfd):
    """VERDICT r05 item 6 on the device side: a HIP failure (a device that does not exist) and the refusals that need a context (a
    clock word / carrier state the reference's loop cannot hold, an overlapping pin) return their code, leave the words in
    mdemod_last_error() and print nothing; the next successful entry empties the text."""
    import torch
    from meteor_demod_amd import _capi
    lib = _capi.lib()
    capfd.readouterr()
    p = DemodConfig(samplerate=230000).to_c(4, torch.cuda.device_count() + 7)
    ctx = C.c_void_p()
    assert lib.mdemod_create(C.byref(p), C.byref(ctx)) == _capi.MDEMOD_ERR_HIP and not ctx.value
    said = _capi.last_error()
    assert "no usable HIP device" in said and str(torch.cuda.device_count() + 7) in said, said
    with Demodulator(DemodConfig(samplerate=230000), 4) as d:
        assert _capi.last_error() == ""                                  # mdemod_create succeeded: nothing left from the failure before
        st = d.get_states()[0]
        st.t_freq = 0.5
        with pytest.raises(_capi.MdemodError) as e:
            d.set_state(0, st)
        assert "t_freq" in e.value.detail and "timing.c:80-86" in e.value.detail
        st = d.get_states()[0]
        st.pll_phase = 20.0
        with pytest.raises(_capi.MdemodError) as e:
            d.set_state_all(st)
        assert "pll_phase" in e.value.detail
        buf = np.zeros((4, 4096, 2), dtype=np.int16)
        d.pin_host(buf)
        with pytest.raises(_capi.MdemodError):
            d.pin_host(buf[1:3])
        d.unpin_host(buf)
        d.status()
        assert _capi.last_error() == ""
        # an error the CALLER's own HIP code left pending on this thread is not the status of the library's next launch (the launch
        # wrappers read hipGetLastError() after their launch; r06: select_device drops what is pending first)
        x = synth.generate_device([synth.make_stream(5 + i, 230000, 72000) for i in range(4)], 6000)
        soft = torch.empty((4, d.max_symbols(6000), 2), dtype=torch.int8, device="cuda")
        torch.cuda.synchronize()
        # THE runtime this process already runs on (torch's own copy of libamdhip64, which serves the library too: _capi.hip_runtime_first) -
        # by the path it is mapped from; dlopen("libamdhip64.so") by name would bring the system's copy in as a SECOND runtime
        # (r06: that is what this test did at first - it then proved nothing, and the process crashed at exit once in five runs)
        mapped = sorted({line.split()[-1] for line in open("/proc/self/maps") if "libamdhip64.so" in line})
        assert len(mapped) == 1, mapped
        hip = C.CDLL(mapped[0])
        assert hip.hipSetDevice(torch.cuda.device_count() + 7) != 0         # (torch's own next call would raise on this pending error)
        assert hip.hipPeekAtLastError() != 0                                 # it IS pending in the runtime the library launches on
        d.process(x, soft=soft)
        torch.cuda.synchronize()
        want = O.oracle_demod(DemodConfig(samplerate=230000), x[2].cpu().numpy())[0]
        assert np.array_equal(soft[2, : len(want)].cpu().numpy(), want)
    out = capfd.readouterr()
    assert "meteor_demod_amd" not in out.err and "failed" not in out.err, out.err


def test_a_context_the_device_has_no_room_for_and_indices_out_of_range(gpu_device, capfd):
    """MDEMOD_ERR_NOMEM from mdemod_create (600 M streams: the history alone is 384 GB), with the failing hipMalloc named in
    mdemod_last_error(), everything allocated before it given back (a normal context fits right afterwards and gives the oracle's
    bytes); MDEMOD_ERR_RANGE for stream indices beyond the context from every per-stream entry; nothing printed."""
    import torch
    from meteor_demod_amd import _capi
    lib = _capi.lib()
    capfd.readouterr()
    free0 = torch.cuda.mem_get_info()[0]
    p = DemodConfig(samplerate=230000).to_c(600_000_000, 0)
    ctx = C.c_void_p()
    assert lib.mdemod_create(C.byref(p), C.byref(ctx)) == _capi.MDEMOD_ERR_NOMEM and not ctx.value
    said = _capi.last_error()
    assert "hipMalloc" in said and ("memory" in said.lower()), said
    assert torch.cuda.mem_get_info()[0] >= free0 - (64 << 20)            # what had been allocated was given back
    with Demodulator(DemodConfig(samplerate=230000), 8) as d:
        x = synth.generate_device([synth.make_stream(50 + i, 230000, 72000) for i in range(8)], 5000)
        soft = d.process(x)
        torch.cuda.synchronize()
        want = O.oracle_demod(DemodConfig(samplerate=230000), x[7].cpu().numpy())[0]
        assert np.array_equal(soft[7, : len(want)].cpu().numpy(), want)
        st = d.get_state(7)
        for call in (lambda: d.get_state(8), lambda: d.set_state(8, st), lambda: d.status(4, 5), lambda: d.status(8, 1)):
            with pytest.raises(_capi.MdemodError) as e:
                call()
            assert e.value.code == _capi.MDEMOD_ERR_RANGE, e.value
        ev = (_capi.MdemodLockEvent * 4)() if hasattr(_capi, "MdemodLockEvent") else None
        if ev is not None:
            n = C.c_uint32()
            assert lib.mdemod_get_lock_events(d._ctx, 8, ev, 4, C.byref(n), None) == _capi.MDEMOD_ERR_RANGE
        assert d.get_state(0).n_samples == 5000                             # the context is unharmed
    out = capfd.readouterr()
    assert "meteor_demod_amd" not in out.err, out.err


@pytest.mark.parametrize("pools", ["", "3"], ids=["default-pools", "3-pools-of-4"])
def test_host_entry_from_concurrent_host_threads(gpu_device, pools):
    """One library context per host thread (the C host's --devices workers; here three threads on one GPU), all inside
    mdemod_process_host at once, staged and pinned rows alike: every thread gets byte for byte what it gets alone.  Run in a child
    process, once with the pack pools as the host allows and once forced to three pools of four threads (r06: callers that find
    the pool busy get one of their own)."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    code = r'''
import threading, numpy as np, torch
from meteor_demod_amd import DemodConfig, Demodulator, synth
cfg = DemodConfig(samplerate=230000)
ns, n = 96, 60000                      # 23 MB of input per call: well above the pool's 1 MB threshold, several sub-blocks
one = [synth.generate_host(synth.make_stream(700 + k, 230000, 72000, f0_hz=40.0 * k), n + ns) for k in range(3)]
bufs = [np.stack([o[s: s + n] for s in range(ns)]) for o in one]
def run(k, pin):
    with Demodulator(cfg, ns) as d:
        if pin:
            d.pin_host(bufs[k])
        a = d.process_host([bufs[k][s] for s in range(ns)])
        b = d.process_host([bufs[k][s][: 20000 + 13 * s] for s in range(ns)])        # ragged, chained on the first call
        return a, b
alone = [run(k, False) for k in range(3)]
for pin in (False, True):
    out, err = [None] * 3, []
    def work(k):
        try:
            out[k] = run(k, pin)
        except Exception as e:
            err.append((k, repr(e)))
    for rounds in range(3):
        th = [threading.Thread(target=work, args=(k,)) for k in range(3)]
        for t in th: t.start()
        for t in th: t.join()
        assert not err, err
        for k in range(3):
            for x, y in zip(out[k][0] + out[k][1], alone[k][0] + alone[k][1]):
                assert np.array_equal(x, y), (pin, rounds, k)
print("OK")
'''
    env = dict(os.environ)
    if pools:
        env.update(MDEMOD_PACK_POOLS=pools, MDEMOD_PACK_THREADS="4")
    root = Path(__file__).resolve().parent.parent
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=str(root), env=env, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), (r.stdout[-500:], r.stderr[-2000:])


def test_misuse_comes_back_as_a_code_and_harms_nobody(gpu_device):
    """Entries handed the wrong things: a bank's state into a bank of another size / modulation / filter, histories of the wrong
    length, NULL contexts, a soft buffer that is too small, a row pitch below the symbol counts, seeds of the wrong size.  Every one
    is a code (or the Python mirror's own refusal), and a neighbouring context - and the abused one after a reset - still gives the
    oracle's bytes."""
    import torch
    from meteor_demod_amd import _capi
    lib = _capi.lib()
    C1 = DemodConfig(samplerate=230000)
    C3 = DemodConfig(samplerate=230000, symrate=80000, oqpsk=True)
    CF = DemodConfig(samplerate=230000, rrc_order=24)
    x = synth.generate_device([synth.make_stream(5 + i, 230000, 72000) for i in range(8)], 6000)
    want = [O.oracle_demod(C1, x[i].cpu().numpy())[0] for i in (0, 3, 7)]
    with Demodulator(C1, 8) as a, Demodulator(C1, 4) as b, Demodulator(C3, 8) as c, Demodulator(CF, 8) as f, Demodulator(C1, 8) as twin:
        a.process(x)
        torch.cuda.synchronize()
        for other in (b, c, f):
            assert lib.mdemod_copy_state(other._ctx, a._ctx, None) == _capi.MDEMOD_ERR_PARAM
            assert "mdemod_copy_state" in _capi.last_error()
        assert lib.mdemod_copy_state(twin._ctx, a._ctx, None) == 0 and lib.mdemod_copy_state(a._ctx, a._ctx, None) == 0
        assert lib.mdemod_get_status(None, 0, 1, None, None) == _capi.MDEMOD_ERR_PARAM
        assert lib.mdemod_copy_state(None, a._ctx, None) == _capi.MDEMOD_ERR_PARAM
        lib.mdemod_destroy(None)                                            # a no-op, like free(NULL)
        h = a.get_history(0)
        with pytest.raises((ValueError, _capi.MdemodError)):
            a.set_history(0, h[:10])
        # a soft buffer that is too small: the device entries are asynchronous, so the overflow is the status's to report - the rows
        # hold their first 16 symbols, nothing is written behind them
        big = torch.full((8 * 16 * 2 + 4096,), CANARY, dtype=torch.int8, device="cuda")
        small = big[: 8 * 16 * 2].view(8, 16, 2)
        a.reset()
        a.process(x, soft=small)
        torch.cuda.synchronize()
        sts = a.status()
        assert all(s.overflow == 1 for s in sts) and bool((big[8 * 16 * 2:] == CANARY).all())
        assert np.array_equal(small[3].cpu().numpy(), want[1][:16])
        with pytest.raises((AssertionError, ValueError, _capi.MdemodError)):
            a.rotate_carrier(torch.zeros(3, dtype=torch.int32, device="cuda"))
        with pytest.raises((AssertionError, ValueError, _capi.MdemodError)):
            a.set_clock_seeds(torch.zeros(3, dtype=torch.float32, device="cuda"))
        a.reset()
        s2 = a.process(x)
        st = twin.process(x[:, :10])                                        # the twin took a's state after 6000 samples: it continues, untouched by the abuse
        torch.cuda.synchronize()
        for k, i in enumerate((0, 3, 7)):
            assert np.array_equal(s2[i, : len(want[k])].cpu().numpy(), want[k]), i
        assert twin.get_state(0).n_samples == 6010
        counts = a.symbol_counts()
        packed = a.compact(s2, 8)                                           # a pitch below the counts: rows are cut, nothing is written past them
        assert packed.shape == (8, 8, 2) and torch.equal(packed[5], s2[5, :8]) and int(counts.min()) > 8
