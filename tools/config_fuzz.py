"""Soak test: random -f/-O/-r/-s/-b/-d/-m/--bps combinations (every kernel geometry gets selected), several streams,
chained ragged blocks, GPU vs oracle byte for byte incl. final loop state.  Usage: config_fuzz.py [n_configs] [seed]"""
import os, sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np, torch
import oracle_py as O
from meteor_demod_amd import DemodConfig, Demodulator, synth

n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad, kinds = [], {}
t0 = time.time()
for ci in range(n_cfg):
    symrate = int(rng.choice([72000, 72000, 80000, 36000, 144000]))
    import os
    osf_list = [float(v) for v in os.environ["FUZZ_OSF"].split(",")] if os.environ.get("FUZZ_OSF") else \
        [0.6, 0.9, 1.0, 1.3, 2.0, 2.5, 2.875, 3.1944, 3.6, 4.0, 6.0, 9.0, 13.9, 14.9, 20.0]
    osf = float(rng.choice(osf_list))
    samplerate = int(symrate * osf * (1 + rng.uniform(-0.01, 0.01)))
    oqpsk = bool(rng.random() < 0.35)
    cfg = DemodConfig(samplerate=samplerate, symrate=symrate, oqpsk=oqpsk,
                      rrc_order=int(rng.choice([4, 8, 16, 17, 24, 32, 33, 40, 48, 63, 64, 65, 80])),
                      interp_factor=int(rng.choice([1, 2, 3, 4, 5, 6, 8, 10, 16, 29, 32, 64])),
                      pll_bw=float(rng.choice([0.01, 0.5, 1.0, 2.0, 5.0, 100.0, 3000.0])),
                      freq_max=float(rng.choice([-1.0, 0.0, 0.001, 0.05, 0.3, 1.5])),
                      bps=int(rng.choice([8, 16, 16, 16, 32])))
    if os.environ.get("FUZZ_HYB"):          # float input with 66..129 taps only: the v3 hybrid window (VGPRs + AccVGPRs)
        cfg = DemodConfig(samplerate=cfg.samplerate, symrate=cfg.symrate, oqpsk=cfg.oqpsk, rrc_order=int(rng.choice([8, 16, 24, 32, 33, 40, 48, 63, 64])),
                          interp_factor=int(rng.choice([1, 2, 3, 4, 5, 6, 8, 10, 16])), pll_bw=cfg.pll_bw, freq_max=cfg.freq_max, bps=32)
    # the reference divides 0/0 when an RRC tap falls on t = 1/(4*alpha): undefined there, skip
    if cfg.samplerate * (2 if cfg.oqpsk else 1) < cfg.symrate * 0.25:
        continue            # less than a quarter of a sample per firing: refused by mdemod_create (untested region)
    try:
        ost_probe = O.OracleStream(cfg)
        if not np.isfinite(ost_probe.rrc_table()).all():
            continue
    except Exception:
        continue
    ns = int(rng.integers(1, 70)) if not os.environ.get("FUZZ_HYB") else int(rng.choice([1, 63, 64, 65, 255, 256, 257, 300, 513]))
    blocks = [int(rng.choice([0, 1, 7, 64, 129, 1000, 4097, 9000])) for _ in range(int(rng.integers(1, 4)))]
    blocks.append(int(rng.integers(2000, 12000)))
    total = sum(blocks)
    rms = {8: 50.0, 16: 5000.0, 32: 0.7}[cfg.bps]
    streams = [synth.make_stream(int(rng.integers(1 << 30)), cfg.samplerate, cfg.symrate, f0_hz=float(rng.uniform(-2000, 2000)),
                                 clock_ppm=float(rng.uniform(-40, 40)), esn0_db=float(rng.uniform(5, 25)), rms=rms,
                                 dc=(rms / 150, -rms / 250), oqpsk=cfg.oqpsk, fmt=cfg.bps) for _ in range(min(ns, 6))]
    iqs = [synth.generate_host(s, total) for s in streams]
    # pathological inputs on some streams: silence, full-scale noise, a pure tone, a DC level
    DTs = {8: np.uint8, 16: np.int16, 32: np.float32}
    full = {8: 127, 16: 32767, 32: 1.0}[cfg.bps]
    zero = 128 if cfg.bps == 8 else 0
    for i in range(len(iqs)):
        kind = int(rng.integers(0, 8))
        if kind == 0:
            iqs[i] = np.full((total, 2), zero, dtype=DTs[cfg.bps])
        elif kind == 1:
            v = rng.choice([-full, full], size=(total, 2))
            iqs[i] = (v + zero).astype(DTs[cfg.bps]) if cfg.bps != 32 else v.astype(np.float32)
        elif kind == 2:
            ph = 2 * np.pi * float(rng.uniform(-0.2, 0.2)) * np.arange(total)
            v = np.stack([np.cos(ph), np.sin(ph)], axis=1) * full * 0.9
            iqs[i] = (np.round(v) + zero).astype(DTs[cfg.bps]) if cfg.bps != 32 else v.astype(np.float32)
        elif kind == 3:
            iqs[i] = np.full((total, 2), zero + (full // 2 if cfg.bps != 32 else 0.5), dtype=DTs[cfg.bps])
        elif kind == 4:                                   # silence, then a full-scale burst (AGC gain high when it arrives)
            v = rng.choice([-full, full], size=(total, 2)); v[: int(rng.integers(100, max(101, total // 2)))] = 0
            iqs[i] = (v + zero).astype(DTs[cfg.bps]) if cfg.bps != 32 else v.astype(np.float32)
        elif kind == 5 and cfg.bps == 32:                 # float input far outside [-1, 1]
            iqs[i] = (iqs[i] * float(rng.choice([1e3, 1e6, 1e-6]))).astype(np.float32)
    if len(sys.argv) > 3 and ci != int(sys.argv[3]):
        continue
    print(f"cfg {ci}: {cfg} ns={ns} blocks={blocks}", flush=True)
    if len(sys.argv) > 3:
        for i, a in enumerate(iqs):
            print("   input", i, a.dtype, a.shape, "finite", bool(np.isfinite(a).all()), "absmax", float(np.abs(a).max()))
    tc = time.time()
    import os
    # every second configuration through the latency kernel (one stream per wave), and a third of those switch kernels between
    # the chained blocks (same context, same state arrays: the kernels must hand over to each other exactly)
    os.environ["MDEMOD_LAT"] = str(ci & 1)
    try:
        with Demodulator(cfg, ns) as d:
            kinds[d.kernel_name] = kinds.get(d.kernel_name, 0) + 1
            got = [[] for _ in range(ns)]
            pos = 0
            for bi, b in enumerate(blocks):
                if (ci & 1) and ci % 3 == 0:
                    os.environ["MDEMOD_LAT"] = str((bi + 1) & 1)
                if b == 0:
                    x = torch.zeros((ns, 1, 2), dtype=torch.from_numpy(iqs[0][:1]).dtype, device="cuda")
                    soft = d.process(x, n_samples=0)
                else:
                    soft = d.process(torch.from_numpy(np.stack([iqs[i % len(iqs)][pos:pos + b] for i in range(ns)])).cuda())
                torch.cuda.synchronize()
                cnt = d.symbol_counts()
                for i in range(ns):
                    got[i].append(soft[i, : int(cnt[i])].cpu().numpy())
                pos += b
            st = d.status()
            for i in range(ns):
                ost = O.OracleStream(cfg)
                want = ost.run(iqs[i % len(iqs)])[0]
                g = np.concatenate(got[i])
                ok = st[i].overflow == 0 and g.shape == want.shape and np.array_equal(g, want) and np.float32(st[i].pll_freq) == np.float32(ost.state.pll_freq) \
                    and st[i].locked == ost.state.locked and np.float32(st[i].gain) == np.float32(ost.state.gain)
                if not ok:
                    k = min(len(g), len(want)); diff = np.flatnonzero((g[:k] != want[:k]).any(axis=1))
                    why = (f"overflow {st[i].overflow}, len {g.shape} vs {want.shape}, first diff {diff[:3]}, ndiff {len(diff)}, freq {st[i].pll_freq!r} vs {ost.state.pll_freq!r}, "
                           f"locked {st[i].locked} vs {ost.state.locked}, gain {st[i].gain!r} vs {ost.state.gain!r}, ns {ns}")
                    bad.append((ci, i, cfg, d.kernel_name, blocks, why))
                    if len(sys.argv) > 3:
                        import os
                        os.makedirs("gpurun_out", exist_ok=True)
                        np.savez_compressed("gpurun_out/fuzz_fail.npz", iq=iqs[i % len(iqs)], blocks=np.asarray(blocks))
                        tr = O.OracleStream(cfg).run(iqs[i % len(iqs)], True)[1]
                        f = int(diff[0]) if len(diff) else k
                        print("   per-block gpu counts", [len(p) for p in got[i]], "oracle sample idx near diff", tr["sample_index"][max(0, f - 3): f + 6])
                        print("   gpu", g[max(0, f - 3): f + 6].tolist()); print("   orc", want[max(0, f - 3): f + 6].tolist())
                        a = iqs[i % len(iqs)]; print("   input absmax by block", [int(np.abs(a[sum(blocks[:j]): sum(blocks[:j + 1])].astype(np.int64)).max()) if blocks[j] else 0 for j in range(len(blocks))])
                    break
    except Exception as e:
        # (tables that fit no kernel's LDS used to be refused by mdemod_create; the v1 kernel reads them from global memory since r03)
        if "mdemod_create: error -1" in repr(e) and cfg.interp_factor >= 16 and cfg.interp_factor * cfg.taps > 2300:
            kinds["refused: table too large for LDS"] = kinds.get("refused: table too large for LDS", 0) + 1
        else:
            bad.append((ci, -1, cfg, repr(e), blocks))
    print(f"   {time.time()-tc:.2f} s, failures so far {len(bad)}", flush=True)
print(f"{n_cfg} configs in {time.time()-t0:.0f} s; kernels used: {kinds}; failures: {len(bad)}")
for b in bad[:10]:
    print("  BAD", b)
