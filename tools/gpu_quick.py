"""Quick GPU sanity run: primitives, parity on a few cases, rough throughput."""
import sys, time, numpy as np, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import oracle_py as O
from golden_cases import BY_NAME
from meteor_demod_amd import DemodConfig, Demodulator, synth

def parity(name, nrep=3):
    case = BY_NAME[name]; iq = case.generate()
    so, to, ev = O.oracle_demod(case.cfg, iq, True)
    with Demodulator(case.cfg, n_streams=nrep) as d:
        x = torch.from_numpy(np.stack([iq]*nrep)).cuda()
        t=time.time(); soft = d.process(x); torch.cuda.synchronize(); dt=time.time()-t
        st = d.status()
        ok = True
        for s in range(nrep):
            m = st[s].symbols_this_call
            g = soft[s,:m].cpu().numpy()
            eq = g.shape == so.shape and np.array_equal(g, so)
            ok &= eq
            if not eq:
                k = min(len(g), len(so)); bad = np.flatnonzero((g[:k]!=so[:k]).any(axis=1))
                print(f"  stream {s}: gpu {g.shape} oracle {so.shape} first bad {bad[:5]} maxdiff {np.abs(g[:k].astype(int)-so[:k].astype(int)).max()}")
        print(f"{name}: parity={ok} sym={st[0].symbols_this_call} first_lock={st[0].first_lock_symbol} (oracle {to['locked'].argmax() if to['locked'].any() else -1}) "
              f"events={d.lock_events(0)} vs {ev} freq={st[0].pll_freq:.6g}/{to[-1]['pll_freq']:.6g} gain={st[0].gain:.6g}/{to[-1]['gain']:.6g} {dt*1e3:.1f} ms")
    return ok

cfg = DemodConfig(samplerate=230000)
with Demodulator(cfg, 1) as d:
    x = np.concatenate([np.linspace(-9, 9, 1<<20, dtype=np.float32), np.random.default_rng(1).uniform(-9,9,1<<18).astype(np.float32)])
    s, c = d.selftest_sincos(x)
    L = O.lib()
    idx = np.random.default_rng(2).integers(0, x.size, 20000)
    so = np.array([L.orc_fast_sin(float(x[i])) for i in idx], dtype=np.float32)
    co = np.array([L.orc_fast_cos(float(x[i])) for i in idx], dtype=np.float32)
    print("sincos parity", np.array_equal(s[idx], so), np.array_equal(c[idx], co))
    xy = np.random.default_rng(3).normal(0, 300, (1<<20, 2)).astype(np.float32)
    h = d.selftest_hypot(xy)
    ref = np.sqrt(xy[:,0].astype(np.float64)**2 + xy[:,1].astype(np.float64)**2).astype(np.float32)
    print("hypot parity", np.array_equal(h, ref), int((h!=ref).sum()))

allok = True
for n in ["c1_short", "c3_short", "u8_short", "f32_short", "odd_cfg", "c1_fade", "c4_os8"]:
    try:
        allok &= parity(n)
    except Exception as e:
        print(n, "EXC", repr(e)); allok = False
print("ALL PARITY", allok)

# throughput probe
for ns, n in [(65536, 16384), (262144, 8192)]:
    st = [synth.make_stream(100+i, 230000, 72000, f0_hz=(i%7-3)*400.0, clock_ppm=(i%11-5)*8.0) for i in range(64)]
    base = synth.generate_device(st, n)           # 64 distinct streams
    x = base.repeat(ns//64, 1, 1).contiguous()
    with Demodulator(cfg, ns) as d:
        soft = d.process(x); torch.cuda.synchronize()
        d.reset()
        t=time.time(); d.process(x, soft=soft); torch.cuda.synchronize(); dt=time.time()-t
        print(f"streams={ns} n={n}: {dt*1e3:.2f} ms -> {ns*n/dt/1e9:.2f} GS/s")
