import os, sys, ctypes as C, numpy as np, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
from meteor_demod_amd import DemodConfig, Demodulator, synth, _capi
cfg = DemodConfig(samplerate=230000)
T, L = 393216, 64
rec = synth.make_stream(1000, 230000, 72000, f0_hz=1200.0)
one = synth.generate_device([rec], L)
x = one.expand(T, L, 2)
d = Demodulator(cfg, T)
cap = d.max_symbols(L)
soft = torch.zeros((T, cap, 2), dtype=torch.int8, device="cuda")
d.process(x, soft=soft); torch.cuda.synchronize()
neq = (soft != soft[:1]).flatten(1).any(dim=1)
idx = torch.nonzero(neq).flatten().cpu().numpy()
print(len(idx), "bad tiles; first", idx[:5])
def dump(stream):
    arr = (_capi.MdemodLockEvent * 32)()
    n = C.c_uint32()
    _capi.check(d._lib.mdemod_get_lock_events(d._ctx, stream, arr, 32, C.byref(n), d._stream()), "ev")
    return np.frombuffer(bytes(arr), dtype=np.float32).copy(), n.value
np.set_printoptions(linewidth=200, precision=6, suppress=False)
g, ng = dump(0)
names = "a bank fire_sub v_cur base yre yim gain t_phase skipF skipL rowoff".split()
print("GOOD tile 0 n=", ng); print(dict(zip(names, g[:12]))); print("win64..79", g[12:44]); print("win[0,8,..]", g[44:52]); print("coef even", g[52:92])
for t in idx[:3]:
    b, nb = dump(int(t))
    print("BAD tile", t, "n=", nb); print(dict(zip(names, b[:12]))); 
    print("  win64..79 same:", np.array_equal(b[12:44], g[12:44]), " win0.. same:", np.array_equal(b[44:52], g[44:52]), " coef same:", np.array_equal(b[52:92], g[52:92]))
    if not np.array_equal(b[52:92], g[52:92]): print("  coef", b[52:92])
    print("  hist re-read [0].x [0].y [1].x [63].x:", b[92:96], " stream,n,block,thread:", b[96:100])
    print("  win[0..7] (re,im):", b[100:116], " win[60..63]:", b[116:124])
