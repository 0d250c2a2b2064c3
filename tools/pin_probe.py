import ctypes as C, numpy as np, torch
torch.cuda.init()
# the runtime this process already runs on (torch's copy), by the path it is mapped from - by NAME the system's copy would come in as a second runtime
hip = C.CDLL(sorted({l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64.so" in l})[0])
class Attr(C.Structure):
    _fields_ = [("type", C.c_int), ("device", C.c_int), ("devicePointer", C.c_void_p), ("hostPointer", C.c_void_p), ("isManaged", C.c_int), ("allocationFlags", C.c_uint)]
raw = np.zeros(1 << 24, dtype=np.uint8)
base = raw.ctypes.data + (-raw.ctypes.data % 4096)
size = 1 << 23
half = size // 2
print("register first half:", hip.hipHostRegister(C.c_void_p(base), C.c_size_t(half), 0))
for name, p in (("base", base), ("half-1", base + half - 1), ("half", base + half), ("half+4096", base + half + 4096), ("end-1", base + size - 1)):
    a = Attr(); rc = hip.hipPointerGetAttributes(C.byref(a), C.c_void_p(p))
    ab = C.c_void_p(); asz = C.c_size_t()
    rc2 = hip.hipMemGetAddressRange(C.byref(ab), C.byref(asz), C.c_void_p(p))
    print(name, "attrs rc", rc, "type", a.type, "host", hex(a.hostPointer or 0), "dev", hex(a.devicePointer or 0), "| range rc", rc2, hex(ab.value or 0), asz.value, "(base", hex(base), "half", half, ")")
    hip.hipGetLastError()
print("register whole:", hip.hipHostRegister(C.c_void_p(base), C.c_size_t(size), 0)); hip.hipGetLastError()
print("register second half:", hip.hipHostRegister(C.c_void_p(base + half), C.c_size_t(half), 0)); hip.hipGetLastError()
for name, p in (("base", base), ("half", base + half)):
    ab = C.c_void_p(); asz = C.c_size_t()
    rc2 = hip.hipMemGetAddressRange(C.byref(ab), C.byref(asz), C.c_void_p(p))
    print(name, "range rc", rc2, hex(ab.value or 0), asz.value)
t = torch.empty(1 << 20, dtype=torch.uint8).pin_memory()
ab = C.c_void_p(); asz = C.c_size_t()
print("torch pinned:", hip.hipMemGetAddressRange(C.byref(ab), C.byref(asz), C.c_void_p(t.data_ptr())), hex(ab.value or 0), asz.value, hex(t.data_ptr()))
