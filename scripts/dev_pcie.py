"""Host <-> device copy rates on this box: pageable, page-locked (torch) and registered (hipHostRegister) memory."""
import ctypes as C, time
import numpy as np, torch
n = 3840 * 2160 * 16
dev = torch.empty(n, dtype=torch.uint8, device="cuda")
def rate(host, label):
    for _ in range(2):
        dev.copy_(host, non_blocking=True); torch.cuda.synchronize()
    t = time.perf_counter(); dev.copy_(host, non_blocking=True); torch.cuda.synchronize(); up = time.perf_counter() - t
    t = time.perf_counter(); host.copy_(dev, non_blocking=True); torch.cuda.synchronize(); down = time.perf_counter() - t
    print(f"{label}: upload {n / up / 1e9:.1f} GB/s ({up * 1e3:.2f} ms), download {n / down / 1e9:.1f} GB/s ({down * 1e3:.2f} ms)", flush=True)
rate(torch.empty(n, dtype=torch.uint8), "pageable")
rate(torch.empty(n, dtype=torch.uint8).pin_memory(), "page-locked")
hip = C.CDLL("libamdhip64.so")
arr = np.zeros(n, np.uint8)
t = time.perf_counter(); rc = hip.hipHostRegister(C.c_void_p(arr.ctypes.data), C.c_size_t(n), 0); print("hipHostRegister rc", rc, f"{(time.perf_counter() - t) * 1e3:.1f} ms")
rate(torch.from_numpy(arr), "registered numpy")
raw = np.zeros(n + 8192, np.uint8)
off = (-raw.ctypes.data) % 4096
al = raw[off:off + n]
for flags in (0, 1, 2):
    rc = hip.hipHostRegister(C.c_void_p(al.ctypes.data), C.c_size_t(n), flags); print("aligned ptr, flags", flags, "rc", rc)
    if rc == 0:
        rate(torch.from_numpy(al), f"registered (aligned, flags {flags})")
        hip.hipHostUnregister(C.c_void_p(al.ctypes.data))
un = raw[off + 64:off + 64 + n]
rc = hip.hipHostRegister(C.c_void_p(un.ctypes.data), C.c_size_t(n), 0); print("unaligned ptr rc", rc)
# registering the enclosing pages of an unaligned buffer
base = un.ctypes.data & ~4095
span = ((un.ctypes.data + n + 4095) & ~4095) - base
rc = hip.hipHostRegister(C.c_void_p(base), C.c_size_t(span), 0); print("enclosing pages rc", rc)
if rc == 0:
    rate(torch.from_numpy(un), "registered enclosing pages")
