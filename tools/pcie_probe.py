"""PCIe link probe: H2D alone, D2H alone, both at once on two streams (SDMA), and D2H by a copy KERNEL writing pinned
host memory while SDMA does H2D.  Tells whether the two directions overlap on this box."""
import time
import torch

N = 1 << 30   # 1 GiB each way
h_src = torch.empty(N, dtype=torch.uint8, pin_memory=True); h_src.fill_(1)
h_dst = torch.empty(N, dtype=torch.uint8, pin_memory=True)
d_in = torch.empty(N, dtype=torch.uint8, device="cuda")
d_out = torch.ones(N, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def run(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def h2d():
    with torch.cuda.stream(s1):
        d_in.copy_(h_src, non_blocking=True)


def d2h():
    with torch.cuda.stream(s2):
        h_dst.copy_(d_out, non_blocking=True)


def both():
    h2d(); d2h()


t = run(h2d); print(f"H2D alone      : {N / t / 1e9:6.1f} GB/s")
t = run(d2h); print(f"D2H alone      : {N / t / 1e9:6.1f} GB/s")
t = run(both); print(f"H2D + D2H SDMA : {2 * N / t / 1e9:6.1f} GB/s aggregate ({t * 1e3:.1f} ms for 1 GiB each way)")

# D2H through a kernel: map the pinned destination into the device address space and let an elementwise kernel write it
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
dp = ctypes.c_void_p()
rc = hip.hipHostGetDevicePointer(ctypes.byref(dp), ctypes.c_void_p(h_dst.data_ptr()), 0)
print("hipHostGetDevicePointer rc", rc, hex(dp.value or 0), hex(h_dst.data_ptr()))
