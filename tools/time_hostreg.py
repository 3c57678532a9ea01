#!/usr/bin/env python3
"""hipHostRegister / hipHostUnregister cost and pageable vs registered H2D copy rate (decides how the host-buffer
batch entry points stage large inputs)."""
import ctypes as C, time, numpy as np, torch
hip = C.CDLL("libamdhip64.so")
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
hip.hipHostUnregister.argtypes = [C.c_void_p]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
torch.cuda.init()
for mb in (32, 128, 512):
    n = mb << 20
    a = np.random.randint(0, 255, n, dtype=np.uint8)
    d = torch.empty(n, dtype=torch.uint8, device="cuda")
    p = a.ctypes.data
    hip.hipMemcpy(d.data_ptr(), p, n, 1); torch.cuda.synchronize()
    t0 = time.perf_counter(); hip.hipMemcpy(d.data_ptr(), p, n, 1); torch.cuda.synchronize(); tp = time.perf_counter() - t0
    t0 = time.perf_counter(); rc = hip.hipHostRegister(p, n, 0); tr = time.perf_counter() - t0
    t0 = time.perf_counter(); hip.hipMemcpy(d.data_ptr(), p, n, 1); torch.cuda.synchronize(); tc = time.perf_counter() - t0
    t0 = time.perf_counter(); hip.hipHostUnregister(p); tu = time.perf_counter() - t0
    print("%4d MB: pageable copy %.2f ms (%.1f GB/s) | register %.2f ms rc=%d | registered copy %.2f ms (%.1f GB/s) | unregister %.2f ms"
          % (mb, tp * 1e3, n / tp / 1e9, tr * 1e3, rc, tc * 1e3, n / tc / 1e9, tu * 1e3))
