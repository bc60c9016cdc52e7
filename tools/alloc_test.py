"""diagnostic: hipMalloc / hipHostMalloc latency on an idle device"""
import torch, time, ctypes
torch.cuda.init()
hip = ctypes.CDLL("libamdhip64.so")
for gb in (1, 4, 16):
    p = ctypes.c_void_p()
    t = time.time(); rc = hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(gb << 30)); dt = time.time() - t
    print(f"hipMalloc {gb} GiB: rc={rc} {dt*1e3:.1f} ms", flush=True)
    t = time.time(); hip.hipFree(p); print(f"  hipFree {(time.time()-t)*1e3:.1f} ms", flush=True)
for gb in (1, 2):
    p = ctypes.c_void_p()
    t = time.time(); rc = hip.hipHostMalloc(ctypes.byref(p), ctypes.c_size_t(gb << 30), 0); dt = time.time() - t
    print(f"hipHostMalloc {gb} GiB: rc={rc} {dt*1e3:.1f} ms", flush=True)
    t = time.time(); hip.hipHostFree(p); print(f"  hipHostFree {(time.time()-t)*1e3:.1f} ms", flush=True)
# many mid-size allocations
t = time.time(); ps = []
for i in range(24):
    p = ctypes.c_void_p(); hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(1 << 30)); ps.append(p)
print(f"24 x 1 GiB hipMalloc: {(time.time()-t)*1e3:.1f} ms")
