"""achievable pure-write bandwidth on this MI355X: hipMemsetAsync of 8 GiB, timed with HIP events (calibrates k_latent_to_w)"""
import ctypes as C, time
hip = C.CDLL("/opt/rocm/lib/libamdhip64.so")
n = 8 << 30
p = C.c_void_p()
assert hip.hipMalloc(C.byref(p), C.c_size_t(n)) == 0
e0, e1 = C.c_void_p(), C.c_void_p()
hip.hipEventCreate(C.byref(e0)); hip.hipEventCreate(C.byref(e1))
for rep in range(3):
    hip.hipEventRecord(e0, None)
    for _ in range(4):
        hip.hipMemsetAsync(p, 0, C.c_size_t(n), None)
    hip.hipEventRecord(e1, None); hip.hipEventSynchronize(e1)
    ms = C.c_float(); hip.hipEventElapsedTime(C.byref(ms), e0, e1)
    print("hipMemset 4 x 8 GiB: %.3f ms -> %.0f GB/s" % (ms.value, 4 * n / ms.value / 1e6))
