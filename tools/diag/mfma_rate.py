import ctypes, os
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmfma_rate.so"))
lib.run.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
iters = 4000
for var in (0, 1, 2):
    for wg in (1, 2, 3):
        ms = ctypes.c_float()
        lib.run(var, wg, iters, ctypes.byref(ms))
        flops = 256 * wg * 4 * iters * 16 * 2048
        print(f"variant {var} ({['regs only','+ fragment copies','+ L1-resident loads'][var]}), {wg} wave(s)/SIMD: {ms.value:.3f} ms  {flops/ms.value/1e9:.1f} TFLOP/s")
