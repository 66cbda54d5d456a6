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

# the same core with random operands, at the length of the point-head launch (~1 ms) and 20x that: the rate a real
# kernel can be held against (the nominal 157.3 TFLOP/s assumes 2.4 GHz under load)
lib.run2.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_float), ctypes.c_int]
for rnd in (0, 1):
    for it in (4000, 80000):
        ms = ctypes.c_float()
        lib.run2(0, 2, it, ctypes.byref(ms), rnd)
        flops = 256 * 2 * 4 * it * 16 * 2048
        print(f"regs only, 2 waves/SIMD, {'random' if rnd else 'zero'} operands, {it} iterations: {ms.value:.3f} ms  {flops/ms.value/1e9:.1f} TFLOP/s")
