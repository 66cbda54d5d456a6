"""Which Python lines of the training step make dtype / layout copies and zero fills, by bytes (one eager step; torch's own
internal copies, e.g. inside autograd's accumulation, do not pass through these entry points and are not listed)."""
import collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench

MIN_BYTES = int(os.environ.get("MIN_BYTES", 1 << 20))     # MIN_BYTES=0: every call, listed by call count (the tiny launches)
sites = collections.Counter()
calls = collections.Counter()

def where():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "/pdm_ssd_amd/" in fr.filename or fr.filename.endswith("bench.py"):
            return f"{os.path.basename(fr.filename)}:{fr.lineno}"
    return "?"

def wrap(obj, name, size_of):
    orig = getattr(obj, name)
    def f(*a, **k):
        out = orig(*a, **k)
        try:
            nbytes, changed = size_of(a, k, out)
            if changed and nbytes >= MIN_BYTES:
                key = f"{name:12s} {where()}"
                sites[key] += nbytes; calls[key] += 1
        except Exception:
            pass
        return out
    setattr(obj, name, f)

T = torch.Tensor
wrap(T, "to", lambda a, k, o: (o.numel() * o.element_size(), o.data_ptr() != a[0].data_ptr()))
wrap(T, "float", lambda a, k, o: (o.numel() * o.element_size(), o.data_ptr() != a[0].data_ptr()))
wrap(T, "bfloat16", lambda a, k, o: (o.numel() * o.element_size(), o.data_ptr() != a[0].data_ptr()))
wrap(T, "contiguous", lambda a, k, o: (o.numel() * o.element_size(), o.data_ptr() != a[0].data_ptr()))
wrap(T, "copy_", lambda a, k, o: (o.numel() * o.element_size(), True))
wrap(T, "clone", lambda a, k, o: (o.numel() * o.element_size(), True))
wrap(T, "zero_", lambda a, k, o: (o.numel() * o.element_size(), True))
wrap(torch, "zeros", lambda a, k, o: (o.numel() * o.element_size(), True))
wrap(torch, "zeros_like", lambda a, k, o: (o.numel() * o.element_size(), True))
wrap(torch, "cat", lambda a, k, o: (o.numel() * o.element_size(), True))
wrap(T, "new_zeros", lambda a, k, o: (o.numel() * o.element_size(), True))
wrap(T, "fill_", lambda a, k, o: (o.numel() * o.element_size(), True))
wrap(torch, "full", lambda a, k, o: (o.numel() * o.element_size(), True))
wrap(torch, "ones", lambda a, k, o: (o.numel() * o.element_size(), True))
wrap(T, "new_full", lambda a, k, o: (o.numel() * o.element_size(), True))
wrap(T, "new_ones", lambda a, k, o: (o.numel() * o.element_size(), True))
wrap(T, "reshape", lambda a, k, o: (o.numel() * o.element_size(), o.data_ptr() != a[0].data_ptr()))

sys.argv = ["bench.py", "--train", "--steps", "1", "--warmup", "2", "--no-cpu-baseline"]
sites_reset = [False]
orig_tb = bench.train_bench
def tb(*a, **k):
    return orig_tb(*a, **k)
bench.main()
print("MB per 3 steps (2 warm-up + 1 timed), calls, site")
order = sites.most_common(40) if MIN_BYTES else sorted(sites.items(), key=lambda kv: -calls[kv[0]])[:70]
for k, v in order:
    print(f"{v / 1e6:10.1f} MB {calls[k]:5d}  {k}")
