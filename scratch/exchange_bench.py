"""Duration of the node-exchange kernel on a one-rank exchange (no xGMI: launch + the trips to the uncached window),
back to back and behind a kernel that leaves the L2 full of dirty lines."""
import ctypes as C, importlib, sys, os, time, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
_lib = importlib.import_module('3d_reconstruction_amd._lib')
lib = _lib.load()
h = C.c_void_p()
_lib.check(lib.ba_exchange_create(0, 1, 131072, C.byref(h)))
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for n in (17210, 12001, 1024):
    t = torch.ones(n, dtype=torch.float64, device='cuda:0')
    big = torch.zeros(12_000_000, dtype=torch.float64, device='cuda:0')
    for label, dirty in (('back to back', False), ('behind a 96 MB fill', True)):
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(500):
                if dirty:
                    big.add_(1.0)
                _lib.check(lib.ba_exchange_sum(h, C.c_void_p(t.data_ptr()), n, st))
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 500
        base = 0.0
        if dirty:
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(500):
                big.add_(1.0)
            torch.cuda.synchronize(); base = (time.perf_counter() - t0) / 500
        print(f'n={n:6d} {label:22s}: {(dt - base) * 1e6:6.2f} us per call (sum {float(t[0]):.0f})', flush=True)
        t.fill_(1.0)
_lib.check(lib.ba_exchange_status(h))
lib.ba_exchange_destroy(h)
