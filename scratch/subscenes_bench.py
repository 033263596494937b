"""fit_subscenes_GD's device path at a large size: subscenes_bench.py [N] [P] [per_point]"""
import importlib, sys, os, time, numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
if os.environ.get('BA_LIB'):
    importlib.import_module('3d_reconstruction_amd._lib').LIB_PATH = os.path.abspath(os.environ['BA_LIB'])
ba = importlib.import_module('3d_reconstruction_amd')
sub = importlib.import_module('3d_reconstruction_amd.subscenes')
N = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000
P = int(sys.argv[2]) if len(sys.argv) > 2 else 50
per = int(sys.argv[3]) if len(sys.argv) > 3 else 3
rng = np.random.default_rng(0)
Xw = rng.normal(size=(N, 3))
X = np.full((N, P, 3), np.nan)
start = rng.integers(0, P - per + 1, size=N)
S = rng.uniform(0.5, 2.0, size=P); T = rng.normal(size=(P, 3)); Psi = rng.uniform(-0.5, 0.5, size=(P, 3))
for k in range(per):
    j = start + k
    X[np.arange(N), j] = Xw * S[j, None] + T[j] + 0.01 * rng.normal(size=(N, 3))
st = sub.SubsceneSet(X, np.ones(P), np.zeros((P, 3)), np.zeros((P, 3)), 'cuda:0')
n = st.n
for _ in range(5):
    st.eval()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(50):
    st.eval()
b.record(); torch.cuda.synchronize()
t = a.elapsed_time(b) / 50 * 1e3
byts = n * 28 + 8 * (N + 1)
print(f'N={N} P={P} entries={n}: eval (table + sums + pass) {t:.1f} us per call, {byts / t / 1e6:.3f} TB/s by {byts / 1e6:.1f} MB of entries', flush=True)
if os.environ.get('SUB_EVAL_ONLY'):
    sys.exit(0)
t0 = time.perf_counter()
out = ba.fit_subscenes_GD(X, 1e-2, 200, optimizer='Adam', print_step=10**9) if 'print_step' in ba.fit_subscenes_GD.__code__.co_varnames else ba.fit_subscenes_GD(X, 1e-2, 200)
torch.cuda.synchronize()
print(f'fit_subscenes_GD, 200 iterations incl. set-up: {1e3 * (time.perf_counter() - t0):.1f} ms')
