"""Schur build: matrix-core kernel (with the pair kernel for what it leaves) against the pair kernel alone on scenes of
random shape: schur_fuzz.py [cases] [seed].  Prints the worst relative differences."""
import importlib, os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
ba = importlib.import_module('3d_reconstruction_amd')
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
p = ba.engine.ptr
worst = 0.0
for c in range(cases):
    K = int(rng.integers(8, 300))
    N = int(rng.integers(200, 30000))
    track = int(rng.integers(2, min(17, K)))
    sc = ba.scenes.orbit_scene(N=N, K=K, track_len=track, seed=int(rng.integers(1 << 30)))
    probs = rng.dirichlet([4.0, 2.0, 0.3])
    keep = rng.random(sc['i'].size) < rng.choice([1.0, 0.6, 0.0], size=N, p=probs)[sc['i']]
    i, j, uv = sc['i'][keep], sc['j'][keep], sc['uv'][keep]
    if i.size == 0:
        continue
    tile_obs = rng.choice(['', '256', '1024', '2048'])
    if tile_obs:
        os.environ['BA_TILE_OBS'] = str(tile_obs)
    try:
        sh = ba.engine.Shard(i, j, uv, N, K, sc['W'], sc['H'], 'cuda:0')
    finally:
        os.environ.pop('BA_TILE_OBS', None)
    sh.set_params(sc['X_0'], sc['C_0'], sc['Theta_0'], sc['phi_x_0'])
    sh.eval_jtj()
    first = torch.zeros(sh.ld_red // 64 + 1, dtype=torch.int32, device='cuda:0')
    def build():
        sh.S.zero_()
        ba.engine.check(sh.lib.ba_schur_build_tracked(sh.handle, p(sh.X), 1e-3, p(sh.V), p(sh.Xphi), p(sh.gX), p(sh.B), p(sh.bphi),
                                                      p(sh.hphi), p(sh.gC), p(sh.gTheta), p(sh.gphi), 1, p(sh.S), p(sh.rhs), p(first), 1,
                                                      ba.engine._lib.stream_ptr()))
        n = sh.n_red
        return torch.tril(sh.S[:n, :n]).cpu().numpy(), sh.rhs[:n].cpu().numpy(), sh.Xphi.cpu().numpy().copy(), first.cpu().numpy().copy()
    got = build()
    os.environ['BA_SCHUR_NO_MFMA'] = '1'
    try:
        want = build()
    finally:
        os.environ.pop('BA_SCHUR_NO_MFMA', None)
    errs = [float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)) for a, b in zip(got[:3], want[:3])]
    same_first = bool((got[3] == want[3]).all())
    finite = all(np.isfinite(a).all() for a in got[:3])
    worst = max(worst, max(errs))
    flag = '' if max(errs) < 1e-10 and same_first and finite else '   <-- DIFFERENT'
    print(f'case {c}: N={N} K={K} track={track} tile_obs={tile_obs or "default"} nV={i.size}: S {errs[0]:.1e} rhs {errs[1]:.1e} Xphi {errs[2]:.1e} '
          f'profile {"same" if same_first else "DIFFERENT"}{flag}', flush=True)
    sh.close()
print(f'worst relative difference {worst:.2e}')
