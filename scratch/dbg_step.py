import importlib, sys, numpy as np
sys.path.insert(0, '.')
ba = importlib.import_module('3d_reconstruction_amd')
g = np.load('tests/golden/f2_moscow_ride.npz')
W, H = int(g['W']), int(g['H'])
N, K = int(g['N']), int(g['K'])
sh = ba.engine.Shard(g['obs_i'], g['obs_j'], g['obs_xy'] + 0.5, N, K, W, H, 'cuda:0')
sh.set_params(g['X_0'], g['C_0'], g['Theta_0'], float(g['phi_x_0']))
fit = ba.SceneFit(sh, 1e-3, 10, printer=lambda *a: None)
for s in range(3):
    # params at which gradient of iteration s is taken
    for name, t in (('X', sh.X), ('C', sh.C), ('Theta', sh.Theta), ('phi_x', sh.phi)):
        want = g['tf__step_' + name][s]
        got = t.cpu().numpy().reshape(want.shape)
        d = np.abs(got - want)
        print(s, name, 'max abs diff', d.max(), 'at', np.unravel_index(d.argmax(), d.shape) if d.ndim else 0)
    fit.step()
    nv = sh.nV
    for name, t in (('X', sh.gX), ('C', sh.gC), ('Theta', sh.gTheta), ('phi_x', sh.gphi)):
        pass
    print('E', fit.E[s], g['tf__E'][s])
