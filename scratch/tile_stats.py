"""Tile statistics of the engine (BA_VERBOSE=1 prints them at set-up): tile_stats.py [K] [N]"""
import importlib, os, sys
os.environ['BA_VERBOSE'] = '1'
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
ba = importlib.import_module('3d_reconstruction_amd')
K = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
sc = ba.scenes.orbit_scene(N=N, K=K, track_len=10, seed=0)
sh = ba.engine.Shard(sc['i'], sc['j'], sc['uv'], sc['N_local'], sc['K'], sc['W'], sc['H'], 'cuda:0')
