"""Dense Cholesky solve at n unknowns with every tile computed (bench.dense_cholesky_rate): chol_dense.py [n] [repeats]"""
import importlib, os, sys, json
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import torch
import bench
ba = importlib.import_module('3d_reconstruction_amd')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30001
rep = int(sys.argv[2]) if len(sys.argv) > 2 else 2
print(json.dumps(bench.dense_cholesky_rate(ba, torch.device('cuda:0'), n, rep)), flush=True)
