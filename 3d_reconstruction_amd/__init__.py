"""MI355X-native bundle adjustment behind the entry points of OlegRokin/3D_reconstruction's
`modules/pix2world.py` (fit_scene path).  The directory name starts with a digit; import it as

    ba = importlib.import_module('3d_reconstruction_amd')

The compute path needs libba_mi355x.so (built in-tree by `__graft_entry__.build()`) and a GPU;
there is no CPU fallback."""
from .pix2world import (align_scene, error, fit_scene, fit_scene_sharded, fit_scene_sparse, gd_adam, gd_func_dict, get_C_Theta_from_X,  # noqa: F401
                        get_H_diff_matrix, get_rotations, get_X_from_C_Theta, normalize_scene, ryxz, sgd, transform,
                        validate_scene_parameters)
from .fit_loop import SceneFit  # noqa: F401
from .subscenes import fit_subscenes_GD  # noqa: F401
from . import engine, pix2world, scenes, sharding, subscenes, tracks  # noqa: F401

__version__ = '0.1.0'
