"""dbw_amd -- MI355X-native differentiable superquadric renderer behind the rendering call of
monniert/differentiable-blocksworld's src/model.

    from dbw_amd import create_model, Renderer, DifferentiableBlocksWorld
"""
from . import _lib, mesh, ops, structures                                  # noqa: F401
from .dbw import DifferentiableBlocksWorld                                 # noqa: F401
from .renderer import Renderer                                             # noqa: F401
from .structures import Meshes, TexturesUV, join_meshes_as_scene          # noqa: F401


def create_model(cfg, img_size, **kwargs):
    """src/model/__init__.py:12-17: cfg['model'] = {name: 'dbw', mesh:, renderer:, rend_optim:, loss:}."""
    kwargs = dict(cfg['model'])
    name = kwargs.pop('name')
    if name != 'dbw':
        raise KeyError(name)
    return DifferentiableBlocksWorld(img_size, **kwargs)
