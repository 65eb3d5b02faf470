# Same public names as africanus/gridding/wgridder/__init__.py.
from .im2vis import model, plane_precision  # noqa: F401
from .vis2im import dirty, residual, hessian  # noqa: F401
