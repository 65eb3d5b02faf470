# Same public names as africanus/rime/__init__.py:3-10 (hot-path subset).
from .phase import phase_delay  # noqa: F401
from .predict import predict_vis, apply_gains  # noqa: F401
from .fast_beam_cubes import beam_cube_dde, freq_grid_interp  # noqa: F401
from .fused import fused_predict_vis, fused_plan, cached_plan  # noqa: F401
from .wsclean_predict import wsclean_predict  # noqa: F401
from .feeds import feed_rotation  # noqa: F401
