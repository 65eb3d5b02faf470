# Same public name as africanus/dft/__init__.py:3 (hot-path subset).
from .kernels import im_to_vis, im_to_vis_chi2, vis_to_im, im_to_vis_from_model, set_mode, get_mode, mode  # noqa: F401
