# Same public name as africanus/model/shape/__init__.py (hot-path subset).
from .gaussian_shape import gaussian  # noqa: F401
