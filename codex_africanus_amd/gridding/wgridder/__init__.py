# Same public name as africanus/gridding/wgridder/__init__.py (the image -> visibility direction).
from .im2vis import model  # noqa: F401
