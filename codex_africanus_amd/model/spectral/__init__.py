# Same public name as africanus/model/spectral/__init__.py.
from .spec_model import spectral_model  # noqa: F401
