from .conversion import convert, DimensionMismatch, MissingConversionInputs  # noqa: F401
