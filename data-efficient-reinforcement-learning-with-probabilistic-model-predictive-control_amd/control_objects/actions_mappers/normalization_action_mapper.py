from .mappers import NormalizationActionMapper  # noqa: F401
