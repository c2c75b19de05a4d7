from .mappers import DerivativeActionMapper  # noqa: F401
