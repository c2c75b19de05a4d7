from .mappers import AbstractActionMapper  # noqa: F401
