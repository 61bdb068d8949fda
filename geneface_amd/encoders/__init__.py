from .encoding import get_encoder  # noqa: F401
