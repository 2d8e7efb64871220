from .image_prep import ImagePrep  # noqa: F401
