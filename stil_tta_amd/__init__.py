"""stil_tta_amd: MI355X-native (gfx950) implementation of the STiL semi-supervised training step.

Importing the package does not need a GPU; running any operator does (there is no CPU fallback).
"""
from ._lib import lib, build, LIB_PATH  # noqa: F401
from .stil_model import STiLModel  # noqa: F401

# trainers/evaluate.py:146 imports STiLModel from STiLModel_SAINT although the class there is called
# SemiDisCoPseudoSmooth; both names are exported once the SAINT variant lands.
__all__ = ["STiLModel", "lib", "build", "LIB_PATH"]
