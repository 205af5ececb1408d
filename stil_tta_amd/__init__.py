"""stil_tta_amd: MI355X-native (gfx950) implementation of the STiL semi-supervised training step.

Importing the package does not need a GPU; running any operator does (there is no CPU fallback).
"""
from ._lib import lib, build, LIB_PATH  # noqa: F401
from .stil_model import STiLModel  # noqa: F401
from .mmatch import CoTraining, MMatch  # noqa: F401
from .match import CoMatch, FreeMatch, SimMatch  # noqa: F401



class SemiDisCoPseudoSmooth(STiLModel):
    """models/Disentangle/STiLModel_SAINT.py:29 -- the SAINT-encoder variant.  trainers/evaluate.py:146 imports it as
    `STiLModel` from that module although the class is called SemiDisCoPseudoSmooth; both spellings work here."""

    def __init__(self, hparams):
        hp = dict(hparams) if isinstance(hparams, dict) else dict(getattr(hparams, "__dict__", None) or {k: hparams[k] for k in hparams.keys()})
        hp["tabular_encoder"] = "saint"
        super().__init__(hp)


def create_model(hparams):
    """The `algorithm_name` dispatch of trainers/evaluate.py:142-166: the module the reference would build for these hparams."""
    name = hparams["algorithm_name"] if isinstance(hparams, dict) else getattr(hparams, "algorithm_name", None)
    table = {"STiL": STiLModel, "STiL_SAINT": SemiDisCoPseudoSmooth, "MMatch": MMatch, "SimMatch": SimMatch, "CoMatch": CoMatch,
             "FreeMatch": FreeMatch, "CoTrain_Pseudo": CoTraining, "CoTrain_Pseudo_SAINT": CoTraining}
    if name not in table:
        raise ValueError(f"Algorithm name not found: {name!r} (one of {sorted(table)})")
    return table[name](hparams)


__all__ = ["create_model", "STiLModel", "SemiDisCoPseudoSmooth", "MMatch", "CoTraining", "CoMatch", "SimMatch", "FreeMatch", "lib", "build", "LIB_PATH"]
