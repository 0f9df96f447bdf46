"""Enumerations of the flavor-ratio analysis that the hot path consumes.

Same member names and integer values as the reference (golemflavor/enums.py:17-63) so
scripts written against it keep working; the integer values are also what the C ABI
(`include/golemflavor_hip.h`) uses for `gf_prior_kind` / `gf_texture`.
"""
from enum import Enum

__all__ = [
    "DataType", "Likelihood", "ParamTag", "PriorsCateg", "MCMCSeedType",
    "StatCateg", "SteeringCateg", "Texture", "str_enum",
]


def str_enum(x):
    """'Texture.OET' -> 'OET' (golemflavor/enums.py:13-14)."""
    return str(x).rsplit(".", 1)[-1]


DataType = Enum("DataType", ["REAL", "ASIMOV", "REALISATION"])
Likelihood = Enum("Likelihood", ["GOLEMFIT", "GF_FREQ"])
ParamTag = Enum(
    "ParamTag",
    ["NUISANCE", "SM_ANGLES", "MMANGLES", "SCALE", "SRCANGLES", "BESTFIT", "NONE"],
)
PriorsCateg = Enum("PriorsCateg", ["UNIFORM", "GAUSSIAN", "LIMITEDGAUSS"])
MCMCSeedType = Enum("MCMCSeedType", ["UNIFORM", "GAUSSIAN"])
StatCateg = Enum("StatCateg", ["BAYESIAN", "FREQUENTIST"])
SteeringCateg = Enum("SteeringCateg", ["P2_0", "P2_1"])
Texture = Enum("Texture", ["OEU", "OET", "OUT", "NONE"])
