"""Host-side parameter data model: the boundary *input* of the hot path.

`Param` / `ParamSet` keep the reference's constructor arguments, attribute names and
`from_tag` semantics (golemflavor/param.py:24-214) so that the paramsets the reference's
scripts and notebooks build can be handed to this package unchanged.  Nothing here runs on
the GPU: `golemflavor_amd.descriptor.compile_model` flattens a ParamSet into the POD
`gf_model_desc` once per run, and unlike the reference's callback (llh.py:72-73,
fr.py:410-411) evaluation never writes back into the ParamSet.
"""
from collections.abc import Sequence
from copy import deepcopy

import numpy as np

from .enums import ParamTag, PriorsCateg

__all__ = ["Param", "ParamSet"]


class Param:
    """One scan/nuisance parameter.

    `nominal_value` is frozen at construction (reference: param.py:36); it is the centre
    of the (truncated) Gaussian prior even if `.value` is later overwritten from argv.
    A missing `seed` falls back to `ranges`, a missing prior to UNIFORM, a missing tag to
    ParamTag.NONE (param.py:52-90).
    """

    __slots__ = ("name", "value", "nominal_value", "_prior", "_ranges", "_seed",
                 "std", "_tex", "_tag")

    def __init__(self, name, value, ranges, prior=None, seed=None, std=None,
                 tex=None, tag=None):
        self.name = name
        self.value = value
        self.nominal_value = deepcopy(value)
        self._seed = None
        self.prior = prior
        self.ranges = ranges
        self.seed = seed
        self.std = std
        self.tex = tex
        self.tag = tag

    # -- ranges / seed ------------------------------------------------------
    @property
    def ranges(self):
        return tuple(self._ranges)

    @ranges.setter
    def ranges(self, values):
        self._ranges = list(values)

    @property
    def seed(self):
        return self.ranges if self._seed is None else tuple(self._seed)

    @seed.setter
    def seed(self, values):
        if values is not None:
            self._seed = list(values)

    # -- prior / tag / tex --------------------------------------------------
    @property
    def prior(self):
        return self._prior

    @prior.setter
    def prior(self, value):
        if value is None:
            value = PriorsCateg.UNIFORM
        if not isinstance(value, PriorsCateg):
            raise AssertionError("prior must be a PriorsCateg, got %r" % (value,))
        self._prior = value

    @property
    def tag(self):
        return self._tag

    @tag.setter
    def tag(self, value):
        if value is None:
            value = ParamTag.NONE
        if not isinstance(value, ParamTag):
            raise AssertionError("tag must be a ParamTag, got %r" % (value,))
        self._tag = value

    @property
    def tex(self):
        return "{0}".format(self._tex)

    @tex.setter
    def tex(self, t):
        self._tex = t if t is not None else r"{\rm %s}" % self.name

    def __repr__(self):
        return "Param(%r, value=%r, ranges=%r, prior=%s, tag=%s)" % (
            self.name, self.value, self.ranges, self.prior.name, self.tag.name)


class ParamSet(Sequence):
    """Ordered, name-addressable container of `Param` (reference: param.py:93-214).

    Accepts any mix of Params and iterables of Params; duplicate names raise ValueError.
    Order is declaration order, which is also the order of the columns of `theta`.
    """

    def __init__(self, *args):
        flat = []
        for arg in args:
            if isinstance(arg, Param):
                flat.append(arg)
            else:
                flat.extend(arg)
        for p in flat:
            if not isinstance(p, Param):
                raise AssertionError('All params must be of type "Param"')
        names = [p.name for p in flat]
        dup = sorted({n for n in names if names.count(n) > 1})
        if dup:
            raise ValueError("Duplicate definitions found for param(s): " + ", ".join(dup))
        self._params = flat

    def __len__(self):
        return len(self._params)

    def __getitem__(self, key):
        if isinstance(key, (int, np.integer)):
            return self._params[key]
        if isinstance(key, str):
            for p in self._params:
                if p.name == key:
                    return p
            raise KeyError(key)
        if isinstance(key, slice):
            return ParamSet(self._params[key])
        raise TypeError("ParamSet indices must be int, str or slice")

    def __iter__(self):
        return iter(self._params)

    def __str__(self):
        rows = ["== {0:<15} = {1!s:<15}, tag={2!s:<15}".format(p.name, p.value, p.tag)
                for p in self._params]
        return "\n" + "\n".join(rows) + "\n"

    def _column(self, attr):
        return tuple(getattr(p, attr) for p in self._params)

    names = property(lambda self: self._column("name"))
    labels = property(lambda self: self._column("tex"))
    values = property(lambda self: self._column("value"))
    nominal_values = property(lambda self: self._column("nominal_value"))
    seeds = property(lambda self: self._column("seed"))
    ranges = property(lambda self: self._column("ranges"))
    stds = property(lambda self: self._column("std"))
    tags = property(lambda self: self._column("tag"))
    params = property(lambda self: self._params)

    def to_dict(self):
        return {p.name: p.value for p in self._params}

    def from_tag(self, tag, values=False, index=False, invert=False):
        """Select by tag(s), keeping declaration order (param.py:181-196)."""
        if values and index:
            raise AssertionError("values and index are mutually exclusive")
        wanted = set(np.atleast_1d(tag).tolist())
        picked = [(i, p) for i, p in enumerate(self._params)
                  if (p.tag in wanted) != bool(invert)]
        if values:
            return tuple(p.value for _, p in picked)
        if index:
            return tuple(i for i, _ in picked)
        return ParamSet([p for _, p in picked])

    def remove_params(self, params):
        drop = set(params.names)
        return ParamSet([p for p in self._params if p.name not in drop])

    def extend(self, p):
        if isinstance(p, Param):
            return ParamSet(self._params + [p])
        return ParamSet(self._params + list(p))
