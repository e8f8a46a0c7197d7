"""Components written in Python, for graphs that mix them with the GPU components.

Mirrors the surface of the reference's ``rscm.component`` (python/rscm/component.py: ``Component`` with
``Input`` / ``Output`` / ``State`` declarations, generated ``Inputs`` / ``Outputs`` classes, a class
registry) and ``PythonComponent.build`` (python/rscm/_lib/core/__init__.pyi); the behaviour follows
the reference's tests/test_typed_python_component.py.  This is NOT a GPU path: ``solve`` runs on the
host, once per member and step, between the launches of the linked ensembles of a ``GraphModel``
(``rscm_amd.core``) -- the rows it reads come back from the device, the rows it produces are stored
in a device series so that GPU components can link to them.  Meant for small ensembles and for the
glue a model needs around the fast components.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import numpy as np

__all__ = ["Component", "Input", "Output", "State", "PythonComponent", "RequirementDefinition", "TimeseriesWindow"]


class _Declaration:
    kind = ""

    def __init__(self, name: str, unit: str = "", grid: str = "Scalar"):
        if grid != "Scalar":
            raise NotImplementedError("Python components exchange scalar variables on this path")
        self.name, self.unit, self.grid = name, unit, grid


class Input(_Declaration):
    kind = "Input"


class Output(_Declaration):
    kind = "Output"


class State(_Declaration):
    kind = "State"


class RequirementDefinition:
    def __init__(self, name: str, unit: str, requirement_type: str, grid_type: str = "Scalar"):
        self.name, self.unit, self.requirement_type, self.grid_type = name, unit, requirement_type, grid_type

    def __repr__(self) -> str:
        return f"RequirementDefinition({self.name!r}, {self.unit!r}, {self.requirement_type})"


class TimeseriesWindow:
    """What ``solve`` sees of one variable: the member's series and the current index, read by the
    reference's rules (state/windows.rs:155-247): ``at_start`` = index n, ``at_end`` = n+1 (None past
    the end), ``get`` picks by the variable's source, ``previous`` = n-1."""

    def __init__(self, series: np.ndarray, index: int, source: str):
        self._series, self._index, self._source = series, index, source

    def at_start(self) -> float:
        return float(self._series[self._index])

    def at_end(self) -> Optional[float]:
        j = self._index + 1
        return float(self._series[j]) if j < len(self._series) else None

    def get(self) -> float:
        if self._source == "UpstreamOutput":
            e = self.at_end()
            return self.at_start() if e is None else e
        return self.at_start()

    def at_offset(self, k: int) -> Optional[float]:
        j = self._index + k
        return float(self._series[j]) if 0 <= j < len(self._series) else None

    @property
    def current(self) -> float:
        return self.at_start()

    @property
    def previous(self) -> float:
        if self._index == 0:
            raise ValueError("no previous value at the first time step")
        return float(self._series[self._index - 1])


class Component:
    """Base class of Python components.  Declare the variables as class attributes::

        class Scale(Component):
            x = Input("X", unit="K")
            y = Output("Y", unit="K")
            def solve(self, t_current, t_next, inputs):
                return self.Outputs(y=2.0 * inputs.x.at_start())
    """

    _registry: Dict[str, type] = {}
    _declarations: Dict[str, _Declaration] = {}

    def __init_subclass__(cls, register: bool = True, **kwargs):
        super().__init_subclass__(**kwargs)
        decl: Dict[str, _Declaration] = {}
        for base in reversed(cls.__mro__[1:]):
            decl.update(getattr(base, "_declarations", {}))
        decl.update({k: v for k, v in vars(cls).items() if isinstance(v, _Declaration)})
        cls._declarations = decl
        readable = [k for k, d in decl.items() if d.kind in ("Input", "State")]
        writable = [k for k, d in decl.items() if d.kind in ("Output", "State")]

        class Inputs:
            __slots__ = tuple(readable)

            def __init__(self, **windows):
                for k in readable:
                    setattr(self, k, windows[k])

        class Outputs:
            _fields = tuple(writable)

            def __init__(self, **values):
                missing = [k for k in writable if k not in values]
                if missing:
                    raise TypeError("Missing required output fields: " + ", ".join(missing))
                unknown = [k for k in values if k not in writable]
                if unknown:
                    raise TypeError("Unknown output fields: " + ", ".join(unknown))
                self._values = {k: float(values[k]) for k in writable}

            def to_dict(self) -> Dict[str, float]:
                return {decl[k].name: v for k, v in self._values.items()}

        Inputs.__qualname__, Outputs.__qualname__ = f"{cls.__qualname__}.Inputs", f"{cls.__qualname__}.Outputs"
        cls.Inputs, cls.Outputs = Inputs, Outputs
        if register:
            Component._registry[cls.__name__] = cls

    @classmethod
    def get_registered_components(cls) -> Dict[str, type]:
        return dict(Component._registry)

    @classmethod
    def get_component(cls, name: str) -> type:
        if name not in Component._registry:
            raise KeyError(f"No component registered with name {name!r}")
        return Component._registry[name]

    def definitions(self) -> List[RequirementDefinition]:
        return [RequirementDefinition(d.name, d.unit, d.kind, d.grid) for d in type(self)._declarations.values()]

    def solve(self, t_current: float, t_next: float, inputs):
        raise NotImplementedError


class PythonComponent:
    """A Python component as ``ModelBuilder.with_py_component`` takes it (``PythonComponent.build``).
    Carries what the graph resolution needs: a type name and the definitions in macro order
    (inputs, outputs, states)."""

    is_python = True
    parameters: Dict[str, float] = {}

    def __init__(self, component: Component):
        self.component = component
        self.type_name = type(component).__name__
        d = list(type(component)._declarations.items())
        order = [x for x in d if x[1].kind == "Input"] + [x for x in d if x[1].kind == "Output"] + [x for x in d if x[1].kind == "State"]
        self.fields: List[Tuple[str, _Declaration]] = order
        self.definitions = [(decl.name, decl.unit, decl.kind) for _, decl in order]

    @staticmethod
    def build(component: Component) -> "PythonComponent":
        if not isinstance(component, Component):
            raise TypeError("PythonComponent.build takes an instance of rscm_amd.component.Component")
        return PythonComponent(component)

    def input_names(self) -> List[str]:
        return [n for n, _, k in self.definitions if k in ("Input", "State")]

    def output_names(self) -> List[str]:
        return [n for n, _, k in self.definitions if k in ("Output", "State")]

    def solve_member(self, t0: float, t1: float, series: Dict[str, np.ndarray], member: int, index: int,
                     sources: Dict[str, str]) -> Dict[str, float]:
        windows = {attr: TimeseriesWindow(series[decl.name][:, member], index, sources.get(decl.name, "Exogenous"))
                   for attr, decl in self.fields if decl.kind in ("Input", "State")}
        out = self.component.solve(t0, t1, type(self.component).Inputs(**windows))
        if isinstance(out, dict):
            return {k: float(v) for k, v in out.items()}
        if not isinstance(out, type(self.component).Outputs):
            raise TypeError(f"{self.type_name}.solve must return self.Outputs(...) or a dict of variable names")
        return out.to_dict()
