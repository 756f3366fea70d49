"""PlanarQuadruped model parameters (mirror of src/planar_quadruped.jl:11-26)."""
from __future__ import annotations

from dataclasses import dataclass


@dataclass(frozen=True)
class PlanarQuadruped:
    g: float = -9.81   # gravity
    mb: float = 10.0   # body mass
    mf: float = 0.1    # foot mass
    lb: float = 0.5    # body length
    l1: float = 0.25   # thigh length
    l2: float = 0.25   # calf length

    @property
    def Ib(self) -> float:
        return self.mb * self.lb**2 / 12  # src/planar_quadruped.jl:41


def state_dim(_model=None) -> int:
    return 15  # src/planar_quadruped.jl:25


def control_dim(_model=None) -> int:
    return 5  # src/planar_quadruped.jl:26
