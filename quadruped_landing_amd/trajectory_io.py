"""Trajectory files in the reference's format: one float per line, Z order -- what
`writedlm("data_6.csv", Z_sol, ',')` writes (src/main.ipynb:881) and what the reference's plot
scripts read back (src/plot_data.py:12-14: drop the last 15, reshape (-1, 20))."""
from __future__ import annotations

import numpy as np

from .nlp import num_primals, unpackZ


def save_trajectory(path: str, Z) -> None:
    Z = np.asarray(Z, dtype=np.float64).reshape(-1)
    with open(path, "w") as fh:
        for v in Z:
            fh.write(repr(float(v)) + "\n")  # shortest round-trip decimal, like Julia's print


def load_trajectory(path: str, N: int | None = None):
    Z = np.loadtxt(path, delimiter=",", dtype=np.float64).reshape(-1)
    if N is None:
        if (Z.size + 5) % 20:
            raise ValueError(f"{path}: {Z.size} entries is not 20N-5 for any N")
        N = (Z.size + 5) // 20
    if Z.size != num_primals(N):
        raise ValueError(f"{path}: expected {num_primals(N)} entries for N={N}, found {Z.size}")
    return Z


def as_plot_table(Z):
    """(N-1) x 20 table [state 15 | control 5] the reference's plot scripts build (src/plot_data.py:12-44)."""
    Z = np.asarray(Z, dtype=np.float64).reshape(-1)
    return Z[:-15].reshape(-1, 20)


def states_controls(Z):
    Z = np.asarray(Z, dtype=np.float64).reshape(-1)
    return unpackZ((Z.size + 5) // 20, Z)
