"""``Op(I, SOC, T)`` - same dataclass as the reference's ``src/operating_point.py``."""

from dataclasses import dataclass

import numpy as np


@dataclass
class Op:
    I: float  # noqa: E741
    SOC: float
    T: float

    def into_array(self) -> np.ndarray:
        return np.array([self.I, self.SOC, self.T])

    def into_row_vector(self) -> np.ndarray:
        return np.array([self.I, self.SOC, self.T]).reshape(1, -1)

    def disp_str(self) -> str:
        return f"I = {self.I:.2f} A, SOC = {self.SOC:.2f} %, T = {self.T:.2f} °C"


def get_cell_tag(cellnr: int) -> str:
    """Column tag of a cell (``src/batt_models/cellnr.py:4-8``): -1 is the pack model."""
    return "pack" if cellnr == -1 else f"c{cellnr}"


def get_causal_tag(causal: bool) -> str:
    return "causal" if causal else "acausal"
