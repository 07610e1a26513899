"""The operating point ``(I, SOC, T)`` a cell's R0 is predicted at, and the column tags of the result frame.

Interface record of the reference (``src/operating_point.py``: attributes ``I`` [A], ``SOC`` [%], ``T`` [degC], the
vector views and the display string that ends up in ``battgpf_info.json``; tags: ``src/batt_models/cellnr.py``).  The
plugin itself only reads ``.I / .SOC / .T`` (``battcellgp_full.py::predict_r0_op``), so the reference's own ``Op``
objects can be passed in as they are.
"""

from __future__ import annotations

import numpy as np

_FIELDS = ("I", "SOC", "T")
_UNITS = ("A", "%", "°C")
PACK = -1  # cell number of the pack model


class Op:
    """Mutable record, constructed positionally or by keyword; compares and prints like the reference's dataclass."""

    __slots__ = _FIELDS

    def __init__(self, I: float, SOC: float, T: float) -> None:  # noqa: E741, N803
        for name, value in zip(_FIELDS, (I, SOC, T)):
            setattr(self, name, value)

    def _values(self) -> tuple:
        return tuple(getattr(self, name) for name in _FIELDS)

    def __eq__(self, other) -> bool:
        if not all(hasattr(other, name) for name in _FIELDS):
            return NotImplemented
        return self._values() == tuple(getattr(other, name) for name in _FIELDS)

    __hash__ = None  # mutable, like a plain dataclass

    def __repr__(self) -> str:
        return "Op(" + ", ".join(f"{name}={value!r}" for name, value in zip(_FIELDS, self._values())) + ")"

    def into_array(self) -> np.ndarray:
        return np.asarray(self._values())

    def into_row_vector(self) -> np.ndarray:
        return self.into_array()[np.newaxis, :]

    def disp_str(self) -> str:
        return ", ".join(f"{name} = {value:.2f} {unit}" for name, value, unit in zip(_FIELDS, self._values(), _UNITS))


def get_cell_tag(cellnr: int) -> str:
    """``c<n>`` for a cell, ``pack`` for the pack model (cell number -1)."""
    return "pack" if cellnr == PACK else f"c{cellnr}"


def get_causal_tag(causal: bool) -> str:
    return ("acausal", "causal")[bool(causal)]
