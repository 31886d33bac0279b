"""The literals the reference hard-codes (S1 = Simulation_testing/Simulation_1/test_DCARL.py)."""
from __future__ import annotations

from dataclasses import dataclass

from ._lib import CParams


@dataclass(frozen=True)
class Params:
    rule_act: int = 0          # S1:43
    n_thres: int = 10          # S1:45
    alpha: float = 0.05        # S1:10 default argument
    scale: float = 150.0       # S1:10 default argument
    cap: float = 100.0         # S1:12
    init_rule: float = 100.0   # S1:52
    init_other: float = -50.0  # S1:51

    def to_c(self) -> CParams:
        return CParams(self.rule_act, self.n_thres, self.alpha, self.scale, self.cap, self.init_rule,
                       self.init_other)
