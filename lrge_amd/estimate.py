"""Mirror of liblrge's operator interface (liblrge/src/estimate.rs:8-78): the `Estimate` trait with
its provided `estimate()` method and `EstimateResult`."""
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np

from . import engine

LOWER_QUANTILE = 0.15   # estimate.rs:40-46: the 15th-65th percentile interval (~92 % confidence)
UPPER_QUANTILE = 0.65


class LrgeError(Exception):
    """error.rs:6-33.  `kind` is the variant name."""

    def __init__(self, kind, msg):
        super().__init__("%s: %s" % (kind, msg))
        self.kind = kind


@dataclass
class EstimateResult:          # estimate.rs:8-17
    lower: Optional[float]
    estimate: Optional[float]
    upper: Optional[float]
    no_mapping_count: int


class Estimate:
    """trait Estimate (estimate.rs:21-78)."""

    def generate_estimates(self) -> Tuple[np.ndarray, int]:
        raise NotImplementedError

    def estimate(self, finite: bool, lower_quant: Optional[float], upper_quant: Optional[float]) -> EstimateResult:
        estimates, no_mapping_count = self.generate_estimates()
        lower, med, upper = engine.median(estimates, finite, lower_quant, upper_quant)
        f = lambda v: None if v is None else float(v)
        return EstimateResult(f(lower), f(med), f(upper), int(no_mapping_count))
