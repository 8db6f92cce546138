"""oracle/glue.py -- TEST INFRASTRUCTURE ONLY.

The MATLAB glue of sedumi_amd/driver/glue.py (pretransfo, set-up, getDAtm, one iteration of the hot path: restated `.m` lines) with the COMPILED
REFERENCE as its MEX host: every C routine that runs is the reference's own (oracle/_ref through oracle.refmex.RefMex), so that the reference
MEX can be fed exactly what `sedumi.m` would feed it.  Nothing here is imported by the product (sedumi_amd/)."""
from __future__ import annotations

from sedumi_amd.driver.glue import *  # noqa: F401,F403
from sedumi_amd.driver.glue import Glue as _Glue, _col  # noqa: F401

from .refmex import RefMex


class Glue(_Glue):
    def __init__(self, ref: RefMex | None = None):
        super().__init__(ref or RefMex())
