"""sedumi_amd.driver -- SeDuMi's interior-point loop without MATLAB / Octave: `solve(At, b, c, K)` runs sedumi.m's
iteration (restated in loop.py) on this package's library -- the normal-equations hot path on the resident plan -- with the rest of the cone
algebra on numpy / LAPACK (conemex.py).  Nothing of the reference is needed at run time."""
from .loop import Sedumi, load_mat, solve  # noqa: F401
