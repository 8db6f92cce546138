"""The drop-in boundary on real hardware: the mexFunction shims of sedumi_amd/mexshims built with g++ against
libsedumi_hip.so (the hipcc build for gfx950) and the package's MEX host (sedumi_amd/mexhost), driven through the same mxArray
marshalling as the reference MEX -- the iteration unit, the process-wide factor cache shared between the .mex binaries, the
dense-column gateways, invcholfac, getada -- plus the host-side N3 entry points (incorder, adendotd, adenscale) on the
product library.  The tests themselves are those of test_mexshims.py / test_oracle.py / test_dense_columns.py; only the
library underneath differs."""
import os

import pytest

from helpers import ROOT, use_hip
from test_mexshims import (build_shims, test_factor_cache_is_shared_between_mex_binaries_and_validated_by_content,  # noqa: F401
                           test_factor_cache_shortcut_for_large_arrays_is_exactly_the_documented_one,
                           test_iteration_units_by_reference_reuse_the_device_state, test_lazy_intermediates_leave_ada_on_the_device, test_lazy_tokens_stale_consumed_and_one_by_one,
                           test_lazy_level_2_is_the_default_and_the_environment_turns_it_off, check_whole_solve_at_lazy_levels, test_units_on_a_problem_without_lp_or_lorentz_part,
                           test_getada_shim_updates_the_global, test_shim_errors_go_through_mexErrMsgTxt, test_shim_incorder,
                           test_shim_invcholfac, test_shims_dense_column_path, test_shims_reproduce_an_iteration_unit,
                           test_shims_symbolic_bit_exact)
from test_oracle import test_incorder_is_bit_exact  # noqa: F401
from test_dense_columns import test_adendotd_and_adenscale_match_reference  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _hip():
    use_hip()


@pytest.fixture(scope="module")
def shimlib():
    from sedumi_amd import capi
    use_hip()
    return capi.DEFAULT_LIB


@pytest.fixture(scope="module")
def shimmex(refmex, shimlib):
    import ctypes
    ctypes.CDLL(shimlib).sdm_mexcache_set_lazy(0)      # (as in test_mexshims.py: the array-by-array comparisons; the library's default is level 2)
    return build_shims(shimlib, os.path.join(ROOT, "tests", "hipemu", "_mexshims_hip"))


@pytest.mark.parametrize("name", ["quantum", "nb", "arch0", "control07"])
def test_whole_solves_through_the_shims_are_the_same_at_lazy_level_0_and_2(shimmex, shimlib, name):
    """The reference's examples, whole solves through the built shims on the hipcc library: lazy level 2 (the default) against level 0."""
    check_whole_solve_at_lazy_levels(shimmex, shimlib, name)
