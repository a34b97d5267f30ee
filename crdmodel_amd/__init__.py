"""crdmodel_amd -- MI355X-native RHS / RK4 path of CRDModel behind a C ABI (include/crd.h).

`crdmodel_amd/csrc` holds the hand-written HIP kernels and the C++ host library (libcrd.so, built in-tree);
`crdmodel_amd.solver` mirrors the reference's interface for this path on top of that ABI via ctypes.
"""
from . import _capi  # noqa: F401
from ._capi import CrdError  # noqa: F401
from .solver import (LocalGroup, PinnedArray, Slab, Writer, block_extents, cycle_agreed, cycle_vote, dims_create, grid_of, halo_plan, initial_conditions, kernel_digest, kernel_digest_of_table_row, kernel_table_digest, launch_plan_candidates, load_ini, make_params, plan_key, rccl_unique_id,  # noqa: F401
                     run_config, slab_extents, stable_dt, steady_state, steady_state_as_printed)
