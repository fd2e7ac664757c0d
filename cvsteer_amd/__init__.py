"""cvsteer_amd -- MI355X (gfx950) engine for the cvsteer separable steerable-filter hot path.

The product is the HIP library ``libcvsteer_hip.so`` (C ABI in ``include/cvsteer_hip.h``);
this package is the Python host-side mirror of the reference's ``fa::SteerableFiltersG2`` /
``fa::SteerableFiltersG4`` classes on top of that ABI.  There is no CPU fallback: importing
works anywhere the library loads, but creating a filter object needs a HIP device.
"""
from ._lib import CvsError, abi_version, lib, lib_path  # noqa: F401
from .api import (  # noqa: F401
    SETUP_BASIS, SETUP_FULL, SETUP_ORIENT, SteerableFilters, SteerableFiltersG2, SteerableFiltersG4,
    KIND_G2, KIND_G4, alloc_planes, basis_taps, make_taps, num_basis, pyramid_setup, steer_weights,
)

__all__ = [
    "SteerableFilters", "SteerableFiltersG2", "SteerableFiltersG4", "CvsError", "lib", "lib_path",
    "abi_version", "make_taps", "basis_taps", "num_basis", "steer_weights",
    "SETUP_BASIS", "SETUP_ORIENT", "SETUP_FULL", "KIND_G2", "KIND_G4", "alloc_planes", "pyramid_setup",
]
