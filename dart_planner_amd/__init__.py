"""dart_planner_amd -- MI355X-native SE(3) MPC inner solver behind DART-Planner's
Planner->Controller contract.

Layout: ``csrc/`` hand-written HIP kernels + the C ABI (include/se3mpc.h, built into
``libse3mpc.so``), ``capi.py`` the ctypes binding, ``ops.py`` the tensor front-end,
``planning/``, ``common/`` the host-side mirror of the reference's planner interface,
``distributed.py`` the sharded restart/argmin path over RCCL.  Importing the package does not
load the library; the first op does, and fails loudly if it has not been built.
"""
__version__ = "0.1.0"
