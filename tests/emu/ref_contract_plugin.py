"""pytest plugin (TEST INFRASTRUCTURE, build container only): lets the reference's own
tests/test_planner_controller_contract.py (and test_mapper_trace_ray.py, test_se3_mpc_with_mapper.py,
test_planner_performance.py, test_planner_controller_integration.py) run *in place, unchanged* against this package on a box
without a GPU by pointing the planner at the host-emulated kernels.  Usage:

  PYTHONPATH=dart_planner_amd/compat:.:tests/emu python -m pytest -c /dev/null --rootdir=/tmp \
      -p ref_contract_plugin -p no:cacheprovider /root/reference/tests/test_planner_controller_contract.py

On a GPU box drop `-p ref_contract_plugin`: the planner then uses libse3mpc.so."""
import build_emu
from numpy_backend import TorchCpuBackend


def pytest_configure(config):
    from dart_planner_amd import capi
    from dart_planner_amd.ops import Ops
    from dart_planner_amd.planning import se3_mpc_planner as mod
    ops = Ops(TorchCpuBackend(), capi.Library(build_emu.build()))
    mod.SE3MPCPlanner._get_ops = lambda self: ops
    # the controller and simulator mirrors run on the device too: same emulated library
    from dart_planner_amd.control import geometric_controller as ctl
    from dart_planner_amd.utils import drone_simulator as sim
    ctl.GeometricController._get_ops = lambda self: ops
    sim.DroneSimulator._get_ops = lambda self: ops
    # the mapper mirror builds its device table through Ops(): hand it the emulated library too
    from dart_planner_amd import voxel_map
    real_init = voxel_map.DeviceVoxelMap.__init__

    def init_with_emulated_ops(self, *a, **k):
        k["ops"] = k.get("ops") or ops
        real_init(self, *a, **k)
    voxel_map.DeviceVoxelMap.__init__ = init_with_emulated_ops
