"""CPU: libse3mpc.so (cross-compiled for gfx950 by __graft_entry__.build()) loads without a GPU and
exports exactly the entry points include/se3mpc.h declares; the ctypes binding covers all of them.
No compute call is made here."""
import ctypes
import os
import re

import pytest

from dart_planner_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "se3mpc.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(se3mpc_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib_path():
    if not os.path.exists(capi.DEFAULT_LIBRARY):
        import __graft_entry__
        __graft_entry__.build()
    return capi.DEFAULT_LIBRARY


def test_header_and_binding_agree():
    assert declared_functions() == sorted(capi.exported_symbols())


def test_library_exports_every_declared_symbol(lib_path):
    dll = ctypes.CDLL(lib_path)
    missing = [f for f in declared_functions() if not hasattr(dll, f)]
    assert not missing, missing


def test_library_loads_through_the_binding_without_a_gpu(lib_path):
    lib = capi.Library(lib_path)
    assert lib.abi_version() == 1
    assert lib.default_params().as_dict() == capi.Params.reference_defaults().as_dict()
    assert lib.check_params(capi.Params.reference_defaults(horizon=65)) == -2
    assert lib.device_count() >= 0          # 0 on a CPU-only box; must not crash
    assert ctypes.sizeof(capi.Params) == 160 and ctypes.sizeof(capi.SolveInfo) == 24 and ctypes.sizeof(capi.VoxelMapDesc) == 48
    assert ctypes.sizeof(capi.ControllerParams) == 304 and ctypes.sizeof(capi.SimulatorParams) == 56      # static_asserts in csrc/closed_loop.hip
    assert bytes(lib.controller_default_params()) == bytes(capi.ControllerParams.from_config(_ReferenceControllerDefaults()))


class _ReferenceControllerDefaults:
    """GeometricControllerConfig after the "sitl_optimized" profile, as the reference instantiates it (tests/golden/controller_cases.json)."""
    kp_pos, ki_pos, kd_pos = (20.0, 20.0, 25.0), (1.5, 1.5, 2.0), (10.0, 10.0, 12.0)
    kp_att, kd_att = (18.0, 18.0, 8.0), (7.0, 7.0, 3.5)
    inertia, max_torque_xyz, max_integral_per_axis = (0.02, 0.02, 0.04), (0.5, 0.5, 0.05), (2.0, 2.0, 3.0)
    max_integral_pos, max_tilt_angle, mass, gravity, max_thrust, min_thrust = 2.5, 0.7853981633974483, 1.0, 9.80665, 22.0, 0.8
    tracking_error_threshold, velocity_error_threshold = 1.0, 0.6
    back_calculation_gain, integral_decay_factor, saturation_threshold, yaw_singularity_threshold, default_heading_yaw = 0.1, 0.99, 0.95, 0.1, 0.0
    anti_windup_method, yaw_singularity_fallback_method = "clamping", "skip_yaw"


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(capi.Se3mpcLibraryError, match="no CPU fallback"):
        capi.Library(str(tmp_path / "libse3mpc.so"))


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under dart_planner_amd/ may import it."""
    bad = []
    for d, _, files in os.walk(os.path.join(ROOT, "dart_planner_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                txt = open(os.path.join(d, f), errors="ignore").read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M) or re.search(r"#include\s+[<\"].*oracle", txt):
                    bad.append(os.path.join(d, f))
    assert not bad, bad


def test_header_is_plain_c(tmp_path):
    """include/se3mpc.h is the drop-in boundary: it must compile as C99 (no C++ or torch types in the signatures)."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    src = tmp_path / "hdr.c"
    src.write_text('#include "se3mpc.h"\nint main(void) { se3mpc_params p; se3mpc_voxel_map m; se3mpc_solve_info i;\n'
                   '  (void)p; (void)m; (void)i; return se3mpc_abi_version() == SE3MPC_ABI_VERSION ? 0 : 1; }\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"),
                    str(src)], check=True)


def test_solver_kernels_do_not_spill():
    """The packed solver kernels (solve_kernel<float|double, 8|16|32|64>) are compiled for two wavefronts per SIMD (256 registers) and must
    need neither spilled vector registers nor scratch memory: spill code inside the divergent control flow of co-resident problems is
    what DESIGN.md section 5.2 rules out instead of reasoning about it.  Read from the ISA metadata of a device-only `hipcc -S`."""
    import re
    import shutil
    import subprocess
    import tempfile
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc) and shutil.which("hipcc") is None:
        pytest.skip("no hipcc on this machine")
    src = os.path.join(ROOT, "dart_planner_amd", "csrc", "solve_kernel.hip")
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "solve_kernel.s")
        subprocess.run([hipcc if os.path.exists(hipcc) else "hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
                        "-I" + os.path.join(ROOT, "dart_planner_amd", "csrc"), "-fno-gpu-rdc", "-fno-slp-vectorize", "--cuda-device-only", "-S", src, "-o", out],
                       check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        text = open(out).read()
    meta = text[text.index("amdhsa.kernels:"):]
    kernels = [blk for blk in meta.split("  - .agpr_count:")[1:] if "solve_kernel" in blk]
    assert len(kernels) == 8, len(kernels)
    for blk in kernels:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        get = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))
        assert get("vgpr_spill_count") == 0 and get("private_segment_fixed_size") == 0, (name, get("vgpr_spill_count"), get("private_segment_fixed_size"))
        assert get("vgpr_count") <= 256 and int(blk.split("\n")[0].strip()) == 0, (name, get("vgpr_count"))     # two wavefronts per SIMD, no AGPRs
