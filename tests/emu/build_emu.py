"""Build tests/emu/libse3mpc_emu.so: the UNMODIFIED product sources (dart_planner_amd/csrc/*.hip)
compiled by g++ against the host emulation of HIP in tests/emu/hip/hip_runtime.h.  Test
infrastructure only (see that header); the product never loads this library."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "dart_planner_amd", "csrc")
OUT = os.path.join(HERE, "libse3mpc_emu_san.so" if os.environ.get("SE3MPC_EMU_FLAGS") else "libse3mpc_emu.so")


def build(force: bool = False) -> str:
    srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    deps = srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".inc"))]
    deps += [os.path.join(ROOT, "include", "se3mpc.h"), os.path.join(HERE, "hip", "hip_runtime.h"),
             os.path.join(HERE, "wave_ops.hpp")]
    deps = [d for d in deps if os.path.exists(d)]
    if not force and not os.environ.get("SE3MPC_EMU_FLAGS") and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps):
        return OUT
    cmd = ["g++", "-std=c++20", "-O1", "-g", "-fPIC", "-shared", "-ffp-contract=off",
           "-Wno-unknown-pragmas", "-Wno-attributes",
           "-I" + HERE, "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
    cmd += os.environ.get("SE3MPC_EMU_FLAGS", "").split()        # e.g. -fsanitize=undefined -fno-sanitize-recover=all (a sanitizer run of the kernels on the host)
    for s in srcs:
        cmd += ["-x", "c++", s]
    cmd += ["-o", OUT]
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    print(build(force=True))
