#!/bin/bash
# Developer build (CPU): libse3mpc with -DSE3MPC_SOLVE_PROFILE (per-section s_memtime sums of every solve wavefront) into tools/probes/,
# next to -- never instead of -- the shipped dart_planner_amd/libse3mpc.so.  Read by tools/gpu_profile_solve_sections.py on the GPU box.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
make -C "$ROOT/dart_planner_amd/csrc" OUT="$ROOT/tools/probes/libse3mpc_solve_profile.so" OBJDIR="$ROOT/build/csrc_solve_profile" \
     EXTRA_HIPFLAGS=-DSE3MPC_SOLVE_PROFILE
