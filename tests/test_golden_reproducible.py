"""Build container only (needs /root/reference; skipped on the GPU box, where the reference does not exist): the golden generators are
re-run into a scratch directory and must reproduce the committed fixtures BYTE FOR BYTE -- the fixtures are outputs of the reference's own
code, not hand-edited data.  (wire_messages.json carries wall-clock stamps and is compared through its payloads.)"""
import filecmp
import json
import os
import subprocess
import sys

import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/src/dart_planner"), reason="the reference is only present in the build container")


@pytest.mark.parametrize("script,files", [
    ("make_golden_cfg1.py", ["cfg1_solves.npz", "cfg1_solves.json"]),
    ("make_golden_bifurcation.py", ["bifurcation_case.npz", "bifurcation_case.json"]),
    ("make_golden_controller.py", ["controller_cases.npz", "controller_cases.json"]),
    ("make_golden.py", ["solve_cases.npz", "solve_cases.json", "path_functions.npz", "path_functions.json", "mapper_spheres.npz", "mapper_spheres.json"]),
])
def test_generator_reproduces_committed_fixtures(tmp_path, script, files):
    env = dict(os.environ, SE3MPC_GOLDEN_OUT=str(tmp_path), PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, os.path.join(GOLDEN, script)], cwd=str(tmp_path), env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    for f in files:
        assert filecmp.cmp(os.path.join(GOLDEN, f), os.path.join(str(tmp_path), f), shallow=False), f"{f} differs from what {script} writes"
    if script == "make_golden.py":
        a, b = json.load(open(os.path.join(GOLDEN, "wire_messages.json"))), json.load(open(os.path.join(str(tmp_path), "wire_messages.json")))
        assert a["secret"] == b["secret"] and [json.loads(m["raw"])["data"] for m in a["messages"]] == [json.loads(m["raw"])["data"] for m in b["messages"]]
