"""experiments/bvhlab -- the host-side laboratory the round-5 tree-quality pass was prototyped in -- still builds and runs (it is not part of the
product; this keeps it from rotting): PLOC, eight batches of reinsertion, the 4-wide collapse and the traversal simulator on a few thousand
triangles; the pass must lower the cost and must not change what the rays hit."""
import os
import re
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lab_builds_and_reinsertion_lowers_the_cost(tmp_path):
    exe = str(tmp_path / "bvhlab")
    subprocess.check_call(["g++", "-O2", "-fopenmp", "-std=c++17", "-o", exe, os.path.join(ROOT, "experiments", "bvhlab", "bvhlab.cpp")])
    rs = np.random.RandomState(1)
    n = 6000
    c = rs.uniform(-1, 1, (n, 1, 3))
    tris = (c + rs.normal(0, 0.03, (n, 3, 3))).astype(np.float32)
    tris[:40] = (rs.uniform(-1, 1, (40, 1, 3)) + rs.normal(0, 0.8, (40, 3, 3))).astype(np.float32)  # a few large triangles: what PLOC merges badly
    base = str(tmp_path / "scene")
    tris.reshape(n, 9).tofile(base + ".tris")
    o = np.tile(np.array([[0.0, 0.0, 4.0]], np.float32), (3000, 1))
    d = np.concatenate([rs.uniform(-0.4, 0.4, (3000, 2)), -np.ones((3000, 1))], 1)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    np.concatenate([o, d.astype(np.float32)], 1).astype(np.float32).tofile(base + ".rays")
    env = dict(os.environ, OMP_NUM_THREADS="2")

    def run(*args):
        out = subprocess.run([exe, base] + list(args), capture_output=True, text=True, env=env, timeout=300).stdout
        sah = [float(x) for x in re.findall(r"SAH(?:\(internal area / root area\))? = ([0-9.]+)", out)]
        rows = {m[0]: (float(m[1]), float(m[2]), float(m[3])) for m in re.findall(r"(camera|bounce|shadow)\s+\d+ rays:\s+([0-9.]+) nodes\s+([0-9.]+) tris per ray \(([0-9.]+) %", out)}
        return sah, rows, out

    sah0, rows0, _ = run()
    sah8, rows8, out = run("reinsert=8", "sparse=1", "fullevery=3")
    assert len(sah8) == 9 and sah8[-1] < 0.97 * sah0[0], out
    assert rows8["camera"][2] == rows0["camera"][2] and rows8["bounce"][2] == rows0["bounce"][2]  # same hit rates: the tree is still a tree over all triangles
    assert rows8["camera"][0] < rows0["camera"][0]
