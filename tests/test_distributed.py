"""The N > 1 path: one process per rank over torch.distributed (gloo on CPU), each rank rendering its shard
through the C ABI -- on the CPU oracle here, on the HIP engine in the gpu-marked variant -- and the assembled result
compared bit-for-bit with an unsharded render."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch(world, mode, lib, V, P, T, tmp_path):
    port = free_port()
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), str(r), str(world), str(port),
                               mode, lib, str(V), str(P), str(T), str(tmp_path)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n{outs[r][-2000:]}"
        assert open(os.path.join(tmp_path, f"rank{r}.txt")).read() == "ok"


@pytest.mark.parametrize("mode", ["partials", "voices", "time"])
def test_world2_gloo_oracle(oracle_lib, tmp_path, mode):
    launch(2, mode, "oracle", 3, 64, 48, tmp_path)


def test_world4_gloo_partials_oracle(oracle_lib, tmp_path):
    launch(4, "partials", "oracle", 2, 128, 32, tmp_path)


def test_combine_order_is_the_trees_order():
    """(s0+s1)+(s2+s3), not a left fold: the two differ in f32, and only the former is the graph's association."""
    import numpy as np
    from libfriendship_amd import shard
    s = [np.float32(x) for x in (1e8, 1.0, -1e8, 1.0)]
    tree = (s[0] + s[1]) + (s[2] + s[3])
    assert shard.combine_partial_mixes([np.array([x], np.float32) for x in s])[0] == tree
    fold = ((s[0] + s[1]) + s[2]) + s[3]
    assert fold != tree


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["partials", "voices", "time"])
def test_world2_hip_engine(hip_lib, oracle_lib, tmp_path, mode):
    """Two ranks sharing the one GPU of the test box (gloo carries the host buffers)."""
    launch(2, mode, "hip", 4, 256, 192, tmp_path)
