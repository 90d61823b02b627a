"""The N > 1 path, one process per rank over torch.distributed (gloo): every rank receives the same graph through the
C ABI, fr_set_shard makes it rank r of N, and the exchange step travels through the fr_comm host callback over gloo.
On the CPU the renderer is the host-logic simulator (the engine's own planner / input store / exchange code with
plain-loop kernels, tests/sim_tools.py); the gpu-marked variants run the HIP engine, two ranks sharing the test GPU.
Rows are compared bit-for-bit with an unsharded oracle render.  (In production the transport is the engine's RCCL
communicator: bench.py --gpus N.)"""
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


def launch(world, mode, lib, case, tmp_path):
    if lib == "sim":
        import sim_tools
        sim_tools.build_sim()
    port = free_port()
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), str(r), str(world), str(port),
                               mode, lib, case, str(tmp_path)],
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
        res = os.path.join(tmp_path, f"rank{r}.txt")
        assert p.returncode == 0, f"rank {r} failed: {open(res).read() if os.path.exists(res) else ''}\n{outs[r][-2000:]}"
        assert open(res).read() == "ok"


@pytest.mark.parametrize("mode,case", [("partials", "additive"), ("partials", "effects"), ("voices", "random"), ("voices", "effects"),
                                       ("partials", "additive_tiled")])
def test_world2_gloo(oracle_lib, tmp_path, mode, case):
    launch(2, mode, "sim", case, tmp_path)


@pytest.mark.parametrize("mode,case", [("partials", "effects"), ("voices", "random"), ("partials", "triangle"), ("partials", "effects_tiled")])
def test_world4_gloo(oracle_lib, tmp_path, mode, case):
    launch(4, mode, "sim", case, tmp_path)


@pytest.mark.gpu
@pytest.mark.parametrize("mode,case", [("partials", "additive"), ("partials", "effects"), ("voices", "random"), ("partials", "triangle"),
                                       ("partials", "additive_tiled"), ("partials", "effects_tiled")])
def test_world2_hip_engine(hip_lib, oracle_lib, tmp_path, mode, case):
    """Two ranks sharing the one GPU of the test box; the exchange goes through pinned host memory and gloo."""
    launch(2, mode, "hip", case, tmp_path)
