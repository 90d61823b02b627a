"""`python bench.py --gpus N` with no launcher around it starts its own N ranks (bench.launch_ranks): one child per rank
with the torch.distributed environment, rank 0's one line forwarded, any failing rank fails the job.  The children here
are stand-ins (no GPU on this box); the gpu-marked test runs the real thing, two ranks sharing the test GPU over gloo."""
import json
import os
import subprocess
import sys
import textwrap
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _child(tmp_path, body):
    path = os.path.join(tmp_path, "child.py")
    with open(path, "w") as f:
        f.write(textwrap.dedent(body))
    return [sys.executable, path]


def test_launcher_sets_the_rank_environment_and_forwards_rank0(tmp_path):
    import bench
    cmd = _child(tmp_path, """
        import json, os, sys
        env = {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "FR_BENCH_LAUNCHER")}
        with open(os.path.join(sys.argv[1], "rank" + env["RANK"] + ".json"), "w") as f:
            json.dump({"env": env, "argv": sys.argv[2:]}, f)
        print("noise from rank", env["RANK"])
        if env["RANK"] == "0":
            print(json.dumps({"n_gpus": int(env["WORLD_SIZE"]), "value": 1.0}))
        """)
    rc, line = bench.launch_ranks(3, [str(tmp_path), "--gpus", "3", "--steps", "5"], child_cmd=cmd, timeout=60)
    assert rc == 0
    assert json.loads(line) == {"n_gpus": 3, "value": 1.0}
    ports = set()
    for r in range(3):
        got = json.load(open(os.path.join(tmp_path, f"rank{r}.json")))
        assert got["env"]["RANK"] == str(r) and got["env"]["LOCAL_RANK"] == str(r) and got["env"]["WORLD_SIZE"] == "3"
        assert got["env"]["MASTER_ADDR"] == "127.0.0.1" and got["env"]["FR_BENCH_LAUNCHER"] == "self"
        assert got["argv"] == ["--gpus", "3", "--steps", "5"]
        ports.add(got["env"]["MASTER_PORT"])
    assert len(ports) == 1


def test_a_failing_rank_fails_the_job_and_stops_the_others(tmp_path):
    import bench
    cmd = _child(tmp_path, """
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(7)
        time.sleep(120)          # a rank waiting in a collective for the one that died
        print("{}")
        """)
    t0 = time.monotonic()
    rc, line = bench.launch_ranks(2, [], child_cmd=cmd, timeout=100)
    assert rc == 7 and line is None
    assert time.monotonic() - t0 < 30


def test_a_hung_job_times_out(tmp_path):
    import bench
    cmd = _child(tmp_path, "import time\ntime.sleep(120)\n")
    t0 = time.monotonic()
    rc, line = bench.launch_ranks(2, [], child_cmd=cmd, timeout=1.0)
    assert rc == 124 and line is None
    assert time.monotonic() - t0 < 30


def test_bench_main_with_world_size_set_does_not_launch(tmp_path):
    """Under torch.distributed.run (WORLD_SIZE set) bench.py must be a rank, not a launcher: `--gpus 2` with WORLD_SIZE in
    the environment never reaches launch_ranks.  (Checked on the argument handling only: no GPU here.)"""
    code = ("import sys, os; sys.argv = ['bench.py', '--gpus', '2']; os.environ['WORLD_SIZE'] = '2';"
            "import bench; bench.launch_ranks = lambda *a, **k: (_ for _ in ()).throw(SystemExit(99));"
            "bench.run = lambda: None; bench.main(); print('rank path')")
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "rank path" in out.stdout, out.stderr[-2000:]
    code = code.replace("os.environ['WORLD_SIZE'] = '2';", "os.environ.pop('WORLD_SIZE', None);")
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert out.returncode == 99, (out.returncode, out.stderr[-2000:])


@pytest.mark.gpu
def test_bench_starts_its_own_two_ranks_on_the_gpu(hip_lib):
    """The real thing on the one-GPU box: two ranks sharing the device, gloo for the barrier (`--backend gloo`), voices
    sharding inside the engine; a small tree so that the run takes seconds."""
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--voices", "8",
                          "--partials", "256", "--frames", "512", "--steps", "5", "--warmup", "2", "--repeats", "3"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout[-2000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["config"]["plan"]["shard"]["world"] == 2
    assert res["config"]["launcher"].startswith("bench.py") and res["config"]["ranks"] == 2
    assert res["config"]["rows_of_rank0"] == [0, 4]
    assert res["value"] > 0
