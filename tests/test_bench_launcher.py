"""bench.py --gpus N started plainly (no torch.distributed.run): the parent must hand every child
the rendezvous environment torch.distributed.run would, relay rank 0's line only, and report the
worst return code -- all without importing torch or touching a GPU itself.  CPU only."""
import io
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_rank_env_is_what_torchrun_sets():
    env = bench.rank_env(2, 4, 29517, base={"PATH": "/bin", "WORLD_SIZE": "9"})
    assert env["RANK"] == "2" and env["LOCAL_RANK"] == "2" and env["WORLD_SIZE"] == "4"
    assert env["MASTER_ADDR"] == "127.0.0.1" and env["MASTER_PORT"] == "29517"
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and env["PATH"] == "/bin"


def test_defaults_follow_baseline_configs():
    a = bench.parse_args([])
    assert (a.gpus, a.xrows, a.yrows, a.dim, a.mode) == (1, 1_000_000, 1_000_000, 128, "ranks")
    a = bench.parse_args(["--gpus", "8"])
    assert (a.xrows, a.yrows) == (4_000_000, 500_000)
    assert "configs[4]: 4M x 4M over 8 GPUs" in bench.workload_name(a.xrows, a.yrows, a.dim, 8)


def _child(tmp_path, body):
    p = tmp_path / "child.py"
    p.write_text(textwrap.dedent(body))
    return str(p)


def test_launcher_spawns_ranks_and_relays_rank0(tmp_path):
    script = _child(tmp_path, """
        import json, os, sys
        keys = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")
        print(json.dumps({"argv": sys.argv[1:], **{k: os.environ[k] for k in keys}}))
    """)
    out = io.StringIO()
    rc = bench.launch_ranks(3, ["--gpus", "3", "--steps", "2"], script=script, out=out)
    assert rc == 0
    lines = [l for l in out.getvalue().splitlines() if l.strip()]
    assert len(lines) == 1  # only rank 0's stdout is relayed
    doc = json.loads(lines[0])
    assert doc["RANK"] == "0" and doc["WORLD_SIZE"] == "3" and doc["MASTER_ADDR"] == "127.0.0.1"
    assert doc["argv"] == ["--gpus", "3", "--steps", "2"]


def test_launcher_reports_a_failing_rank_and_ends_the_others(tmp_path):
    script = _child(tmp_path, """
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(7)
        time.sleep(60)  # a rank waiting in a collective for the dead one
    """)
    import time
    t0 = time.time()
    rc = bench.launch_ranks(2, [], script=script, out=io.StringIO())
    assert rc == 7 and time.time() - t0 < 30


def test_parent_does_not_import_torch():
    """`python bench.py --gpus 2` must not initialise anything in the parent: run the real script with
    a python wrapper that fails on `import torch` in the PARENT only."""
    code = ("import sys, runpy, os\n"
            "class Block:\n"
            "    def find_spec(self, name, path=None, target=None):\n"
            "        if name == 'torch': raise ImportError('parent imported torch')\n"
            "sys.meta_path.insert(0, Block())\n"
            "import bench\n"
            "bench.launch_ranks = lambda world, argv, **kw: print('LAUNCH', world, argv) or 0\n"
            "sys.argv = ['bench.py', '--gpus', '2', '--steps', '1']\n"
            "os.environ.pop('WORLD_SIZE', None)\n"
            "bench.main()\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "LAUNCH 2 ['--gpus', '2', '--steps', '1']" in r.stdout
    assert "torch" not in r.stderr


def test_roofline_arithmetic_matches_survey_8d():
    """achieved = pairs/s x 32 v_sad lane-ops / (CUs x 64 x f_clk); the BASELINE reading
    streams 128 B per pair against 8 TB/s; compulsory bytes (N + M) x 128 + 24 N."""
    a = bench.parse_args([])
    r = bench.roofline_of(a, launches=4, tile_ms=4 * 900.0, merge_ms=4 * 0.05)
    pairs_per_s = 1e12 / 0.9
    assert abs(r["achieved"] - pairs_per_s * 32 / 1e12) < 1e-6
    assert abs(r["peak"] - 256 * 64 * 2.4e9 / 1e12) < 1e-9 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert abs(r["avg_launch_ms"] - 900.0) < 1e-9 and r["launches_timed"] == 4
    assert abs(r["hbm_streaming_equiv"]["achieved"] - pairs_per_s * 128 / 1e9) < 1e-3
    assert r["hbm_compulsory"]["bytes"] == (1_000_000 + 1_000_000) * 128 + 24 * 1_000_000
    assert r["bound"] == "valu" and r["kernel"] == "l1k2_tile_kernel"


def test_host_cpu_reports_model_and_counts():
    model, physical, logical = bench.host_cpu()
    assert isinstance(model, str) and model and logical == os.cpu_count()
    assert physical is None or 1 <= physical <= logical
