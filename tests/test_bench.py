"""``bench.py``'s launch contract: ``--gpus N`` from a bare shell starts its own N ranks as a child
process (before anything touches the GPU) and relays the one JSON line and the exit code."""

import json
import os
import subprocess
import sys

from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _bare_env():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def test_gpus_n_without_a_launcher_starts_torch_distributed_run_as_a_child(monkeypatch, capsys):
    import bench

    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env

        class R:
            returncode = 7
            stdout = "[Gloo] Rank 0 is connected to 1 peer ranks.\n{\"metric\": \"m\"}\n"
        return R()

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setenv("RANK", "5")          # a stale variable of some outer launcher must not leak in
    with pytest.raises(SystemExit) as exc:
        bench.main()
    assert exc.value.code == 7               # the child's exit code is the bench's
    out = capsys.readouterr()
    assert out.out == '{"metric": "m"}\n' and "[Gloo]" in out.err     # stdout carries the JSON line only
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    i = cmd.index(str(ROOT / "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert "RANK" not in seen["env"]


def test_ranks_that_do_not_match_gpus_are_refused(monkeypatch):
    import bench

    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--no-cpu-baseline"])
    monkeypatch.setenv("WORLD_SIZE", "3")
    with pytest.raises(SystemExit) as exc:
        bench.main()
    assert "WORLD_SIZE=3" in str(exc.value.code)


@pytest.mark.gpu
def test_bench_gpus_2_from_a_bare_shell_on_one_card():
    """The rehearsal: two ranks share the one card (gloo carries the barrier), started by bench.py itself."""
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--workload", "small"], cwd=ROOT, env=_bare_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0
    assert "cpu_baseline" not in line


def test_config1_runs_on_the_cpu_through_the_product_path():
    """BASELINE configs[0] as a bench line: deskew only, no GPU, the host twin on a CPU tensor, equal to the
    oracle bit for bit (the oracle is the timed baseline beside it, never the thing measured)."""
    r = subprocess.run([sys.executable, "bench.py", "--device", "cpu", "--workload", "config1", "--steps", "1", "--warmup", "0"],
                       cwd=ROOT, env=_bare_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 0 and line["dtype"] == "f32" and line["value"] > 0
    assert line["config"]["equals_oracle_bit_for_bit"] is True and line["config"]["raw_shape"] == [256, 64, 256]
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["cores"] == 1
