"""CPU test of bench.py's N-rank launch path (`--dry-run`: rank plumbing only, gloo, no kernels): started without a
torchrun environment, `python bench.py --gpus 2` must itself become the launcher of 2 ranks, time with a barrier on both
sides, take the max over ranks and have rank 0 print exactly ONE line with n_gpus = 2; a rank-count mismatch must refuse to
print a line."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, extra_env=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env or {})
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)


def test_gpus2_spawns_two_ranks_and_prints_one_line():
    p = _run(["--gpus", "2", "--dry-run", "--steps", "4", "--warmup", "1"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["warmup"] == 1 and d["dry_run"] is True
    # rank 1 sleeps 2 ms per step, rank 0 1 ms: the reported time is the MAX over ranks
    assert d["ms_per_step"] >= 2.0


def test_rank_count_mismatch_refuses_to_print():
    p = _run(["--gpus", "1", "--dry-run"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert "refusing to print" in (p.stderr + p.stdout)


def test_scale_check_script_end_to_end_on_cpu_ranks():
    """tools/scale_check.sh N = 1, 2 with SCALE_EXTRA=--dry-run: the script launches bench.py per N (which respawns itself
    under torch.distributed.run), checks n_gpus and the `rccl` object (world_size as the process group reports it) and
    prints one efficiency line per N - the path the driver's SCALE record takes, minus the GPUs."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(SCALE_EXTRA="--dry-run", SCALE_STEPS="4")
    p = subprocess.run(["bash", os.path.join(ROOT, "tools", "scale_check.sh"), "1", "2"], env=env, capture_output=True, text=True,
                       timeout=600, cwd=ROOT)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-2000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("N=")]
    assert len(lines) == 2 and lines[0].startswith("N=1:") and lines[1].startswith("N=2:"), p.stdout
    assert "'world_size': 2" in lines[1] and "'backend': 'gloo'" in lines[1] and "efficiency" in lines[1], lines[1]


def test_scale_check_rejects_a_line_without_the_ranks(tmp_path):
    """The script's per-N check fails when the process group did not see N ranks (a bench line with n_gpus = 2 but a
    one-rank `rccl` object)."""
    fake = tmp_path / "bench.py"
    fake.write_text('import json,sys\nn=int(sys.argv[sys.argv.index("--gpus")+1])\n'
                    'print(json.dumps({"n_gpus": n, "value": 10.0*n, "ms_per_step": 1.0, "rccl": {"world_size": 1, "backend": "nccl", "buckets": 3}}))\n')
    p = subprocess.run(["bash", os.path.join(ROOT, "tools", "scale_check.sh"), "1", "2"], capture_output=True, text=True,
                       timeout=120, cwd=str(tmp_path))
    assert p.returncode != 0 and "process group reports 1 ranks" in (p.stdout + p.stderr)
