"""N > 1 path of bench.py on CPU: two ranks under torch.distributed (gloo) run the timing harness —
barriers on both sides of exactly K timed steps, MAX over ranks, one JSON line from rank 0.
The data path itself has no collective (pages are sharded, weak scaling)."""

import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def run_bench(nproc, port):
	env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
	if nproc == 1:
		cmd = [sys.executable, str(ROOT / "bench.py"), "--harness-selftest", "--steps", "5", "--warmup", "2"]
	else:
		cmd = [
			sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
			"--master-addr", "127.0.0.1", "--master-port", str(port), str(ROOT / "bench.py"),
			"--gpus", str(nproc), "--harness-selftest", "--steps", "5", "--warmup", "2"]
	out = subprocess.run(cmd, env=env, cwd=str(ROOT), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
	assert out.returncode == 0, out.stderr.decode()[-2000:]
	lines = [json.loads(l) for l in out.stdout.decode().splitlines() if l.startswith("{")]
	assert len(lines) == 1, "rank 0 prints exactly one JSON line"
	return lines[0]


def test_harness_world_1_and_2():
	one = run_bench(1, 0)
	assert one["n_gpus"] == 1 and one["steps"] == 5 and one["warmup"] == 2
	assert 10.0 <= one["ms_per_step"] < 30.0
	two = run_bench(2, 29517)
	assert two["n_gpus"] == 2
	# rank 1 sleeps 20 ms per step: the harness reports the slowest rank
	assert 20.0 <= two["ms_per_step"] < 45.0


def test_plain_command_with_gpus_2_launches_two_ranks():
	"""`python bench.py --gpus 2 ...` WITHOUT torchrun (WORLD_SIZE unset): bench.py starts the two ranks itself
	as a child process and the one JSON line says n_gpus 2 — a scaling run can never silently measure one rank."""
	env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
	env.update(MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
	cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--harness-selftest", "--steps", "4", "--warmup", "1"]
	out = subprocess.run(cmd, env=env, cwd=str(ROOT), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
	assert out.returncode == 0, out.stderr.decode()[-2000:]
	lines = [json.loads(l) for l in out.stdout.decode().splitlines() if l.startswith("{")]
	assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["steps"] == 4
	assert 20.0 <= lines[0]["ms_per_step"] < 45.0


def test_world_size_that_contradicts_gpus_is_refused():
	env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
	cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--harness-selftest"]
	out = subprocess.run(cmd, env=env, cwd=str(ROOT), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
	assert out.returncode != 0 and b"WORLD_SIZE" in out.stderr
