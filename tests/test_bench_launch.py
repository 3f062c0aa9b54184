"""bench.py's launch contract on CPU: `--gpus N` outside a launcher starts N ranks itself, every rank
refuses to run when WORLD_SIZE disagrees with --gpus, and the line reports the ranks that actually ran.

On this box there is no GPU, so `--rehearse` takes the CPU stand-in step: launcher, process group (gloo), batch shards,
pipelined in-place all-gather, barrier + max-over-ranks timing and the JSON line are the real code of the N > 1 path."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    env["CUDA_VISIBLE_DEVICES"] = ""         # the CPU form of the rehearsal, also on a box that has a GPU
    env["HIP_VISIBLE_DEVICES"] = ""
    return subprocess.run([sys.executable, BENCH, *args], env=env, capture_output=True, text=True, timeout=timeout)


def _line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.lstrip().startswith("{")]
    assert lines, stdout
    return json.loads(lines[-1])


def test_gpus2_self_launches_two_ranks_and_reports_them():
    r = _run(["--gpus", "2", "--rehearse", "--steps", "3", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert "launching 2 ranks" in r.stderr
    line = _line(r.stdout)
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1
    assert line["collective"] == {"backend": "gloo", "ranks": 2, "chunks_per_step": 1}
    assert line["config"]["global_batch"] == 2 * 8 and line["config"]["parallelism"].startswith("dp2")
    assert line["output_check"]["status"] == "ok" and line["output_check"]["gathered_shards_match_their_ranks"] is True
    assert "rehearsal" in line and line["scaling"] == "weak"


def test_gpus2_with_the_shard_cut_into_chunks():
    # round 4: every step's shard in 4 groups of sequences, each with its own in-place all-gather (ChunkedContextGatherer)
    r = _run(["--gpus", "2", "--rehearse", "--steps", "3", "--warmup", "1", "--gather-chunks", "4"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = _line(r.stdout)
    assert line["n_gpus"] == 2 and line["collective"] == {"backend": "gloo", "ranks": 2, "chunks_per_step": 4}
    assert line["output_check"]["status"] == "ok" and line["output_check"]["gathered_shards_match_their_ranks"] is True


def test_gpus_must_match_the_world_the_launcher_made():
    # a 1-rank environment (WORLD_SIZE=1, as a launcher with --nproc-per-node 1 sets it) asked for 2 GPUs: no line, rc != 0
    r = _run(["--gpus", "2", "--rehearse"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.lstrip().startswith("{")]
    assert "refusing to run" in r.stderr


def test_single_process_cannot_rehearse_many_gpus_silently():
    # --gpus 1 inside a 2-rank world is the converse mismatch
    r = _run(["--gpus", "1", "--rehearse"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1",
                                             "MASTER_PORT": "1"})
    assert r.returncode != 0 and "refusing to run" in r.stderr
