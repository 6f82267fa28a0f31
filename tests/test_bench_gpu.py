"""bench.py as the driver starts it: `python bench.py --gpus N` from a plain shell.  On the one-GPU box the N = 2 run shares the device
(UNICEPTION_AMD_BENCH_SHARE_GPU=1: gloo instead of RCCL, the line is marked as a dry run) — the point is the control flow: self-launch,
rendezvous, one model per rank, fences, max over ranks, ONE line from rank 0."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, extra_env=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return lines[0]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["fwd", "train"])
def test_bench_two_ranks_from_plain_python(gpu, mode):
    line = _bench(["--gpus", "2", "--mode", mode, "--pairs", "1", "--img", "224", "--steps", "1", "--warmup", "1", "--no-cpu-baseline",
                   "--no-reference-policy", "--no-roofline", "--train-pairs", "1", "--train-steps", "1"], {"UNICEPTION_AMD_BENCH_SHARE_GPU": "1"})
    assert line["n_gpus"] == 2 and line["config"]["global_pairs_per_step"] == 2 and line["scaling"] == "weak"
    assert line["value"] > 0 and "shared_gpu_dry_run" in line["config"]
    # at N > 1 the line carries the exchange it ran — in the forward line through the `train_step` leg every rank runs
    assert line["rccl_ranks"] == 2
    ex = line["train_step"]["exchange"] if mode == "fwd" else line["exchange"]
    assert ex["rccl_ranks"] == 2 and ex["buckets"] >= 1 and sum(ex["bucket_bytes"]) > 4 * 300e6
    assert ex["comm_ms_per_step"] > 0 and ex["exposed_comm_ms_per_step"] >= 0 and ex["steps_measured"] >= 1
    if mode == "fwd":
        assert line["train_step"]["n_gpus"] == 2 and line["train_step"]["pairs_per_s"] > 0


@pytest.mark.gpu
def test_bench_batch_sweep_line(gpu):
    line = _bench(["--img", "224", "--sweep", "1,2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-reference-policy", "--no-roofline"])
    sw = line["batch_sweep"]
    assert [e["pairs"] for e in sw] == [1, 2] and all(e["ms_per_batch"] > 0 and e["pairs_per_s"] > 0 for e in sw)
