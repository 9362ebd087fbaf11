"""CPU, world_size 2 over gloo: `python bench.py --gpus N` starts itself (bench.launch: the counterpart of the reference's
`launch(main, num_gpus, ...)`, tools/plain_train_net.py:660-671).  The detector is stubbed; what runs is bench.py's own
argument / environment handling, the `time_steps` barrier + MAX-over-ranks reduction, the whole-job value and the
rank-0-only JSON line."""
import json
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _stub_make_step(args, rank, world, device):
    import torch.distributed as dist
    from dafne_amd.evaluation.gather import gather_detections
    calls = {"n": 0}
    rows = torch.zeros(args.batch, 8, 18)
    counts = torch.full((args.batch,), rank + 1, dtype=torch.int32)

    def step():
        calls["n"] += 1
        time.sleep(0.02 * (rank + 1))            # rank 1 is the slow one: the MAX reduction must report ITS time
        if world > 1:
            return gather_detections(rows, counts, dst=0)   # the per-step detection gather of the real step
        return None

    def finish(out):
        out["stub"] = {"calls_rank0": calls["n"], "env": {k: os.environ.get(k) for k in
                                                          ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")},
                       "dist_initialized": dist.is_initialized()}
    return step, finish


def stub_target(args):
    """Picklable worker body for bench.launch: bench.run with the detector stubbed, gloo on CPU."""
    import bench
    return bench.run(args, make_step=_stub_make_step, backend="gloo", device_kind="cpu")


_DRIVER = """
import sys
sys.path.insert(0, %r)
sys.path.insert(0, %r)
import bench
from test_bench_launch import stub_target
bench.launch(bench.parse_args(sys.argv[1:]), target=stub_target)
"""


def _run(argv, env_extra=None, launcher=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    cmd = [sys.executable] + (launcher or []) + ["-c", _DRIVER % (ROOT, os.path.join(ROOT, "tests"))] + argv
    if launcher:        # torch.distributed.run wants a script path
        path = os.path.join(ROOT, "tests", "_bench_stub_main.py")
        cmd = [sys.executable] + launcher + [path] + argv
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    return lines


def test_plain_command_self_launches_two_ranks():
    lines = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "4", "--regions", "1"])
    assert len(lines) == 1, lines                          # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 8 and out["config"]["per_gpu_batch"] == 4
    assert out["stub"]["calls_rank0"] == 5                 # W + K steps, exactly, + the one recorded step behind the timed region
    assert out["stub"]["env"]["WORLD_SIZE"] == "2" and out["stub"]["env"]["RANK"] == "0"
    assert out["stub"]["env"]["MASTER_ADDR"] == "127.0.0.1" and out["stub"]["dist_initialized"]
    # 3 steps of the SLOW rank (40 ms each): max over ranks, not rank 0's own 20 ms
    assert out["ms_per_step"] >= 39.0
    # the line proves what the collective delivered: group size and backend from the process group itself, the images and
    # detections the last gather landed on rank 0 (rank r contributes 4 images with r + 1 detections each), every rank's time
    d = out["distributed"]
    assert d["process_group_size"] == 2 and d["world_size_from_env"] == 2 and d["backend"] == "gloo"
    assert d["gathered_images"] == 8 == d["expected_images"] and d["gathered_detections"] == 4 * 1 + 4 * 2
    pr = d["per_rank_ms_per_step"]
    # (the per-step gather makes the fast rank wait for the slow one: both clocks read ~40 ms)
    assert pr["ranks"] == 2 and 19.0 <= pr["min"] <= pr["max"] and pr["max"] >= 39.0 and abs(pr["max"] - out["ms_per_step"]) < 1e-6
    assert abs(out["value"] - 8 * 3 / (out["ms_per_step"] * 3e-3)) < 1e-6 * out["value"]


def test_value_is_the_median_of_several_timed_regions():
    """--regions R (default 5): R back-to-back regions of exactly K steps, each bracketed by barrier + synchronize and reduced
    with MAX over the ranks; `value` / `ms_per_step` are the MEDIAN region's (ms_per_step x steps is one measured region),
    value_min / value_max and every region's figure sit beside it; the warm-up runs once."""
    lines = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4", "--regions", "3"])
    out = json.loads(lines[0])
    assert out["stub"]["calls_rank0"] == 1 + 3 * 2 + 1     # W + R x K + the recorded step
    r = out["regions"]
    assert r["count"] == 3 and r["steps_each"] == 2 and len(r["images_per_sec"]) == 3
    assert out["value_min"] == min(r["images_per_sec"]) and out["value_max"] == max(r["images_per_sec"])
    assert out["value"] == sorted(r["images_per_sec"])[1] and out["value_min"] <= out["value"] <= out["value_max"]
    assert abs(out["value"] - 8 * 2 / (out["ms_per_step"] * 2e-3)) < 1e-6 * out["value"]
    assert out["ms_per_step"] >= 39.0                      # the slow rank's 40 ms per step in every region
    pr = out["distributed"]["per_rank_ms_per_step"]
    assert abs(pr["max"] - out["ms_per_step"]) < 1e-6      # the ranks' own times are the median region's


def test_single_gpu_runs_in_process():
    lines = _run(["--gpus", "1", "--steps", "2", "--warmup", "0", "--regions", "1"])
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["stub"]["calls_rank0"] == 3 and not out["stub"]["dist_initialized"]     # W + K + the recorded step
    assert out["distributed"]["process_group_size"] == 1 and out["distributed"]["backend"] is None and "gathered_images" not in out["distributed"]
    assert out["stub"]["env"]["WORLD_SIZE"] is None


def test_under_torch_distributed_run():
    """The driver's N > 1 form: python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    lines = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--regions", "1"],
                 launcher=["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                           "--master-port", str(port)])
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["stub"]["calls_rank0"] == 4 and out["stub"]["env"]["MASTER_PORT"] == str(port)
    assert out["distributed"]["gathered_images"] == out["config"]["global_batch"]


def test_world_size_mismatch_is_an_error_not_an_assert():
    env = {k: v for k, v in os.environ.items()}
    env.update(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="1")
    p = subprocess.run([sys.executable, "-c", _DRIVER % (ROOT, os.path.join(ROOT, "tests")), "--gpus", "4"], env=env,
                       capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert p.returncode != 0 and "world size" in p.stderr
