"""The one-node launcher behind `python bench.py --gpus N` (baby-vision-curriculum_amd/launch.py), on CPU: children are started
with the rendezvous environment torch.distributed expects, a failing rank takes the launch down with its exit code, and asking
for more ranks than visible GPUs ends with a message instead of a hang.  Reference launch model: mp.spawn(DDP_process,
nprocs=world_size), pretraining/generative/pretrain_videomae.py:509-513."""
import importlib.util
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch():
    spec = importlib.util.spec_from_file_location("bvc_launch", os.path.join(ROOT, "baby-vision-curriculum_amd", "launch.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


RANK_SCRIPT = r'''
import json, os, sys, time
out, fail_rank = sys.argv[1], int(sys.argv[2])
rank = int(os.environ["RANK"])
keys = ["RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "BVC_LAUNCHED_BY_PARENT",
        "HSA_ENABLE_IPC_MODE_LEGACY"]
json.dump({k: os.environ.get(k) for k in keys} | {"argv": sys.argv[1:]}, open(os.path.join(out, f"rank{rank}.json"), "w"))
if rank == fail_rank:
    sys.exit(3)
if fail_rank >= 0:
    time.sleep(60)      # the launcher must not wait for this
'''


def test_children_get_the_rendezvous_environment(tmp_path):
    L = _launch()
    script = tmp_path / "rank.py"
    script.write_text(RANK_SCRIPT)
    rc = L.spawn_ranks([str(script), str(tmp_path), "-1"], 3, check_gpus=False)
    assert rc == 0
    envs = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(3)]
    for r, e in enumerate(envs):
        assert e["RANK"] == str(r) and e["LOCAL_RANK"] == str(r)
        assert e["WORLD_SIZE"] == "3" and e["LOCAL_WORLD_SIZE"] == "3"
        assert e["MASTER_ADDR"] == "127.0.0.1" and e["BVC_LAUNCHED_BY_PARENT"] == "1"
        assert e["HSA_ENABLE_IPC_MODE_LEGACY"] is not None
        assert e["argv"] == [str(tmp_path), "-1"]
    ports = {e["MASTER_PORT"] for e in envs}
    assert len(ports) == 1 and 1024 < int(ports.pop()) < 65536


def test_failing_rank_ends_the_launch_with_its_code(tmp_path):
    import time
    L = _launch()
    script = tmp_path / "rank.py"
    script.write_text(RANK_SCRIPT)
    t0 = time.time()
    rc = L.spawn_ranks([str(script), str(tmp_path), "1"], 2, check_gpus=False)
    assert rc == 3
    assert time.time() - t0 < 30          # rank 0 (sleeping) was terminated, not waited for


def test_more_ranks_than_gpus_is_refused(tmp_path, capsys):
    L = _launch()
    have = L.visible_gpus()
    rc = L.spawn_ranks([str(tmp_path / "never_started.py")], have + 1)
    assert rc == 2
    assert "not starting" in capsys.readouterr().err


def test_bench_gpus_flag_self_launches_or_refuses():
    # `python bench.py --gpus N` without torchrun's environment becomes the launcher; with fewer than N GPUs visible (none on
    # the build box) it exits non-zero with a message, before anything touches a GPU
    import torch
    n = max(2, torch.cuda.device_count() + 1)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 2, p.stderr[-2000:]
    assert f"{n} ranks requested" in p.stderr and p.stdout.strip() == ""
