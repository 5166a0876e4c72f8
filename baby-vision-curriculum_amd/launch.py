"""One process per GPU on one node, started from a plain `python script.py --gpus N` command line.

The reference starts its ranks with ``mp.spawn(DDP_process, args=(world_size, args), nprocs=world_size)``
(pretraining/generative/pretrain_videomae.py:509-513) from a parent that has not touched the GPU.  The same contract here,
with fresh child INTERPRETERS instead of forked/spawned functions: the parent never initialises HIP (a process that has must
not be replaced or re-executed on this pool), picks an OS-assigned rendezvous port, exports the variables
``torch.distributed`` / torchrun use (RANK, LOCAL_RANK, WORLD_SIZE, LOCAL_WORLD_SIZE, MASTER_ADDR, MASTER_PORT) and waits;
a rank that fails takes the others down (by PID) and its exit code becomes the launcher's.
"""
import os
import socket
import subprocess
import sys
import time

LAUNCHED_ENV = "BVC_LAUNCHED_BY_PARENT"


def free_port(host="127.0.0.1"):
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind((host, 0))
        return s.getsockname()[1]


def under_launcher(env=None):
    """True inside a rank started by torchrun / spawn_ranks (the rendezvous variables are present)."""
    env = os.environ if env is None else env
    return "RANK" in env and "WORLD_SIZE" in env


def visible_gpus():
    """GPUs this process may use.  torch.cuda.device_count() does not create a HIP context on this image."""
    import torch
    return torch.cuda.device_count()


def rank_env(rank, world, port, base=None, addr="127.0.0.1"):
    env = dict(os.environ if base is None else base)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
                "MASTER_ADDR": addr, "MASTER_PORT": str(port), LAUNCHED_ENV: "1",
                # dmabuf IPC only on this pool: RCCL's intra-node transport needs it in every rank
                "HSA_ENABLE_IPC_MODE_LEGACY": env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
    return env


def spawn_ranks(argv, nprocs, python=None, port=None, timeout=None, check_gpus=True, poll=0.2):
    """Start `python argv...` nprocs times (rank r gets RANK=LOCAL_RANK=r) and wait.  Returns the launcher's exit code:
    0 when every rank returned 0, otherwise the first non-zero code seen (the remaining ranks are terminated)."""
    if nprocs < 1:
        raise ValueError("spawn_ranks: nprocs must be >= 1")
    if check_gpus:
        have = visible_gpus()
        if have < nprocs:
            print(f"launch: {nprocs} ranks requested but only {have} GPU(s) visible - not starting", file=sys.stderr, flush=True)
            return 2
    port = free_port() if port is None else port
    python = python or sys.executable
    procs = [subprocess.Popen([python, *argv], env=rank_env(r, nprocs, port)) for r in range(nprocs)]
    t0 = time.time()
    rc = 0
    try:
        live = list(procs)
        while live:
            for p in list(live):
                code = p.poll()
                if code is None:
                    continue
                live.remove(p)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
            if rc != 0 or (timeout is not None and time.time() - t0 > timeout):
                if rc == 0:
                    rc = 124
                    print(f"launch: ranks still running after {timeout} s - terminating them", file=sys.stderr, flush=True)
                break
            time.sleep(poll)
    finally:
        for p in procs:              # exact PIDs only
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    return rc
