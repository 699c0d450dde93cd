"""Starts one process per rank on this node without torch.distributed.run.

`python bench.py --gpus N` is expected to run N ranks by itself.  The parent process that
does the starting must never touch the GPU (a process that has initialised HIP must not be
replaced or forked into ranks), so it only sets the rendezvous variables that
torch.distributed's env:// initialisation reads -- RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR,
MASTER_PORT -- starts N fresh interpreters, relays rank 0's standard output and waits.  If one
rank fails the others are ended (by their own process ids) and the failure is the result.
"""
import os
import socket
import subprocess
import sys
import threading
import time


def free_port():
    """A TCP port nobody listens on right now (rendezvous of the ranks on 127.0.0.1)."""
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    try:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]
    finally:
        s.close()


def rank_environment(rank, world, port, base=None):
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    # the IPC mode the host driver supports (needed by RCCL between processes)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def spawn_ranks(world, argv, port=None, env=None, timeout=None, out=None, err=None):
    """Runs `argv` (a full command line) as `world` ranks and returns the exit code of the job:
    0 if every rank ended with 0, else the first non-zero code seen (124 on timeout).  Rank 0's
    standard output goes to `out` (default sys.stdout) untouched; every rank's standard error and
    the other ranks' standard output go to `err` (default sys.stderr), prefixed with the rank."""
    out = sys.stdout if out is None else out
    err = sys.stderr if err is None else err
    port = free_port() if port is None else port
    procs = []
    for r in range(world):
        procs.append(subprocess.Popen(argv, env=rank_environment(r, world, port, env), stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True, bufsize=1))
    lock = threading.Lock()

    def relay(stream, sink, prefix):
        for line in stream:
            with lock:
                sink.write(prefix + line)
                sink.flush()

    threads = []
    for r, p in enumerate(procs):
        threads.append(threading.Thread(target=relay, args=(p.stdout, out if r == 0 else err,
                                                            "" if r == 0 else "[rank %d] " % r), daemon=True))
        threads.append(threading.Thread(target=relay, args=(p.stderr, err, "[rank %d] " % r), daemon=True))
    for t in threads:
        t.start()

    deadline = None if timeout is None else time.monotonic() + timeout
    code = 0
    alive = set(range(world))
    while alive:
        for r in sorted(alive):
            rc = procs[r].poll()
            if rc is None:
                continue
            alive.discard(r)
            if rc != 0 and code == 0:
                code = rc
        timed_out = deadline is not None and time.monotonic() > deadline
        if timed_out and code == 0:
            code = 124
        if code != 0 and alive:
            # a rank failed (or time ran out): the others would wait for it in a collective for ever
            for r in alive:
                procs[r].terminate()
            t_end = time.monotonic() + 10
            for r in list(alive):
                try:
                    procs[r].wait(timeout=max(0.1, t_end - time.monotonic()))
                except subprocess.TimeoutExpired:
                    procs[r].kill()
                    procs[r].wait()
            alive.clear()
        if alive:
            time.sleep(0.05)
    for t in threads:
        t.join(timeout=5)
    return code
