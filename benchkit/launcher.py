"""`python bench.py --gpus N` without a launcher: bench.py starts the N ranks itself (fresh child processes of bench.py, before
this process touches a GPU) and forwards rank 0's line."""
import json
import os
import socket
import subprocess
import sys
import threading
import time

from .common import BENCH_PY, T0_ENV


def _launch_once(args, n, deadline):
    """One attempt: N fresh children, rank 0's stdout captured.  Returns (rcs, rank-0 stdout, seconds until the first exit)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    t_start = time.time()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, BENCH_PY] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out_box = {}
    reader = threading.Thread(target=lambda: out_box.setdefault("out", procs[0].stdout.read()), daemon=True)
    reader.start()                                           # drain the pipe while polling, or rank 0 blocks on a full one
    first_exit = None
    while True:
        rcs = [p.poll() for p in procs]
        if first_exit is None and any(rc is not None for rc in rcs):
            first_exit = time.time() - t_start
        if all(rc is not None for rc in rcs):
            break
        # a rank died with an error: the others would sit in their next collective until its own timeout
        if any(rc not in (None, 0) for rc in rcs) or time.time() > deadline:
            time.sleep(3.0)                                  # (ranks that are on their way out through the watchdog)
            for p in procs:
                if p.poll() is None:
                    p.kill()
            rcs = [p.wait() for p in procs]
            break
        time.sleep(0.2)
    reader.join(timeout=10)
    return rcs, out_box.get("out") or "", first_exit or 0.0


def self_launch(args, process_start=None) -> int:
    """Spawn N fresh child processes (one rank per GPU) BEFORE this process touches a GPU -- nothing here calls into HIP,
    and no process that has is ever replaced by another program.  Rank 0's stdout is captured; its last JSON line is
    checked (n_gpus == N) and forwarded as this process's single output line.  The children are polled: as soon as one
    exits non-zero the rest are killed, and nothing outlives the job's time budget (+ 20 s for the ranks' own watchdogs
    to print and leave first).  A line rank 0 did print is forwarded even when a rank failed; the status stays non-zero."""
    n = args.gpus
    one_device = os.environ.get("SCONE_ONE_DEVICE") == "1"
    if not args.selftest:
        import torch                               # device_count() does not initialise the GPU on this image
        have = torch.cuda.device_count()
        if have < n and not one_device:
            print(f"bench.py: --gpus {n} but only {have} HIP device(s) visible", file=sys.stderr)
            return 2
    os.environ.setdefault(T0_ENV, repr(process_start if process_start is not None else time.time()))
    deadline = float(os.environ[T0_ENV]) + args.time_budget + 20.0
    rcs, out, first_exit = _launch_once(args, n, deadline)
    if rcs[0] not in (0, None) and '{"metric"' not in out and first_exit < 30.0 and time.time() + 60.0 < deadline:
        # the rendezvous port was picked by bind-then-close: another process may have taken it in between.  One retry
        sys.stderr.write(f"bench.py: ranks exited with {rcs} after {first_exit:.0f} s without a result; retrying once on a new port\n")
        rcs, out, first_exit = _launch_once(args, n, deadline)
    line = None
    for ln in out.splitlines():
        if ln.startswith('{"metric"'):
            line = ln
    res = None
    if line is not None:
        try:
            res = json.loads(line)
        except ValueError:
            res = None
    ok = not any(rcs) and res is not None
    if res is not None and res.get("n_gpus") != n:
        sys.stderr.write(f"bench.py: --gpus {n} but the result line says n_gpus = {res.get('n_gpus')}\n")
        return 1
    if res is not None:
        res["launcher"] = f"bench.py started {n} ranks itself (WORLD_SIZE was unset)"
        if not ok:
            res["launcher"] += f"; ranks exited with {rcs}"
        print(json.dumps(res), flush=True)
        if ok and res.get("hung_stage"):                     # the line is valid, the job is not a clean pass
            sys.stderr.write(f"bench.py: stage '{res['hung_stage']}' hung; the line above holds what was measured before it\n")
            return 4
    if not ok:
        sys.stderr.write(f"bench.py: ranks exited with {rcs}; rank 0 printed {'no' if res is None else 'a'} result line\n")
        if out and res is None:
            sys.stderr.write(out[-2000:])
        return 1
    return 0
