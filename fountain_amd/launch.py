"""Starting the N ranks of a one-node multi-GPU job (one process per GPU, torch.distributed over RCCL).

`bench.py --gpus N` and `python -m fountain_amd.render --gpus N` can be started two ways:
  * by a launcher (`python -m torch.distributed.run --nproc-per-node N ...`): WORLD_SIZE / RANK / LOCAL_RANK are in the environment;
  * plainly (`python bench.py --gpus 8`): nothing is in the environment, and the program starts its N ranks itself with
    spawn_ranks() -- as CHILD processes, before this process has imported torch or touched HIP (a process that has initialised the
    GPU must never be replaced by another program on this pool), relays their output and exits with the launcher's code.
This module imports neither torch nor the HIP library."""
import os
import signal
import socket
import subprocess
import sys
import threading


def world_from_env():
    """(world_size, rank, local_rank) from the launcher's environment, or None when this process was started plainly."""
    if "WORLD_SIZE" not in os.environ:
        return None
    return int(os.environ["WORLD_SIZE"]), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def free_port():
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _stop(proc, grace=5.0):
    """Ends the launcher AND its ranks: SIGTERM to the launcher (torch.distributed.run forwards it to its workers and reaps them), a few
    seconds of grace, then SIGKILL to the whole process group the launcher was started in (start_new_session: the ranks are in it) --
    a SIGKILL to the launcher alone cannot be handled and would leave the ranks running with their GPUs."""
    if proc.poll() is None:
        try:
            proc.terminate()
            proc.wait(timeout=grace)
        except Exception:
            pass
    try:
        os.killpg(proc.pid, signal.SIGKILL)          # the group outlives its leader while any rank is alive; gone already: ESRCH
    except (ProcessLookupError, PermissionError):
        pass
    try:
        proc.wait(timeout=grace)
    except Exception:
        pass


def spawn_ranks(n_ranks, program_args, module=None, script=None, env=None, timeout=None, json_only=False):
    """Runs `python -m torch.distributed.run --nnodes=1 --nproc-per-node n_ranks ... <script | -m module> program_args` as a child
    process in its own session.  The ranks' stdout is relayed line by line (rank 0 prints the result line; with json_only every other
    line goes to stderr so that stdout carries the result line alone), stderr passes through.  Returns the launcher's exit code:
    non-zero when any rank failed, 124 when `timeout` seconds passed (the deadline holds while a hung rank keeps the pipe open: a
    watchdog thread stops the job, which ends the read loop).  On any exception (KeyboardInterrupt included) the job is stopped first."""
    assert (module is None) != (script is None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(int(n_ranks)),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port())]
    cmd += ["-m", module] if module else [script]
    cmd += list(program_args)
    child_env = dict(os.environ if env is None else env)
    child_env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL between processes needs it on this driver
    child_env.setdefault("OMP_NUM_THREADS", "1")
    proc = subprocess.Popen(cmd, env=child_env, stdout=subprocess.PIPE, text=True, bufsize=1, start_new_session=True)
    expired = threading.Event()
    done = threading.Event()

    def watchdog():
        if not done.wait(timeout):
            expired.set()
            _stop(proc)

    if timeout is not None:
        threading.Thread(target=watchdog, daemon=True).start()
    try:
        for line in proc.stdout:
            out = sys.stdout if (not json_only or line.lstrip().startswith("{")) else sys.stderr     # e.g. gloo's connection banners
            out.write(line)
            out.flush()
        rc = proc.wait()
        return 124 if expired.is_set() else rc
    except BaseException:
        _stop(proc)
        raise
    finally:
        done.set()
