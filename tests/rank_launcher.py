"""Starts the rank processes of the two-process GPU tests.  It is itself started by conftest.py BEFORE pytest has touched
the GPU and never touches it: a process that has initialised the GPU must not fork-and-exec on the GPU boxes, so the test
process asks this one (a line of JSON on stdin) and reads the ranks' exit codes and output back (a line of JSON on stdout).

request:  {"script": "...py", "args": [...], "n": 2, "env": {...}, "timeout": 240}
answer:   {"rc": [..], "out": ["...", "..."], "timed_out": false, "seconds": 12.3}
"""
import json
import os
import socket
import subprocess
import sys
import tempfile
import time


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run(req: dict) -> dict:
    n = int(req.get("n", 2))
    port = free_port()
    procs, logs = [], []
    t0 = time.time()
    for rank in range(n):
        env = dict(os.environ)
        env.update({k: str(v) for k, v in req.get("env", {}).items()})
        env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1",
                    "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
        log = tempfile.TemporaryFile(mode="w+")
        logs.append(log)
        procs.append(subprocess.Popen([sys.executable, req["script"]] + [str(a) for a in req.get("args", [])], env=env,
                                      stdout=log, stderr=subprocess.STDOUT, stdin=subprocess.DEVNULL))
    deadline = t0 + float(req.get("timeout", 240))
    timed_out = False
    while any(p.poll() is None for p in procs):
        if time.time() > deadline:
            timed_out = True
            for p in procs:                      # exactly the processes started here
                if p.poll() is None:
                    p.kill()
            break
        time.sleep(0.05)
    rcs = [p.wait() for p in procs]
    outs = []
    for log in logs:
        log.seek(0)
        outs.append(log.read()[-20000:])
        log.close()
    return {"rc": rcs, "out": outs, "timed_out": timed_out, "seconds": time.time() - t0}


def main():
    for line in sys.stdin:
        line = line.strip()
        if not line:
            continue
        if line == "quit":
            break
        try:
            ans = run(json.loads(line))
        except Exception as exc:      # noqa: BLE001
            ans = {"error": repr(exc)}
        sys.stdout.write(json.dumps(ans) + "\n")
        sys.stdout.flush()


if __name__ == "__main__":
    main()
