"""Child-process launcher for the GPU tests.

The pytest process initialises the GPU early (torch.cuda.is_available() in conftest), and a process that has done so must not
fork + exec another program on the GPU boxes (the pool refuses it).  conftest therefore starts THIS script once, before anything
touches the GPU; it never imports torch, waits for one JSON request per line on stdin -- {"cmd": [...], "env": {...},
"timeout": s} --, runs the command as its own child and answers with {"rc", "stdout", "stderr"} on stdout.
"""
import json
import os
import signal
import subprocess
import sys


def main():
    for line in sys.stdin:
        line = line.strip()
        if not line:
            continue
        req = json.loads(line)
        if req.get('quit'):
            break
        env = dict(os.environ)
        for k in req.get('unset', []):
            env.pop(k, None)
        env.update(req.get('env', {}))
        try:
            # the command (python -m torch.distributed.run ...) has grandchildren -- the ranks -- that inherit the pipes, hold the GPU
            # and the rendezvous port: run it in a session of its own and, on a timeout, kill the whole process GROUP before
            # draining the pipes (killing only the direct child would leave communicate() blocked on pipes the ranks keep open)
            proc = subprocess.Popen(req['cmd'], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=req.get('cwd'),
                                    start_new_session=True)
            try:
                so, se = proc.communicate(timeout=req.get('timeout', 1200))
                ans = {'rc': proc.returncode, 'stdout': so[-200000:], 'stderr': se[-20000:]}
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(proc.pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass
                try:
                    so, se = proc.communicate(timeout=30)
                except subprocess.TimeoutExpired:
                    so, se = '', ''
                ans = {'rc': -9, 'stdout': (so or '')[-20000:], 'stderr': 'timeout: process group killed\n' + (se or '')[-4000:]}
        except Exception as e:                      # noqa: BLE001 -- report, never die
            ans = {'rc': -1, 'stdout': '', 'stderr': repr(e)}
        sys.stdout.write(json.dumps(ans) + '\n')
        sys.stdout.flush()


if __name__ == '__main__':
    main()
