"""Child-process launcher for the GPU tests.

The pytest process initialises the GPU early (torch.cuda.is_available() in conftest), and a process that has done so must not
fork + exec another program on the GPU boxes (the pool refuses it).  conftest therefore starts THIS script once, before anything
touches the GPU; it never imports torch, waits for one JSON request per line on stdin -- {"cmd": [...], "env": {...},
"timeout": s} --, runs the command as its own child and answers with {"rc", "stdout", "stderr"} on stdout.
"""
import json
import os
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
            out = subprocess.run(req['cmd'], env=env, capture_output=True, text=True, timeout=req.get('timeout', 1200),
                                 cwd=req.get('cwd'))
            ans = {'rc': out.returncode, 'stdout': out.stdout[-200000:], 'stderr': out.stderr[-20000:]}
        except subprocess.TimeoutExpired as e:
            ans = {'rc': -9, 'stdout': (e.stdout or b'').decode(errors='replace')[-20000:] if isinstance(e.stdout, bytes) else (e.stdout or ''),
                   'stderr': 'timeout'}
        except Exception as e:                      # noqa: BLE001 -- report, never die
            ans = {'rc': -1, 'stdout': '', 'stderr': repr(e)}
        sys.stdout.write(json.dumps(ans) + '\n')
        sys.stdout.flush()


if __name__ == '__main__':
    main()
