import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)


_LAUNCHER = None


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # Multi-process GPU tests run their ranks through a launcher process started HERE, before this process touches the GPU
    # (tests/launcher.py: a GPU-initialised process must not fork + exec on the GPU boxes).  device_count() does not initialise.
    global _LAUNCHER
    try:
        import torch
        have_gpu = torch.cuda.device_count() > 0
    except Exception:
        have_gpu = False
    if have_gpu and _LAUNCHER is None:
        import subprocess
        _LAUNCHER = subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', 'launcher.py')], stdin=subprocess.PIPE,
                                     stdout=subprocess.PIPE, text=True, bufsize=1)


def pytest_unconfigure(config):
    global _LAUNCHER
    if _LAUNCHER is not None:
        try:
            _LAUNCHER.stdin.write('{"quit": true}\n')
            _LAUNCHER.stdin.flush()
            _LAUNCHER.wait(timeout=10)
        except Exception:
            _LAUNCHER.kill()
        _LAUNCHER = None


@pytest.fixture
def run_child():
    """run_child(cmd, env=None, unset=(), timeout=1200) -> dict(rc, stdout, stderr): the command runs as a child of the launcher."""
    import json

    def run(cmd, env=None, unset=(), timeout=1200):
        if _LAUNCHER is None:
            pytest.skip('no launcher process (no GPU at configure time)')
        _LAUNCHER.stdin.write(json.dumps({'cmd': list(cmd), 'env': env or {}, 'unset': list(unset), 'timeout': timeout, 'cwd': ROOT}) + '\n')
        _LAUNCHER.stdin.flush()
        # a deadline on the answer: the launcher kills a command's process group at `timeout`, so an answer later than that plus a
        # grace period means the launcher itself is stuck -- fail this test instead of hanging the session
        import select
        ready, _, _ = select.select([_LAUNCHER.stdout], [], [], timeout + 120)
        if not ready:
            _LAUNCHER.kill()
            raise RuntimeError(f'launcher process gave no answer within {timeout + 120} s: killed')
        line = _LAUNCHER.stdout.readline()
        if not line:
            raise RuntimeError('launcher process died')
        return json.loads(line)
    return run


def pytest_collection_modifyitems(config, items):
    """GPU tests skip themselves cleanly when no device is present (CPU container)."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


if os.environ.get('STRIBOR_TEST_POISON') == '1':
    # opt-in memory hygiene run of the whole suite: torch.empty / empty_like hand out NaN-filled float tensors on the GPU (set
    # STRIBOR_POISON_SCRATCH=1 as well for the library's scratch), so an op that reads what it never wrote turns a test red
    # instead of depending on what the caching allocator returns
    import torch as _torch
    _e0, _el0 = _torch.empty, _torch.empty_like

    def _fill(t):
        return t.fill_(float('nan')) if t.is_floating_point() and t.is_cuda else t
    _torch.empty = lambda *a, **k: _fill(_e0(*a, **k))
    _torch.empty_like = lambda *a, **k: _fill(_el0(*a, **k))

    @pytest.fixture(autouse=True)
    def _no_pending_device_flag():
        """(poison runs) a range / discriminant flag left behind by a test would surface in a later one: name the culprit"""
        yield
        if _torch.cuda.is_available():
            import stribor_amd as _st
            _torch.cuda.synchronize()
            _st.check_errors()
