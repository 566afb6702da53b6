import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference mounted (build container only)")


def _gpu_present() -> bool:
    return os.path.exists("/dev/kfd")


# The two-process GPU tests need their rank processes started by a process that has NOT touched the GPU (a GPU box refuses
# fork-and-exec from one that has): tests/rank_launcher.py is started here, at configure time -- before any test module is
# imported -- and asked later through its pipes.  Only on a machine with a GPU.
_launcher = None


def pytest_sessionstart(session):
    global _launcher
    if _gpu_present() and _launcher is None:
        import subprocess
        _launcher = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "rank_launcher.py")], stdin=subprocess.PIPE,
                                     stdout=subprocess.PIPE, text=True, bufsize=1, cwd=ROOT)


def pytest_sessionfinish(session, exitstatus):
    global _launcher
    if _launcher is not None:
        try:
            _launcher.stdin.write("quit\n")
            _launcher.stdin.flush()
            _launcher.wait(timeout=10)
        except Exception:
            _launcher.kill()
        _launcher = None


@pytest.fixture(scope="session")
def rank_launcher():
    """run(script, args, n, env, timeout) -> {"rc": [...], "out": [...], "timed_out": bool} through the helper process."""
    import json
    if _launcher is None:
        pytest.skip("no GPU: the rank launcher was not started")

    def run(script, args=(), n=2, env=None, timeout=240):
        req = {"script": os.path.join(ROOT, "tests", script), "args": list(args), "n": n, "env": env or {}, "timeout": timeout}
        _launcher.stdin.write(json.dumps(req) + "\n")
        _launcher.stdin.flush()
        line = _launcher.stdout.readline()
        assert line, "the rank launcher died"
        ans = json.loads(line)
        assert "error" not in ans, ans
        return ans
    return run


def pytest_collection_modifyitems(config, items):
    markexpr = config.getoption("-m") or ""
    for item in items:
        if "gpu" in item.keywords and "gpu" not in markexpr.replace("not gpu", ""):
            if not _gpu_present():
                item.add_marker(pytest.mark.skip(reason="no GPU in this container"))


@pytest.fixture(scope="session")
def ctx():
    from padne_amd import solver
    c = solver.get_context()
    yield c


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture
def switches(monkeypatch):
    """The PADNE_* environment switches of the library, which reads them ONCE per context (at creation): set / unset one and
    every live context reads them again; on the way out the environment is restored and read once more."""
    from padne_amd import _hip

    class Switches:
        def set(self, name, value="1"):
            monkeypatch.setenv(name, str(value))
            _hip.reload_options_everywhere()

        def unset(self, name):
            monkeypatch.delenv(name, raising=False)
            _hip.reload_options_everywhere()
    yield Switches()
    monkeypatch.undo()
    _hip.reload_options_everywhere()
