import faulthandler
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
_FAULT_FILE = None


def pytest_configure(config):
    """A run that dies (SIGABRT from the HSA runtime's fault handler or from std::terminate) must NAME the test it died in: the round-5
    driver run ended in 5 KB of extension-module names and nothing else.  pytest's own faulthandler plugin is off (pytest.ini); the
    Python-level traceback of a fatal signal goes to a FILE (gpurun_out/fault_<pid>.txt where that directory exists, else the temp
    directory), and every test's node id goes to stderr before it starts (`pytest_runtest_logstart`), so the last lines of a log are the
    running test and the runtime's own message ("Memory access fault by GPU ... address", "terminate called after ...")."""
    global _FAULT_FILE
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu`)")
    d = os.path.join(ROOT, "gpurun_out")
    if not (os.path.isdir(d) and os.access(d, os.W_OK)):
        import tempfile
        d = tempfile.gettempdir()
    try:
        _FAULT_FILE = open(os.path.join(d, f"fault_{os.getpid()}.txt"), "w")
        faulthandler.enable(file=_FAULT_FILE, all_threads=True)
    except OSError:
        _FAULT_FILE = None


def pytest_unconfigure(config):
    global _FAULT_FILE
    if _FAULT_FILE is not None:
        faulthandler.disable()
        name = _FAULT_FILE.name
        _FAULT_FILE.close()
        _FAULT_FILE = None
        try:
            if os.path.getsize(name) == 0:
                os.remove(name)
        except OSError:
            pass


def pytest_runtest_logstart(nodeid, location):
    if os.environ.get("RE_TEST_NAMES", "1") == "0":
        return
    try:
        os.write(2, f"\n[test] {nodeid}\n".encode())       # (fd 2 directly: not through pytest's capture, nothing buffered when the process dies)
    except OSError:
        pass


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
