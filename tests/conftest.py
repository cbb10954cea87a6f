"""Shared pieces of the test suite, and what makes a GPU run attributable.

pytest.ini runs every test in ONE xdist worker (a child python process; the pytest the driver starts never touches the
GPU).  This file adds, for every process of a run:
  * gpurun_out/pytest_progress.log -- one line when a test starts and one with its outcome, flushed at once: the last
    START line without an outcome is the test that was running when a process died;
  * gpurun_out/pytest_fault_<pid>.log -- faulthandler's dump of a fatal signal (pytest's own plugin is off: its dump, with
    182 extension-module names, used to be the whole tail of the log);
  * at the end of a failing run, the tail of the progress log and the head of any fault dump in the terminal summary.
"""
import faulthandler
import os
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
LOGDIR = os.path.join(ROOT, "gpurun_out")
PROGRESS = os.path.join(LOGDIR, "pytest_progress.log")
_fault_file = None


def _is_worker(config):
    return hasattr(config, "workerinput")


def _progress(line):
    try:
        with open(PROGRESS, "a") as f:
            f.write(f"{time.strftime('%H:%M:%S')} pid {os.getpid()} {line}\n")
            f.flush()
            os.fsync(f.fileno())
    except OSError:
        pass


def pytest_configure(config):
    global _fault_file
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    try:
        os.makedirs(LOGDIR, exist_ok=True)
        if not _is_worker(config):
            for name in os.listdir(LOGDIR):           # a new run: drop the previous run's records
                if name == "pytest_progress.log" or name.startswith("pytest_fault_"):
                    os.remove(os.path.join(LOGDIR, name))
        _fault_file = open(os.path.join(LOGDIR, f"pytest_fault_{os.getpid()}.log"), "w")
        faulthandler.enable(file=_fault_file, all_threads=True)
    except OSError:
        _fault_file = None
    _progress(("worker" if _is_worker(config) else "session") + " up: " + " ".join(map(str, config.invocation_params.args)))


def pytest_runtest_logstart(nodeid, location):
    # (in an xdist run this hook fires in the worker, right before the test, and in the controller when the first report of
    # the test arrives: only the process that runs the test writes the START line)
    if os.environ.get("PYTEST_XDIST_WORKER") or not _xdist_controller:
        _progress(f"START {nodeid}")
        if not os.environ.get("PYTEST_XDIST_WORKER"):
            sys.__stderr__.write(f"\n[running] {nodeid}\n")
            sys.__stderr__.flush()


_xdist_controller = False


def pytest_sessionstart(session):
    global _xdist_controller
    cfg = session.config
    _xdist_controller = (not _is_worker(cfg)) and bool(getattr(cfg.option, "numprocesses", None))


def pytest_runtest_logreport(report):
    if os.environ.get("PYTEST_XDIST_WORKER") or not _xdist_controller:
        if report.when == "call" or report.outcome != "passed":
            _progress(f"{report.outcome.upper()} {report.nodeid} [{report.when}] {getattr(report, 'duration', 0.0):.2f}s")


def pytest_sessionfinish(session, exitstatus):
    _progress(f"{'worker' if _is_worker(session.config) else 'session'} finished, exit status {int(exitstatus)}")
    if _fault_file is not None:
        _fault_file.flush()


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    if _is_worker(config) or int(exitstatus) == 0:
        return
    tr = terminalreporter
    try:
        lines = open(PROGRESS).read().splitlines()
    except OSError:
        lines = []
    if lines:
        tr.section("last lines of gpurun_out/pytest_progress.log")
        for ln in lines[-12:]:
            tr.write_line(ln)
    try:
        names = sorted(n for n in os.listdir(LOGDIR) if n.startswith("pytest_fault_"))
    except OSError:
        names = []
    for name in names:
        try:
            text = open(os.path.join(LOGDIR, name)).read()
        except OSError:
            continue
        if not text.strip():
            continue
        tr.section(f"fatal signal in a test process: head of gpurun_out/{name}")
        head = text.split("Extension modules:")[0].splitlines()          # (the stacks, not the module list)
        first = [ln for ln in head[:3] if ln.strip()]
        cur = next((i for i, ln in enumerate(head) if ln.startswith("Current thread")), 0)
        for ln in first[:2] + head[cur:cur + 12]:
            tr.write_line(ln)


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))

    return load


def run_isolated(script_args, timeout=300, env=None):
    """Run `python <script_args...>` as a fresh child process (for tests that open an RCCL group or start ranks: whatever the
    runtime does to that process, the test sees a return code and the output's tail).  Returns the CompletedProcess."""
    import subprocess
    full_env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    full_env.update(env or {})
    return subprocess.run([sys.executable] + list(script_args), env=full_env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
