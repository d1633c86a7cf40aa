"""bench.py's LineGuard (N > 1: the optional measurements after the timed region must not take the headline line with them): a
process whose main thread is stuck still gets its line out on the launcher's SIGTERM (exit code 1) and at the deadline (exit code
0); a disarmed guard leaves the process alone.  CPU only, sub-processes."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(body, timeout=60):
    code = "import sys, os, time, signal\nsys.path.insert(0, %r)\nimport bench\n" % ROOT + textwrap.dedent(body)
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=timeout)


def test_line_gets_out_on_sigterm_while_the_main_thread_is_stuck():
    p = _run("""
        g = bench.LineGuard(0, 120)
        g.make_line = lambda why: '{"value": 1.0, "cut": "%s"}' % why
        import threading
        threading.Timer(0.5, lambda: os.kill(os.getpid(), signal.SIGTERM)).start()
        # the main thread sits in C code that does not return (a collective that never completes): a lock nobody releases
        l = threading.Lock(); l.acquire(); l.acquire()
        """)
    assert p.returncode == 1, (p.returncode, p.stdout, p.stderr)
    assert p.stdout.strip().startswith('{"value": 1.0, "cut": "SIGTERM'), p.stdout


def test_line_gets_out_at_the_deadline_with_code_0_and_other_ranks_print_nothing():
    p = _run("""
        g = bench.LineGuard(0, 1)
        g.make_line = lambda why: '{"value": 2.0, "cut": "%s"}' % why
        import threading
        l = threading.Lock(); l.acquire(); l.acquire()
        """)
    assert p.returncode == 0, (p.returncode, p.stdout, p.stderr)
    assert p.stdout.strip() == '{"value": 2.0, "cut": "no end after 1 s"}', p.stdout
    p = _run("""
        g = bench.LineGuard(3, 1)
        import threading
        l = threading.Lock(); l.acquire(); l.acquire()
        """)
    assert p.returncode == 0 and p.stdout.strip() == "", (p.returncode, p.stdout)


def test_disarmed_guard_leaves_the_process_alone():
    p = _run("""
        g = bench.LineGuard(0, 1)
        g.make_line = lambda why: "NOT THIS"
        g.disarm()
        time.sleep(1.5)
        print("the ordinary line")
        """)
    assert p.returncode == 0 and p.stdout.strip() == "the ordinary line", (p.returncode, p.stdout, p.stderr)
    # and SIGTERM has its default action again
    p = _run("""
        g = bench.LineGuard(0, 60)
        g.disarm()
        os.kill(os.getpid(), signal.SIGTERM)
        time.sleep(5)
        print("still here")
        """)
    assert p.returncode == -15 and "still here" not in p.stdout, (p.returncode, p.stdout)


def test_line_gets_out_when_an_optional_phase_raises():
    p = _run("""
        g = bench.LineGuard(0, 60)
        g.make_line = lambda why: '{"value": 3.0, "cut": "%s"}' % why
        try:
            raise RuntimeError("Connection closed by peer")
        except BaseException as e:
            g.failed(e)
        print("not reached")
        """)
    assert p.returncode == 1 and p.stdout.strip().startswith('{"value": 3.0, "cut": "an optional phase raised RuntimeError'), (p.returncode, p.stdout)
