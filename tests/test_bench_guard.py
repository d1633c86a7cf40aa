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


def test_a_line_that_cannot_be_made_falls_back_to_the_one_serialised_at_arming():
    """the helper thread makes the line while the main thread may be writing to its records: if that raises, the line serialised
    before the optional phases is printed instead"""
    p = _run("""
        g = bench.LineGuard(0, 1)
        g.fallback = '{"value": 3.0, "cut": "armed"}'
        def boom(why):
            raise RuntimeError("dictionary changed size during iteration")
        g.make_line = boom
        import threading
        l = threading.Lock(); l.acquire(); l.acquire()
        """)
    assert p.returncode == 0, (p.returncode, p.stdout, p.stderr)
    assert p.stdout.strip() == '{"value": 3.0, "cut": "armed"}' and "printing the one made before" in p.stderr, (p.stdout, p.stderr)


def test_gpus_n_without_a_launcher_starts_its_own_ranks():
    """`python bench.py --gpus 2` by itself (the shape of the driver's 1-GPU command): the process becomes the launcher -- it
    starts torch.distributed.run as a CHILD, never imports torch itself, and returns the ranks' exit code.  In a container
    without a GPU both ranks stop at "no GPU visible", which is what this CPU test sees."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=300)
    import torch
    if torch.cuda.is_available():
        return                                                       # (on a GPU box tests/test_gpu_bench.py covers the real run)
    assert p.returncode != 0
    assert p.stderr.count("no GPU visible; the HIP path has no CPU fallback") >= 1, p.stderr[-2000:]
    assert "must be launched with" not in p.stderr
    # the launcher itself: free port on 127.0.0.1, same arguments, child process (read off the source: no exec)
    src = open(os.path.join(ROOT, "bench.py")).read()
    body = src[src.index("def launch_ranks"):src.index("def main()")]
    assert "subprocess.Popen" in body and "os.exec" not in body and "import torch" not in body
