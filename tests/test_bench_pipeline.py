"""bench.py's two-stage step pipeline (run_steps): ordering guarantees, results and error propagation.
Pure host logic, no GPU."""
import importlib.util
import os
import threading
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def run_steps():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.run_steps


@pytest.mark.parametrize("pipelined", [False, True])
def test_every_step_is_searched_then_finished_in_order(run_steps, pipelined):
    log, lock = [], threading.Lock()

    def search(k):
        with lock:
            log.append(("s", k))

    def finish(k):
        time.sleep(0.002 * (k % 3))
        with lock:
            log.append(("f", k))
        return k * k

    out = run_steps(range(2, 9), search, finish, pipelined)
    assert out == [k * k for k in range(2, 9)]
    pos = {e: i for i, e in enumerate(log)}
    for k in range(2, 9):
        assert pos[("s", k)] < pos[("f", k)]                       # a step is finished after its search
        if k + 1 < 9:
            assert pos[("s", k)] < pos[("s", k + 1)] and pos[("f", k)] < pos[("f", k + 1)]
        if k + 2 < 9:
            assert pos[("f", k)] < pos[("s", k + 2)]               # steps k and k+2 share an engine
    if not pipelined:
        assert [e for e in log] == [x for k in range(2, 9) for x in (("s", k), ("f", k))]


def test_search_of_the_next_step_overlaps_the_finish_of_this_one(run_steps):
    overlap = []
    busy = threading.Event()

    def search(k):
        overlap.append(busy.is_set())
        time.sleep(0.01)

    def finish(k):
        busy.set()
        time.sleep(0.02)
        busy.clear()
        return k

    run_steps(range(4), search, finish, True)
    assert any(overlap[1:])


def test_an_error_in_the_finishing_thread_reaches_the_caller(run_steps):
    def finish(k):
        if k == 1:
            raise ValueError("boom")
        return k

    with pytest.raises(ValueError, match="boom"):
        run_steps(range(5), lambda k: time.sleep(0.001), finish, True)


def test_split_search_queues_the_next_step_before_collecting_this_one(run_steps):
    """begin(k+1, k) is issued before search(k) collects, after step k-1 is finished (engine k+1 mod 2 is free) --
    and every step is begun exactly once, in order."""
    log, lock = [], threading.Lock()

    def begin(k, prev):
        with lock:
            log.append(("b", k, prev))

    def search(k):
        time.sleep(0.003)
        with lock:
            log.append(("s", k))

    def finish(k):
        time.sleep(0.001)
        with lock:
            log.append(("f", k))
        return k

    def stage(k):
        with lock:
            log.append(("st", k))

    out = run_steps(range(3, 10), search, finish, True, stage, 2, begin)
    assert out == list(range(3, 10))
    pos = {e[:2]: i for i, e in enumerate(log)}
    begun = [e for e in log if e[0] == "b"]
    assert [e[1] for e in begun] == list(range(3, 10)) and begun[0][2] is None and all(e[2] == e[1] - 1 for e in begun[1:])
    for k in range(3, 10):
        assert pos[("b", k)] < pos[("s", k)] < pos[("f", k)]
        if k + 1 < 10:
            assert pos[("b", k + 1)] < pos[("s", k)]                 # the next launch is queued before this one is collected
        if k + 2 < 10:
            assert pos[("f", k)] < pos[("b", k + 2)]                 # steps k and k+2 share an engine
            assert pos[("st", k + 2)] < pos[("b", k + 2)]            # and k+2 is staged before it is begun


@pytest.mark.parametrize("with_begin", [False, True])
def test_three_engines_with_a_staging_thread(run_steps, with_begin):
    """lookahead 3 + stager: step k lives on engine k % 3, stage(k+3) runs on its own thread after finish(k) and may
    overlap finish(k+1); a step is begun / searched only after it is staged; everything completes inside the call."""
    log, lock = [], threading.Lock()
    in_finish = threading.Event()
    overlapped = []

    def add(*e):
        with lock:
            log.append(e)

    def begin(k, prev):
        add("b", k)

    def search(k):
        time.sleep(0.002)
        add("s", k)

    def finish(k):
        in_finish.set()
        time.sleep(0.004)
        add("f", k)
        in_finish.clear()
        return -k

    def stage(k):
        overlapped.append(in_finish.is_set())
        time.sleep(0.003)
        add("st", k)

    ks = range(5, 14)
    out = run_steps(ks, search, finish, True, stage, 3, begin if with_begin else None, stager=True)
    assert out == [-k for k in ks]
    pos = {e[:2]: i for i, e in enumerate(log)}
    assert sorted(e[1] for e in log if e[0] == "st") == [k + 3 for k in ks]          # K stagings, all inside the call
    for k in ks:
        assert pos[("s", k)] < pos[("f", k)] < pos[("st", k + 3)]
        if k + 3 in ks:
            first = pos[("b", k + 3)] if with_begin else pos[("s", k + 3)]
            assert pos[("st", k + 3)] < first                                        # staged before it is started
        if k + 1 in ks:
            assert pos[("f", k)] < pos[("f", k + 1)] and pos[("s", k)] < pos[("s", k + 1)]
    assert any(overlapped)                                                           # staging beside a finish


def test_an_error_in_the_staging_thread_reaches_the_caller(run_steps):
    def stage(k):
        if k == 6:
            raise KeyError("stage")

    with pytest.raises(KeyError):
        run_steps(range(8), lambda k: time.sleep(0.001), lambda k: k, True, stage, 3, None, stager=True)
