"""CPU unit tests of cuda-l2_amd/defense.py (f3): every check must pass a legitimate op and catch
the attack it is named after.  Timing is injected (wall clock, fake "device sync")."""
import sys
import threading
import time
from pathlib import Path

import pytest
import torch

PKG = Path(__file__).resolve().parent.parent / "cuda-l2_amd"
if str(PKG) not in sys.path:
    sys.path.insert(0, str(PKG))

import defense  # noqa: E402


def legit(a, b, b_col_major, c):
    c.copy_(torch.matmul(a.float(), b.float()).half())


@pytest.fixture
def operands():
    torch.manual_seed(0)
    a = torch.randn((48, 96)).half()
    b = torch.randn((96, 64)).half()
    return a, b, b.t().contiguous(), torch.zeros((48, 64), dtype=torch.half)


def no_sync():
    pass


class FakeClock:
    """Deterministic time for the timing-based checks: it only moves when the op under test (or the fake device
    drain) says so, so host scheduling noise cannot push an honest op over the 1.5x ratio (the wall-clock version of
    this test failed 1 run in 5 on a busy host: VERDICT r1)."""
    now_ms = 0.0


class FakeTimer:
    def begin(self):
        self.t0 = FakeClock.now_ms

    def end(self):
        self.t1 = FakeClock.now_ms

    def elapsed_ms(self) -> float:
        return self.t1 - self.t0


def test_all_checks_pass_for_a_legitimate_op(operands):
    a, b, bt, c = operands

    def timed_legit(a, b, b_col_major, c):      # a synchronous op: its whole cost is paid inside the call
        legit(a, b, b_col_major, c)
        FakeClock.now_ms += 0.25

    ok, results = defense.run_all_defenses(timed_legit, a, b, bt, c, timer_factory=FakeTimer, sync=no_sync)
    assert ok, results
    assert len(results) == 5


def test_thread_injection_is_caught(operands):
    a, b, bt, c = operands
    stop = threading.Event()

    def attack(a, b, b_col_major, c):
        threading.Thread(target=stop.wait, daemon=True).start()

    try:
        passed, msg = defense.check_thread_injection(lambda: attack(a, b, bt, c))
        assert not passed and "thread" in msg
        passed, _ = defense.check_thread_injection(lambda: legit(a, b, bt, c))
        assert passed
    finally:
        stop.set()


def test_stream_injection_is_caught_by_the_guarded_timing(operands):
    a, b, bt, c = operands
    pending = []

    def attack():            # returns at once; the work is only paid for when the "device" is drained
        pending.append(1)
        FakeClock.now_ms += 0.01

    def device_sync():
        while pending:
            pending.pop()
            FakeClock.now_ms += 4.0

    passed, msg, trusted = defense.check_stream_injection(attack, timer_factory=FakeTimer, sync=device_sync, iterations=4)
    assert not passed and "Stream injection detected" in msg and trusted >= 3.0

    def honest():
        time.sleep(0.002)

    passed, msg, _ = defense.check_stream_injection(honest, timer_factory=defense.WallTimer, sync=no_sync, iterations=4)
    assert passed, msg


def test_lazy_evaluation_variants_are_caught(operands):
    a, b, bt, c = operands
    passed, msg = defense.check_lazy_evaluation(lambda a, b, bt, c: None, a, b, bt, c, sync=no_sync)
    assert not passed and "sentinel" in msg
    passed, msg = defense.check_lazy_evaluation(lambda a, b, bt, c: torch.matmul(a.float(), b.float()).half(), a, b, bt, c, sync=no_sync)
    assert not passed and "in place" in msg

    class Lazy(torch.Tensor):
        pass

    lazy_c = torch.zeros((48, 64), dtype=torch.half).as_subclass(Lazy)
    passed, msg = defense.check_lazy_evaluation(legit, a, b, bt, lazy_c, sync=no_sync)
    assert not passed and "Lazy" in msg

    def half_written(a, b, bt, c):
        c[:24].copy_(torch.matmul(a[:24].float(), b.float()).half())

    passed, msg = defense.check_lazy_evaluation(half_written, a, b, bt, c, sync=no_sync)
    assert not passed and "never written" in msg
    passed, _ = defense.check_lazy_evaluation(legit, a, b, bt, c, sync=no_sync)
    assert passed


def test_precision_downgrade_is_caught(operands):
    a, b, bt, c = operands

    def bf16_inside(a, b, bt, c):
        c.copy_(torch.matmul(a.bfloat16().float(), b.bfloat16().float()).bfloat16().half())

    passed, msg = defense.check_precision_downgrade(bf16_inside, a, b, bt, c, sync=no_sync)
    assert not passed and "Precision downgrade" in msg
    c32 = torch.zeros((48, 64), dtype=torch.float32)
    passed, msg = defense.check_precision_downgrade(lambda a, b, bt, c: c.copy_(a.float() @ b.float()), a, b, bt, c32, sync=no_sync)
    assert not passed and "float32" in msg
    passed, _ = defense.check_precision_downgrade(legit, a, b, bt, c, sync=no_sync)
    assert passed


def test_monkey_patching_is_caught(monkeypatch):
    assert defense.check_elapsed_time_monkey_patching()[0]
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    passed, msg = defense.check_elapsed_time_monkey_patching()
    assert not passed and "torch.cuda.synchronize" in msg
    monkeypatch.undo()
    monkeypatch.setattr(torch.cuda.Event, "elapsed_time", lambda self, other: 0.0)
    passed, msg = defense.check_elapsed_time_monkey_patching()
    assert not passed and "elapsed_time" in msg
