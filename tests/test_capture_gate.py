"""CPU: the gate that lets a hipGraph recording run alone among the host threads of fit_recon(pipeline="chains")
(chore_amd/recon/recon_fit_behave.py _CaptureGate): while one participant is inside exclusive(), the others are parked at a
checkpoint, waiting to enter, or gone; two participants that want to record at once take turns; nobody is left waiting."""
import threading
import time

from chore_amd.recon.recon_fit_behave import _CaptureGate


def test_a_recording_runs_alone_and_everybody_resumes():
    gate = _CaptureGate()
    inside = []                     # what the recorders saw: the progress counters of the others, before and after their recording
    progress = [0, 0]
    stop = threading.Event()
    errors = []

    def worker(k, record_at):
        try:
            gate.enter()
            try:
                n = 0
                while not stop.is_set() and n < 400:
                    gate.checkpoint()
                    progress[k] += 1
                    n += 1
                    if n in record_at:
                        with gate.exclusive():
                            before = list(progress)
                            time.sleep(0.02)        # the "recording": the others must not move meanwhile
                            inside.append((k, before, list(progress)))
                    time.sleep(0.0005)
            finally:
                gate.leave()
        except Exception as e:      # pragma: no cover
            errors.append(e)

    def visitor():                  # the caller's thread around `finish`: enters and leaves, never parks
        try:
            for _ in range(40):
                gate.enter()
                time.sleep(0.001)
                gate.leave()
                time.sleep(0.002)
        except Exception as e:      # pragma: no cover
            errors.append(e)

    ts = [threading.Thread(target=worker, args=(0, {50, 51, 200})), threading.Thread(target=worker, args=(1, {50, 120})),
          threading.Thread(target=visitor)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=60)
    stop.set()
    assert not any(t.is_alive() for t in ts), "a participant is stuck at the gate"
    assert not errors
    assert len(inside) == 5
    for k, before, after in inside:
        assert before[1 - k] == after[1 - k], ("the other participant moved during a recording", k, before, after)
    assert progress == [400, 400]
    assert gate.active == 0 and gate.parked == 0 and not gate.recording


def test_pipelined_pattern_a_recording_waits_for_the_preparing_worker():
    """fit_recon(pipeline=True) (_fit_pipelined): the worker thread holds the gate while it prepares a batch and never parks;
    the calling thread stays entered for the whole loop and records through exclusive() -- which must wait until the worker has
    left and keep its next preparation out until the recording is over"""
    gate = _CaptureGate()
    preparing = threading.Event()
    overlap = []
    errors = []

    def prepare():
        try:
            gate.enter()
            try:
                preparing.set()
                time.sleep(0.01)
                preparing.clear()
            finally:
                gate.leave()
        except Exception as e:      # pragma: no cover
            errors.append(e)

    gate.enter()
    for _ in range(20):
        t = threading.Thread(target=prepare)
        t.start()
        time.sleep(0.002)               # the worker is in the middle of its preparation
        with gate.exclusive():
            overlap.append(preparing.is_set())
            t2 = threading.Thread(target=prepare)       # the next preparation is submitted during the recording
            t2.start()
            time.sleep(0.005)
            overlap.append(preparing.is_set())
        t.join(timeout=10)
        t2.join(timeout=10)
        assert not t.is_alive() and not t2.is_alive()
    gate.leave()
    assert not errors and not any(overlap)
    assert gate.active == 0 and gate.parked == 0 and not gate.recording
