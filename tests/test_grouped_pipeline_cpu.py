"""CPU: the bookkeeping of kd6d.graph.GroupedTeacherKDStep (which batch sits in which block, which teacher segment and
which student step a call issues, what flush() drains) with the device work replaced by a ledger.  The properties the GPU
parity tests rely on, for every group size and every point at which feeding stops:
  * every batch fed is trained on exactly once, in the order it was fed;
  * when a batch is trained on, the teacher has run ALL its segments over the block that held it, after the batch (and
    its whole period) had been loaded -- never over a half-loaded block;
  * in steady state every call issues exactly one teacher segment and one student step (the pacing the step time
    depends on), and a period's last call rotates the blocks;
  * flush() drains one batch per call, returns None when the pipeline is empty, and a later call starts a new pipeline."""
import itertools

import pytest

from kd6d.graph import GroupedTeacherKDStep


class Ledger(GroupedTeacherKDStep):
    def __init__(self, group):             # no device objects: only the counters of the real __init__
        self.group = group
        self.n_loaded = self.p_valid = self.c_valid = self.c_pos = self.t_pos = 0
        self.draining = False
        self.teacher_passes = 0
        self.load = [None] * group          # batch ids per slot
        self.pas = [None] * group
        self.cur = [None] * group
        self.pas_segments = set()           # teacher segments run over the pass block since it was filled
        self.cur_segments = set()           # ... that the current block's cells come from
        self.trained, self.events = [], []

    def _prepare(self, images, tgt):
        return tgt

    def _load_device(self, images, tgt, s):
        self.load[s] = images
        self.events.append(("load", s))

    def _rotate_device(self, pass_valid, n_loaded):
        if 0 < n_loaded < self.group:
            for s in range(n_loaded, self.group):
                self.load[s] = ("filler", self.load[0])
        if pass_valid:
            self.cur, self.cur_segments = list(self.pas), set(self.pas_segments)
        if n_loaded:
            self.pas, self.pas_segments = list(self.load), set()
        self.events.append(("rotate",))

    def _replay_teacher_segment(self, i):
        assert i == len(self.pas_segments), "segments run in order, each once per block"
        self.pas_segments.add(i)
        self.events.append(("teacher", i))

    def _ensure_captured(self):
        pass

    def _replay_student(self, s):
        assert self.cur_segments == set(range(self.group)), "student step on a block the teacher has not finished"
        b = self.cur[s]
        assert b is not None and not (isinstance(b, tuple) and b[0] == "filler")
        self.trained.append(b)
        self.events.append(("student", s))
        return {"batch": b}


@pytest.mark.parametrize("group", [2, 3, 4])
def test_every_batch_trained_once_in_order(group):
    for n in range(0, 4 * group + 2):
        gs = Ledger(group)
        outs = [gs(i, None) for i in range(n)]
        assert all((o is None) == (i < 2 * group) for i, o in enumerate(outs))
        assert [o["batch"] for o in outs if o is not None] == list(range(max(0, n - 2 * group)))
        assert gs.pending_steps == n - len(gs.trained) == min(n, 2 * group)
        drained = []
        while True:
            o = gs.flush()
            if o is None:
                break
            drained.append(o["batch"])
        assert gs.trained == list(range(n)), (group, n, gs.trained)
        assert gs.pending_steps == 0 and not gs.pending and gs.flush() is None
        # a later call starts a new pipeline
        assert gs("again", None) is None and gs.pending_steps == 1


@pytest.mark.parametrize("group", [2, 3, 5])
def test_steady_state_issues_one_segment_and_one_step_per_call(group):
    gs = Ledger(group)
    for i in range(2 * group):
        gs(i, None)
    gs.events.clear()
    for k in range(3 * group):
        mark = len(gs.events)
        gs(2 * group + k, None)
        ev = gs.events[mark:]
        s = k % group
        want = [("load", s), ("teacher", s), ("student", s)] + ([("rotate",)] if s == group - 1 else [])
        assert ev == want, (k, ev)
    assert gs.teacher_passes == 3 + 1          # one during the filling periods, one per steady period


def test_feeding_a_partly_drained_pipeline_is_refused():
    gs = Ledger(2)
    for i in range(6):
        gs(i, None)
    assert gs.flush() is not None and gs.pending_steps == 3
    with pytest.raises(RuntimeError, match="flush"):
        gs(99, None)
    while gs.flush() is not None:
        pass
    assert gs.trained == list(range(6))


def test_group_of_one_is_the_plain_pipeline():
    with pytest.raises(ValueError):
        GroupedTeacherKDStep(None, None, None, group=1)
