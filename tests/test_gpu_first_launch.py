"""First launches in fresh processes: the race-finding runs of round 3 (tools/stress_modes.py, the 24-process reproducer of the
general kernels' exchange race) as a test. Six child processes, one after the other -- each is the first launch of every
kernel form on a cold device context -- must all come back without a differing word, without an exact recomputation and
without RS_ERR_INEXACT (tests/first_launch_stress.py). The race this would have caught passed the ordinary suite nine runs
in ten: wavefronts of a warm process stay close together."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
EXPECTED_FORMS = {"fft/coop8", "fft/coop8_listed", "fft/coop2", "fft/duo", "fft/workgroup", "exact/per_wave", "exact/coop2", "exact/coop4",
                  "split/split_coop", "split/split_duo", "split/split_workgroup"}


def test_six_fresh_processes_differential_stress():
    seen = set()
    for k in range(6):
        r = subprocess.run([sys.executable, os.path.join(HERE, "first_launch_stress.py"), str(300 + k)],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert lines, "child %d printed no summary (rc %d): %s" % (k, r.returncode, r.stderr[-2000:])
        rep = json.loads(lines[-1])
        assert r.returncode == 0 and not rep["findings"], rep
        seen |= set(rep["forms_launched"])
    assert EXPECTED_FORMS <= seen, sorted(EXPECTED_FORMS - seen)      # every kernel form had its first launches
