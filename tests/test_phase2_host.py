"""Host-side pieces of phase 2: paint calibration / tunnel conditions parsers and the
model-temperature estimate (cpp/lib/non_cv_upsp.cpp:19-63,107-215; psp_process.cpp:2287-2310)."""
import os

import numpy as np

from upsp_processing_amd import phase2

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_paint_calibration(tmp_path):
    p = tmp_path / "paint.cal"
    p.write_text("a = 1.25\n b=-0.5\nc =  2e-3\n# comment\nd=0.125\ne = 4\nf=-1e-4\nz = 9\nnot a pair\n")
    assert phase2.read_paint_calibration(str(p)) == [float(np.float32(v)) for v in (1.25, -0.5, 2e-3, 0.125, 4, -1e-4)]


def test_tunnel_conditions_reference_file():
    # the reference's own sample (test/data/wtd_test.wtd), committed as data under tests/golden
    c = phase2.read_tunnel_conditions(os.path.join(GOLD, "wtd_test.wtd"))
    assert c["mach"] == 1.0 and abs(c["alpha"] - 0.05) < 1e-7 and abs(c["beta"] - 0.12) < 1e-7
    assert abs(c["phi"] - 0.9) < 1e-7 and c["qbar"] == 0.0 and c["ps"] == 0.0 and c["ttot"] == 0.0
    assert np.isnan(c["tcavg"])                 # no TCAVG column in that file


def test_model_temperature():
    t = dict(mach=0.8, ttot=100.0, tcavg=float("nan"))
    T0 = 100.0 + 459.67
    tinf = T0 / (1 + 0.2 * 0.64) - 459.67
    want = 0.896 * (100.0 - tinf) + tinf
    assert abs(phase2.model_temperature(t) - want) < 1e-3
    t["tcavg"] = 71.5
    assert phase2.model_temperature(t) == 71.5
