"""CPU: the C-ABI library loads and exports every symbol include/upsp_gpu.h declares
(no compute calls -- there is no GPU in the build container)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "upsp_gpu.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(upsp_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from upsp_processing_amd import _capi
    assert os.path.exists(_capi.LIB_PATH), "run `python -m upsp_processing_amd.build`"
    L = ctypes.CDLL(_capi.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing


def test_ctypes_table_matches_header():
    from upsp_processing_amd import _capi
    assert sorted(_capi.SIGNATURES) == declared_symbols()
    _capi.lib()     # binds every signature; AttributeError if one is missing


def test_host_only_entry_points():
    """Entry points that never touch the device can be exercised on CPU."""
    import numpy as np
    from upsp_processing_amd import engine, _capi
    st, ex = engine.apportion(10, 4)
    assert st == [0, 3, 6, 8] and ex == [3, 3, 2, 2]
    st, ex = engine.apportion(100000, 8)
    assert sum(ex) == 100000 and ex == [12500] * 8
    with pytest.raises(_capi.UpspError):
        engine.apportion(5, 0)
    cam = _capi.make_camera(np.eye(3), np.zeros(5), np.diag([1., -1., -1.]), [0, 0, 4.0], 64, 64)
    c = engine.camera_center(cam)
    assert np.allclose(c, [0, 0, 4.0])
    assert abs(engine.oblique_threshold(70) - np.deg2rad(110)) < 1e-6


def test_project_points_matches_oracle(oracle):
    """cv::projectPoints restatement: host entry point vs oracle, bit for bit."""
    import numpy as np
    from upsp_processing_amd import engine, _capi, synthetic as syn
    c = syn.pinhole_camera(1024, 512, center=(0.3, -0.2, 5.0), half_extent=2.0, k1=-0.09,
                           azimuth_deg=25)
    c["dist"][1:4] = [0.01, 1e-3, -2e-3]
    cam_g = _capi.make_camera(c["K"], c["dist"], c["R"], c["t"], 1024, 512)
    cam_o = oracle.make_camera(c["K"], c["dist"], c["R"], c["t"], 1024, 512)
    pts = np.random.default_rng(3).normal(size=(2000, 3)).astype(np.float32)
    a = engine.project_points(cam_g, pts)
    b = oracle.project_points(cam_o, pts)
    assert np.array_equal(a, b)
    assert np.allclose(engine.camera_center(cam_g), oracle.cam_center(cam_o), rtol=0, atol=0)


def test_header_is_plain_c(tmp_path):
    """The boundary is a C ABI: include/upsp_gpu.h must compile as C99 on its own (what a cgo /
    ctypes / JNI binding sees), and a C translation unit must link against the library."""
    import subprocess
    from upsp_processing_amd import _capi
    hdr = os.path.join(ROOT, "include", "upsp_gpu.h")
    subprocess.run(["gcc", "-x", "c", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror",
                    "-fsyntax-only", hdr], check=True)
    src = tmp_path / "probe.c"
    src.write_text('#include "upsp_gpu.h"\n#include <stdio.h>\n'
                   "int main(void){int s[4],e[4];"
                   "if(upsp_apportion(10,4,s,e)!=UPSP_OK) return 1;"
                   'printf("%d %d %d %d\\n",e[0],e[1],e[2],e[3]);return 0;}\n')
    exe = tmp_path / "probe"
    libdir = os.path.dirname(_capi.LIB_PATH)
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                    "-L", libdir, "-l:" + os.path.basename(_capi.LIB_PATH),
                    "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    assert out.split() == ["3", "3", "2", "2"]


def test_phase_labels_and_exchange_layout_on_the_host():
    """Host-only parts of the round-3 entry points: phase labels (roctx range + the reference's timedBarrierPoint line,
    cpp/exec/psp_process.cpp:585-606) and the misuse paths of the communicator API that need no device."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from upsp_processing_amd import _capi\n"
            "with _capi.phase('phase 1: frame loop', sync=False) as outer:\n"
            "    with _capi.phase('inner', sync=False) as inner:\n"
            "        pass\n"
            "assert inner.seconds >= 0 and outer.seconds >= inner.seconds\n"
            "import ctypes as C\n"
            "s = C.c_double()\n"
            "assert _capi.lib().upsp_phase_end(C.byref(s)) == -1          # end without begin\n"
            "h = C.c_void_p()\n"
            "assert _capi.lib().upsp_comm_create(None, 0, 1, C.byref(h)) == -1\n"
            "assert _capi.lib().upsp_exchange_create(None, 10, 10, 1, C.byref(h)) == -1\n"
            "print('ok')\n" % ROOT)
    env = dict(os.environ, UPSP_PHASE_TIMES="1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stderr
    lines = [l for l in r.stderr.splitlines() if l.startswith("+++ ")]
    assert len(lines) == 2 and lines[0].startswith("+++ inner") and lines[1].startswith("+++ phase 1: frame loop"), r.stderr
