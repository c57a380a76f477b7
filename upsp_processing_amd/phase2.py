"""Phase 2 of psp_process on the GPU engine (cpp/exec/psp_process.cpp:2260-2625).

Inputs are what phase 1 leaves behind -- this rank's node-major intensity slice
`[nodes_r, F]` (the intensity_transpose block), `intensity_avg`, `coverage` -- plus the
paint calibration, the tunnel conditions and (wind-on) the steady-state Cp per node.
Output: delta-Cp time series (pressure_transpose) and its rms / avg / gain per node.

Nodes are already partitioned across ranks by the phase-1 exchange, so phase 2 has no
data-path collective: the per-node rms / avg / gain vectors of the ranks are disjoint
slices (the reference's MPI_Reduce to rank 0, :2521-2527, adds zeros to them).
"""
import math
import os

import numpy as np
import torch
import torch.distributed as dist

from . import engine


def read_paint_calibration(path):
    """PaintCalibration(filename), cpp/lib/non_cv_upsp.cpp:19-63: lines `name = value`,
    names a..f, everything else ignored; whitespace stripped.  Returns [a,b,c,d,e,f] (f32)."""
    cal = dict.fromkeys("abcdef", 0.0)
    with open(path) as f:
        for line in f:
            toks = "".join(line.split()).split("=")
            if len(toks) == 2 and toks[0] in cal:
                cal[toks[0]] = float(toks[1])
    return [float(np.float32(cal[k])) for k in "abcdef"]


_TCOND_KEYS = {"ALPHA": "alpha", "BETA": "beta", "PHI": "phi", "MACH": "mach", "RNU": "rey",
               "PTOT": "ptot", "Q": "qbar", "TTF": "ttot", "PS": "ps", "TCAVG": "tcavg"}


def read_tunnel_conditions(path):
    """upsp::read_tunnel_conditions, cpp/lib/non_cv_upsp.cpp:107-215: the first line whose
    first token is '#' names the columns, the next line holds the values."""
    cond = {v: float("nan") for v in _TCOND_KEYS.values()}
    with open(path) as f:
        lines = f.read().splitlines()
    for i, line in enumerate(lines):
        terms = line.split()
        if terms and terms[0] == "#":
            vals = lines[i + 1].split() if i + 1 < len(lines) else []
            if len(vals) != len(terms) - 1:
                raise ValueError("failed to parse %r: %d names, %d values" % (path, len(terms) - 1, len(vals)))
            for name, v in zip(terms[1:], vals):
                if name in _TCOND_KEYS:
                    try:
                        cond[_TCOND_KEYS[name]] = float(np.float32(float(v)))
                    except ValueError:
                        pass
            break
    return cond


def model_temperature(tcond, r=0.896, gamma=1.4, F_to_R=459.67):
    """Wall temperature estimate, psp_process.cpp:2287-2310 (float arithmetic like the
    reference); the thermocouple average supersedes it when present."""
    f32 = np.float32
    ttot = f32(f32(tcond["ttot"]) + f32(F_to_R))
    mach = f32(tcond["mach"])
    t_inf = f32(float(ttot) / (1.0 + (float(f32(gamma)) - 1.0) * 0.5 * float(mach) * float(mach)))
    ttot = f32(ttot - f32(F_to_R))
    t_inf = f32(t_inf - f32(F_to_R))
    wall = f32(f32(f32(r) * f32(ttot - t_inf)) + t_inf)
    tc = tcond.get("tcavg", float("nan"))
    return float(wall) if math.isnan(tc) else float(f32(tc))


class Phase2:
    def __init__(self, paint_cal, tcond, degree=6, r=0.896, gamma=1.4):
        self.cal = list(paint_cal)
        self.tcond = dict(tcond)
        self.degree = int(degree)
        self.model_temp = model_temperature(tcond, r=r, gamma=gamma)

    def process(self, series, iref, coverage, steady=None, model_temp=None, in_place=True):
        """series: this rank's [nodes_r, F] slice (device); iref / coverage / steady /
        model_temp: per-node values for the same slice.  steady=None = wind-off (:2354)."""
        out = series if in_place else None
        T = self.model_temp if model_temp is None else model_temp
        return engine.phase2_pressure(series, iref, coverage, self.cal, self.tcond["qbar"],
                                      self.tcond["ps"], steady=steady, model_temp=T,
                                      degree=self.degree, out=out)

    @staticmethod
    def gather_finals(res, shard):
        """Whole-model avg / rms / gain on every rank (each rank computed its node slice)."""
        from . import distributed as D
        return {k: D.gather_node_vector(res[k], shard) for k in ("avg", "rms", "gain")}

    def write_outputs(self, out_dir, res, finals, steady, nnodes, node_start=0, model_temp=None):
        """Flat files of phase 2 (:2548-2612): pressure_transpose (each rank at its byte offset),
        rms, avg, gain, steady_state (Cp > 3 -> NaN), model_temp, vv-cp-rms/avg.dat."""
        from .psp import Phase1
        os.makedirs(out_dir, exist_ok=True)
        rank = dist.get_rank() if dist.is_initialized() else 0
        P = res["pressure_t"]
        from . import distributed as D
        fd = D.create_shared_file(os.path.join(out_dir, "pressure_transpose"), nnodes * P.shape[1] * 4)
        try:
            os.pwrite(fd, P.cpu().numpy().astype("<f4").tobytes(), P.shape[1] * node_start * 4)
        finally:
            os.close(fd)
        if rank == 0:
            for name in ("rms", "avg", "gain"):
                finals[name].cpu().numpy().astype("<f4").tofile(os.path.join(out_dir, name))
            st = np.zeros(nnodes, np.float32) if steady is None else np.array(steady, np.float32)
            st[st > 3.0] = np.nan                                       # :2567-2571
            st.astype("<f4").tofile(os.path.join(out_dir, "steady_state"))
            (np.full(nnodes, self.model_temp, "<f4") if model_temp is None
             else np.asarray(model_temp, "<f4")).tofile(os.path.join(out_dir, "model_temp"))
            Phase1.dump_vv(os.path.join(out_dir, "vv-cp-rms.dat"), finals["rms"].cpu().numpy())
            Phase1.dump_vv(os.path.join(out_dir, "vv-cp-avg.dat"), finals["avg"].cpu().numpy())
