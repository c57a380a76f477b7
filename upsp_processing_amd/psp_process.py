"""`psp_process` command-line surface for phase 1 on the GPU engine.

    python -m upsp_processing_amd.psp_process -input_file=run.inp -h5_out=out.h5 -paint_cal=paint.json \\
           [-frames=N] [-add_out_dir=DIR] [-cutoff_x_max=X]

Accepts the reference's flags (cv::CommandLineParser style `-name=value`,
cpp/exec/psp_process.cpp:1193-1218) and its input deck (`@general / @vars / @all / @camera /
@options / @output` with `$var` substitution, docs/sphinx/file-formats.rst:239-470,
cpp/lib/upsp_inputs.cpp) for the part of the pipeline this repository implements:

* grid: Cart3D `.tri` (cpp/lib/TriModel.ipp:117-257) or PLOT3D surface grid `.p3d/.g/.x/.grid/.grd`
  (cpp/lib/plot3d.cpp, cpp/lib/P3DModel.ipp; zone overlaps within 1e-3, psp_process.cpp:1378)
* camera calibration JSON (cpp/lib/CameraCal.cpp:19-54)
* video: 12-bit Photron `.mraw` (cpp/lib/MrawReader.cpp) or Phantom `.cine` (cpp/lib/CineReader.cpp:
  12-bit packed, 10-bit packed with the look-up table given in UPSP_CINE_LUT, 8-bit mode)
* options: registration = none|pixel, filter = none|gaussian|box (+ filter_size),
  overlap = best_view|average_view, oblique_angle, number_frames
  target_patcher = none|polynomial (targets file in @all or @camera; flags -bound_pts,
  -buffer_pts, -target_diam_sf, cpp/exec/psp_process.cpp:1207-1210)

and writes the phase-1 flat files (intensity_transpose, intensity_avg, intensity_rms,
intensity_ratio_0, coverage, camNN-uv, vv-int-*.dat; cpp/exec/psp_process.cpp:524-540).

Phase 2 (cpp/exec/psp_process.cpp:2260-2625) runs when `-paint_cal` names a readable file and
the deck's @all section has `sds` (tunnel conditions): delta-Cp per node and frame ->
pressure_transpose, rms, avg, gain, steady_state, model_temp, vv-cp-*.dat.  `-steady_p3d` /
`-model_temp_p3d` (PLOT3D function files) are read directly for PLOT3D model grids and interpolated
from `-steady_grid` (upsp::interpolate: 10 nearest nodes, inverse-distance weights) for `.tri` models.  The HDF5 container is out of scope; `-h5_out` is
accepted and ignored.

`-count_rays` (not a reference flag): phase 1 reports the number of rays the REFERENCE casts for the cameras
(every in-frame node's ray + retries, which means casting them all) instead of the rays this build really cast.
"""
import json
import os
import sys

import numpy as np


class DeckError(ValueError):
    pass


def parse_flags(argv):
    """cv::CommandLineParser syntax: -name=value or --name=value; bare -name = true."""
    flags = {}
    for a in argv:
        if not a.startswith("-"):
            raise DeckError("unexpected argument %r" % a)
        a = a.lstrip("-")
        k, _, v = a.partition("=")
        flags[k] = v if v != "" else "true"
    for req in ("input_file",):
        if req not in flags:
            raise DeckError("missing required flag -%s" % req)
    return flags


def parse_input_deck(path):
    """FileInputs::Load (cpp/lib/upsp_inputs.cpp): sections with key = value lines."""
    deck = {"general": {}, "vars": {}, "all": {}, "camera": [], "options": {}, "output": {}}
    section = None
    with open(path) as f:
        for raw in f:
            line = raw.split("#", 1)[0].strip()
            if not line:
                continue
            if line.startswith("@"):
                section = line[1:].strip().lower()
                if section not in deck:
                    raise DeckError("unknown section @%s" % section)
                if section == "camera":
                    deck["camera"].append({})
                continue
            if section is None or "=" not in line:
                raise DeckError("malformed line %r" % raw.rstrip())
            k, v = [t.strip() for t in line.split("=", 1)]
            for name, val in deck["vars"].items():       # $var substitution
                v = v.replace("$" + name, val)
            (deck["camera"][-1] if section == "camera" else deck[section])[k] = v
    opts = deck["options"]
    # defaults: cpp/lib/upsp_inputs.cpp:29-33
    opts.setdefault("target_patcher", "none")
    opts.setdefault("registration", "none")
    opts.setdefault("filter", "none")
    opts.setdefault("filter_size", "1")
    opts.setdefault("overlap", "average_view")
    opts.setdefault("oblique_angle", "70")
    # validation: cpp/exec/psp_process.cpp:1284-1319
    if deck["general"].get("tunnel", "ames_unitary") != "ames_unitary":
        raise DeckError("only tunnel = ames_unitary is supported")
    if opts["registration"] not in ("none", "pixel"):
        raise DeckError("registration must be none or pixel")
    if opts["filter"] not in ("none", "gaussian", "box"):
        raise DeckError("filter must be none, gaussian or box")
    if opts["filter"] != "none" and int(opts["filter_size"]) % 2 == 0:
        raise DeckError("filter_size must be odd")
    if opts["overlap"] not in ("best_view", "average_view"):
        raise DeckError("overlap must be best_view or average_view")
    if opts["target_patcher"] not in ("none", "polynomial"):
        raise DeckError("target_patcher must be none or polynomial")
    if opts["target_patcher"] == "polynomial":
        for c in deck["camera"]:
            c.setdefault("targets", deck["all"].get("targets", ""))   # upsp_inputs.cpp:160-166
            if not c["targets"] or not os.path.isfile(c["targets"]):
                raise DeckError("Targets file %r: not found" % c["targets"])
    if not deck["camera"]:
        raise DeckError("no @camera section")
    return deck


def read_tri_grid(path):
    """Cart3D unformatted `.tri` (cpp/lib/TriModel.ipp:117-257): Fortran records
    {n_node, n_tri} / n_node x (x,y,z) f32 / n_tri x (n1,n2,n3) int32 1-based / [components]."""
    with open(path, "rb") as f:
        buf = f.read()
    off = 0

    def record():
        nonlocal off
        n = int(np.frombuffer(buf, "<i4", 1, off)[0])
        body = buf[off + 4:off + 4 + n]
        tail = int(np.frombuffer(buf, "<i4", 1, off + 4 + n)[0])
        if tail != n:
            raise DeckError("corrupt Fortran record in %s" % path)
        off += 8 + n
        return body

    n_node, n_tri = np.frombuffer(record(), "<i4", 2)
    xyz = np.frombuffer(record(), "<f4").reshape(n_node, 3).copy()
    tris = np.frombuffer(record(), "<i4").reshape(n_tri, 3).astype(np.int32) - 1
    comps = None
    if off < len(buf):
        comps = np.frombuffer(record(), "<i4").copy()
    return xyz, tris, comps


def write_tri_grid(path, xyz, tris, comps=None):
    """Inverse of read_tri_grid (used by tests and synthetic data writers)."""
    def rec(a):
        b = np.ascontiguousarray(a).tobytes()
        n = np.array([len(b)], "<i4").tobytes()
        return n + b + n
    with open(path, "wb") as f:
        f.write(rec(np.array([xyz.shape[0], tris.shape[0]], "<i4")))
        f.write(rec(xyz.astype("<f4")))
        f.write(rec((tris + 1).astype("<i4")))
        if comps is not None:
            f.write(rec(np.asarray(comps, "<i4")))


def read_camera_json(path):
    """read_json_camera_calibration (cpp/lib/CameraCal.cpp:19-54): only the first four
    distortion coefficients are read."""
    j = json.load(open(path))
    dist = np.zeros(5)
    dc = np.asarray(j["distCoeffs"], dtype=np.float64).ravel()
    dist[:min(4, dc.size)] = dc[:4]
    return dict(K=np.asarray(j["cameraMatrix"], dtype=np.float64),
                dist=dist, R=np.asarray(j["rmat"], dtype=np.float64),
                t=np.asarray(j["tvec"], dtype=np.float64).ravel(),
                size=(int(j["imageSize"][0]), int(j["imageSize"][1])))


def spawn_ranks(n, argv):
    """`-ranks=N` (the `mpiexec -n N psp_process ...` of the reference's batch templates): start N rank
    processes, one per GPU, before this process has touched a GPU; exit code = theirs."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), "-m", "upsp_processing_amd.psp_process"]
    return subprocess.call(cmd + [a for a in argv if not a.lstrip("-").startswith("ranks=")])


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    flags = parse_flags(argv)
    nranks = int(flags.get("ranks", "1"))
    if nranks > 1 and "RANK" not in os.environ:
        return spawn_ranks(nranks, argv)
    import torch
    from . import engine, psp, synthetic, video, distributed as D
    D.init_from_env()               # device of this rank + process group (MPI_Init of the reference)
    try:
        return run(flags)
    finally:
        D.shutdown()


def run(flags):
    import torch
    from . import engine, psp, synthetic, video, distributed as D
    deck = parse_input_deck(flags["input_file"])
    opts = deck["options"]
    grid = deck["all"].get("grid")
    ext = (grid or "").rsplit(".", 1)[-1]
    overlap_src = None
    if ext == "tri":                                   # GridType::Tri (upsp_inputs.cpp:438-444)
        xyz, tris, _ = read_tri_grid(grid)
        normals = synthetic.node_normals(xyz, tris)    # TriModel_::calcNormals
        s9, tri_nodes = synthetic.soup(xyz, tris)      # TriModel_::extract_tris
    elif ext in ("p3d", "g", "x", "grid", "grd"):      # GridType::P3D
        from . import grids
        model = grids.P3DModel.from_file(grid, 1e-3)   # psp_process.cpp:1378
        xyz, normals = model.nodes(), model.normals
        s9, tri_nodes = model.extract_tris()
        overlap_src = model.overlap_source()
    else:
        raise DeckError("grid must be a Cart3D .tri or a PLOT3D .p3d/.g/.x/.grid/.grd file (got %r)" % grid)
    datanode = None
    if "cutoff_x_max" in flags:                       # psp_process.cpp:1448-1487
        datanode = (xyz[:, 0] <= float(flags["cutoff_x_max"])).astype(np.uint8)
    cams, readers = [], []
    for c in deck["camera"]:
        cal = read_camera_json(c["calibration"])
        cams.append(cal)
        fn = c.get("filename") or c.get("cine")
        if fn and fn.endswith(".mraw"):
            readers.append(video.MrawReader(fn))
        elif fn and fn.endswith(".cine"):
            try:
                readers.append(video.CineReader(fn))              # 10-bit files: UPSP_CINE_LUT
            except (ValueError, NotImplementedError) as e:
                raise DeckError(str(e))
        else:
            raise DeckError("video must be a .mraw or .cine file (got %r)" % fn)
    size = cams[0]["size"]
    for cal, r in zip(cams, readers):
        if cal["size"] != size or (r.width, r.height) != size:
            raise DeckError("camera calibration / video sizes disagree")
    nframes = min(r.num_frames for r in readers)
    if "number_frames" in opts and int(opts["number_frames"]) > 0:
        nframes = min(nframes, int(opts["number_frames"]))
    if "frames" in flags and int(flags["frames"]) > 0:
        nframes = min(nframes, int(flags["frames"]))

    first = [r.read_frames_device(1, 1)[0] for r in readers]
    patch_kw = {}
    if opts["target_patcher"] == "polynomial":                      # phase 0, :2088-2182
        patch_kw = dict(targets=[c["targets"] for c in deck["camera"]], first_frames=first,
                        bit_depth=12, bound_pts=int(flags.get("bound_pts", 2)),
                        buffer_pts=int(flags.get("buffer_pts", 1)),
                        target_diam_sf=float(flags.get("target_diam_sf", 1.2)))
    # phase labels: roctx ranges + (UPSP_PHASE_TIMES=1) the reference's "+++ label [total elapsed ...]" lines
    # (timedBarrierPoint, cpp/exec/psp_process.cpp:585-606)
    from . import _capi
    phases = []

    def begin(label):
        phases.append(_capi.phase(label))
        phases[-1].__enter__()

    def end():
        phases.pop().__exit__(None, None, None)

    begin("phase 1: BVH + projection matrices (createBVH, create_projection_mat)")
    job = psp.Phase1(s9, tri_nodes, xyz, normals, cams, size,
                     oblique_angle=float(opts["oblique_angle"]), overlap=opts["overlap"],
                     datanode=datanode, registration=opts["registration"] == "pixel",
                     filter=None if opts["filter"] == "none" else opts["filter"],
                     filter_size=int(opts["filter_size"]), overlap_src=overlap_src,
                     count_rays="count_rays" in flags, **patch_kw)
    if job.visible_targets is not None and (not D.dist.is_initialized() or D.dist.get_rank() == 0):
        for c, vis in enumerate(job.visible_targets):
            print("camera %d: %d visible targets patched" % (c + 1, len(vis)))
    end()
    shard = D.Shard(nframes, job.nnodes)
    f0, nf = shard.my_frames
    job.set_first_frames(first)
    rows_t = torch.empty((job.nnodes, engine.series_ld(max(nf, 1))), dtype=torch.float32,
                         device="cuda")[:, :max(nf, 1)]
    job.pipe.set_row_padding(True)          # (columns nf .. series_ld(nf) of that allocation are padding)
    chunk = 256
    # frames go disk -> pinned ring -> device on a copy stream of their own (the reference's read-ahead
    # thread, psp_process.cpp:867-1007): the upload of chunk k + 1 overlaps the processing of chunk k
    feeds = [video.FrameFeed(min(chunk, max(nf, 1)) * r.frame_bytes, 3) if getattr(r, "raw_bit_depth", 12) == 12
             else None for r in readers]
    begin("phase 1: frame loop (read, register, patch, project, accumulate)")
    series = None
    try:
        if job.pixel_wire(shard):
            # N > 1, one camera, no float stage: the active pixels' series travel chunk by chunk while the next chunk is
            # read, the owner of a node writes its rows (psp.Phase1.frame_loop_pixel_wire)
            series = job.frame_loop_pixel_wire(
                shard, lambda c0, n: [r.read_frames_device(f0 + c0 + 1, n, feed=fd) for r, fd in zip(readers, feeds)], chunk,
                progress=lambda c0: print("  Rank 0:: processing frame %d" % (f0 + c0)) if shard.rank == 0 and c0 % (chunk * 4) == 0 else None)
        for c0 in range(0, nf if series is None else 0, chunk):
            n = min(chunk, nf - c0)
            batch = [r.read_frames_device(f0 + c0 + 1, n, feed=fd) for r, fd in zip(readers, feeds)]     # 1-based frames
            job.process(batch, first_frame=f0 + c0, rows_t=rows_t, col0=c0)
            if shard.rank == 0 and c0 % (chunk * 4) == 0:
                print("  Rank 0:: processing frame %d" % (f0 + c0))
    finally:
        # pinned slots + device twins (3 x 256 frames per camera) are not left to the garbage collector
        torch.cuda.synchronize()
        for fd in feeds:
            if fd is not None:
                fd.close()
    end()
    begin("phase 1: reductions + time-series exchange (MPI_Reduce, global_transpose)")
    finals = job.finalize(nframes)
    if series is None:
        series = D.exchange_time_series(rows_t[:, :nf], shard)
    end()
    begin("phase 1: output files")
    out_dir = flags.get("add_out_dir") or deck["output"].get("dir") or "."
    job.write_outputs(out_dir, finals, series, node_start=shard.my_nodes[0])
    if shard.rank == 0:
        for name, col in (("X", 0), ("Y", 1), ("Z", 2)):                    # :524-540
            xyz[:, col].astype("<f4").tofile(os.path.join(out_dir, name))
        print("phase 1 complete: %d frames, %d nodes, %d %s" % (nframes, job.nnodes, job.nrays, job.nrays_kind))
    end()

    # ---- phase 2 (psp_process.cpp:2260-2625) ----
    paint_cal, sds = flags.get("paint_cal"), deck["all"].get("sds")
    if paint_cal and os.path.isfile(paint_cal) and sds:
        from . import phase2
        begin("phase 2: intensity -> pressure")
        steady = temp = None
        if flags.get("steady_p3d") or flags.get("model_temp_p3d"):
            # structured models read one scalar per grid point (psp_process.cpp:2327-2335, 2360-2368);
            # unstructured ones interpolate it from the structured steady grid (-steady_grid) with
            # upsp::interpolate, k = 10, p = 2 (:2336-2344, 2369-2377)
            from . import grids
            sgrid = None
            if overlap_src is None:
                if not flags.get("steady_grid"):
                    raise DeckError("-steady_p3d / -model_temp_p3d on a .tri model need -steady_grid")
                try:
                    sgrid = grids.P3DModel.from_file(flags["steady_grid"], 1e-3)
                except (OSError, ValueError) as e:
                    raise DeckError(str(e))
            for key in ("steady_p3d", "model_temp_p3d"):
                if flags.get(key):
                    try:
                        vals = grids.read_plot3d_scalar_function_file(flags[key])
                    except (OSError, ValueError) as e:
                        raise DeckError(str(e))
                    if sgrid is not None:
                        if vals.size != sgrid.size():
                            raise DeckError("%s inconsistent with the steady grid (expect %d values, got %d)"
                                            % (key, sgrid.size(), vals.size))
                        vals = engine.interpolate_idw(sgrid.nodes(), vals, xyz, 10, 2.0).cpu().numpy()
                    if vals.size != job.nnodes:
                        raise DeckError("%s inconsistent with grid (expect %d values, got %d)"
                                        % (key, job.nnodes, vals.size))
                    if key == "steady_p3d":
                        steady = vals
                    else:
                        temp = vals
        p2 = phase2.Phase2(phase2.read_paint_calibration(paint_cal), phase2.read_tunnel_conditions(sds))
        n0, nn = shard.my_nodes
        res = p2.process(series, finals["avg"][n0:n0 + nn].contiguous(),
                         finals["coverage"][n0:n0 + nn].contiguous(),
                         steady=None if steady is None else steady[n0:n0 + nn],
                         model_temp=None if temp is None else temp[n0:n0 + nn], in_place=True)
        p2.write_outputs(out_dir, res, p2.gather_finals(res, shard), steady, job.nnodes, node_start=n0,
                         model_temp=temp)
        if shard.rank == 0:
            print("phase 2 complete: model temperature %.1f F, qbar %.2f" % (p2.model_temp, p2.tcond["qbar"]))
        end()
    job.close()
    return 0


if __name__ == "__main__":
    try:
        sys.exit(main())
    except DeckError as e:
        print("psp_process: %s" % e, file=sys.stderr)
        sys.exit(1)
