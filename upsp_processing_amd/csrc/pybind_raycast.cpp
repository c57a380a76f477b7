// pybind11 module `raycast`: drop-in for the reference's `upsp.raycast`
// (cpp/pybind11/raycast.cpp:8-36) on top of the C ABI of libupsp_gpu.so.
//
// Same four names and semantics -- CreateBVH(raw, stride) -> BVH,
// Ray(x0,y0,z0,xr,yr,zr), Hit() with read-only .pos, BVH.intersect(ray, hit) ->
// bool (mutates hit) -- and the class is named "BVH" (checked by
// python/upsp/cam_cal_utils/external_calibrate.py:1453).  Added, non-breaking:
// batched queries on numpy arrays, extra read-only Hit fields, exceptions instead
// of exit() (pspRT.cpp:362-365 DIEs on an empty BVH).
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <cfloat>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <vector>

#include "upsp_gpu.h"

namespace py = pybind11;

namespace {

struct PyRay {
    float o[3], d[3];
};

struct PyHit {
    float pos[3] = {0, 0, 0}, nrm[3] = {0, 0, 0};
    float t = FLT_MAX, u = 0, v = 0, w = 0;  // rt::Hit::Hit(), pspRT.cpp:21-22
    int primID = -1;
};

void check(int rc)
{
    if (rc != UPSP_OK) throw std::runtime_error(std::string("upsp_gpu: ") + upsp_last_error());
}

struct PyBVH {
    upsp_bvh *h = nullptr;
    explicit PyBVH(const std::vector<float> &raw, size_t stride)
    {
        if (stride != 3) throw std::invalid_argument("stride must be 3 (x,y,z per vertex)");
        if (raw.size() % 9 != 0) throw std::invalid_argument("raw must hold 9 floats per triangle");
        check(upsp_bvh_create(raw.data(), raw.size() / 9, &h));
    }
    PyBVH(const float *raw, size_t ntris) { check(upsp_bvh_create(raw, ntris, &h)); }
    ~PyBVH() { upsp_bvh_destroy(h); }
    PyBVH(const PyBVH &) = delete;
    PyBVH &operator=(const PyBVH &) = delete;

    // rt::BVH::intersect(const Ray&, Hit*) (pspRT.cpp:359-431): closest hit kept in
    // *hit only if nearer than hit->t (strict <), return value = any hit.
    bool intersect(const PyRay &r, PyHit &hit) const
    {
        uint8_t any = 0;
        float t, uvw[3], pos[3], nrm[3];
        int32_t prim;
        upsp_hits out;
        out.hit = &any; out.t = &t; out.prim = &prim; out.uvw = uvw; out.pos = pos; out.nrm = nrm;
        check(upsp_bvh_intersect_host(h, r.o, 3, r.d, 1, &out));
        if (any && prim >= 0 && t < hit.t) {
            hit.t = t; hit.u = uvw[0]; hit.v = uvw[1]; hit.w = uvw[2];
            std::memcpy(hit.pos, pos, sizeof(pos));
            std::memcpy(hit.nrm, nrm, sizeof(nrm));
            hit.primID = prim;
        }
        return any != 0;
    }
};

using farray = py::array_t<float, py::array::c_style | py::array::forcecast>;

size_t rays_of(const farray &org, const farray &dir, int &stride)
{
    if (dir.ndim() != 2 || dir.shape(1) != 3) throw std::invalid_argument("dirs must be (N,3)");
    const size_t n = (size_t)dir.shape(0);
    if (org.size() == 3) {
        stride = 0;
    } else {
        if (org.ndim() != 2 || org.shape(1) != 3 || (size_t)org.shape(0) != n)
            throw std::invalid_argument("origins must be (3,) or (N,3)");
        stride = 3;
    }
    return n;
}

}  // namespace

PYBIND11_MODULE(raycast, m)
{
    m.doc() = "Ray-tracing capabilities using a Bounding Volume Hierarchy (BVH) on MI355X";

    m.def(
        "CreateBVH",
        [](py::object raw, size_t stride) {
            if (py::isinstance<py::array>(raw)) {
                farray a = farray::ensure(raw);
                if (!a) throw std::invalid_argument("raw must be convertible to float32");
                if (stride != 3) throw std::invalid_argument("stride must be 3");
                if (a.size() % 9 != 0) throw std::invalid_argument("raw must hold 9 floats per triangle");
                return std::make_unique<PyBVH>(a.data(), (size_t)a.size() / 9);
            }
            return std::make_unique<PyBVH>(raw.cast<std::vector<float>>(), stride);
        },
        py::arg("raw"), py::arg("stride"), "Create a bounding volume hierarchy");

    py::class_<PyRay>(m, "Ray")
        .def(py::init([](const float x0, const float y0, const float z0, const float xr,
                         const float yr, const float zr) {
            auto r = std::make_unique<PyRay>();
            r->o[0] = x0; r->o[1] = y0; r->o[2] = z0;
            r->d[0] = xr; r->d[1] = yr; r->d[2] = zr;
            return r;
        }));

    py::class_<PyHit>(m, "Hit")
        .def(py::init<>())
        .def_property_readonly("pos", [](const PyHit &h) {
            return std::vector<float>{h.pos[0], h.pos[1], h.pos[2]};
        })
        .def_property_readonly("nrm", [](const PyHit &h) {
            return std::vector<float>{h.nrm[0], h.nrm[1], h.nrm[2]};
        })
        .def_readonly("t", &PyHit::t)
        .def_readonly("u", &PyHit::u)
        .def_readonly("v", &PyHit::v)
        .def_readonly("w", &PyHit::w)
        .def_readonly("primID", &PyHit::primID);

    py::class_<PyBVH>(m, "BVH")
        .def("intersect", &PyBVH::intersect)
        .def("intersect_many",
             [](const PyBVH &b, farray org, farray dir) {
                 int stride = 3;
                 const size_t n = rays_of(org, dir, stride);
                 py::array_t<uint8_t> hit(n);
                 py::array_t<float> t(n), uvw({n, (size_t)3}), pos({n, (size_t)3}),
                     nrm({n, (size_t)3});
                 py::array_t<int32_t> prim(n);
                 upsp_hits out;
                 out.hit = hit.mutable_data(); out.t = t.mutable_data();
                 out.prim = prim.mutable_data(); out.uvw = uvw.mutable_data();
                 out.pos = pos.mutable_data(); out.nrm = nrm.mutable_data();
                 {
                     py::gil_scoped_release nogil;
                     check(upsp_bvh_intersect_host(b.h, org.data(), stride, dir.data(), n, &out));
                 }
                 py::dict d;
                 d["hit"] = hit.attr("astype")("bool");
                 d["t"] = t; d["prim"] = prim; d["uvw"] = uvw; d["pos"] = pos; d["nrm"] = nrm;
                 return d;
             },
             py::arg("origins"), py::arg("dirs"),
             "Closest hit of N rays: dict(hit,t,prim,uvw,pos,nrm)")
        .def("occluded_many",
             [](const PyBVH &b, farray org, farray dir) {
                 int stride = 3;
                 const size_t n = rays_of(org, dir, stride);
                 py::array_t<uint8_t> hit(n);
                 {
                     py::gil_scoped_release nogil;
                     check(upsp_bvh_occluded_host(b.h, org.data(), stride, dir.data(), n,
                                                  hit.mutable_data()));
                 }
                 return py::object(hit.attr("astype")("bool"));
             },
             py::arg("origins"), py::arg("dirs"),
             "Return value of intersect() for N rays (any hit with t >= 0)")
        .def_property_readonly("handle", [](const PyBVH &b) { return (uintptr_t)b.h; },
                               "address of the upsp_bvh (for the C ABI via ctypes)")
        .def_property_readonly("info", [](const PyBVH &b) {
            upsp_bvh_info i;
            check(upsp_bvh_get_info(b.h, &i));
            py::dict d;
            d["ntris"] = i.ntris; d["n_ref_nodes"] = i.n_ref_nodes; d["n_gpu_nodes"] = i.n_gpu_nodes;
            d["depth"] = i.depth; d["max_leaf"] = i.max_leaf; d["device_bytes"] = i.device_bytes;
            d["build_seconds"] = i.build_seconds;
            d["bounds_min"] = std::vector<float>(i.bounds_min, i.bounds_min + 3);
            d["bounds_max"] = std::vector<float>(i.bounds_max, i.bounds_max + 3);
            return d;
        });
}
