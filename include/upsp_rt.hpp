// upsp_rt.hpp -- header-only C++ shim with the reference's `rt::` names on top of the C ABI
// (include/upsp_gpu.h), for source compatibility with the call sites of
// cpp/include/utils/pspRT.h:26-148 in psp_process (cpp/exec/psp_process.cpp:44-53, 85-92,
// 257-267, 281-294):
//
//     std::vector<std::shared_ptr<rt::Primitive>> prims = rt::CreateTriangleMesh(tris, 3);
//     auto scene = std::make_shared<rt::BVH>(prims, 4);
//     rt::Ray ray(orig, dir);  rt::Hit hitrec;
//     bool hit = scene->intersect(ray, &hitrec);   // hitrec.t, .pos, .primID
//
// Vector arguments are any type indexable with [0..2] (Imath::V3f, std::array, float[3]).
// One ray per call goes through the small-batch path of the library (~40 us); hot loops should
// call rt::BVH::intersect_many / the C ABI batch entry points instead.
#ifndef UPSP_RT_HPP
#define UPSP_RT_HPP

#include <cfloat>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "upsp_gpu.h"

namespace rt {

struct V3f {
    float x = 0, y = 0, z = 0;
    V3f() = default;
    V3f(float a, float b, float c) : x(a), y(b), z(c) {}
    float &operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};

struct Hit {  // rt::Hit, pspRT.h:26-36 / pspRT.cpp:21-22
    V3f pos, nrm;
    float t = FLT_MAX, u = 0, v = 0, w = 0;
    int geomID = -1, primID = -1;
};

struct Ray {  // rt::Ray(o, d), pspRT.h:37-50
    V3f o, d;
    Ray() = default;
    template <class A, class B>
    Ray(const A &orig, const B &dir) : o(orig[0], orig[1], orig[2]), d(dir[0], dir[1], dir[2]) {}
};

// rt::Primitive / rt::CreateTriangleMesh: the soup itself is the primitive list
struct Primitive {
    std::shared_ptr<const std::vector<float>> soup;  // 9 floats per triangle
};

inline std::vector<std::shared_ptr<Primitive>> CreateTriangleMesh(const std::vector<float> &raw,
                                                                  size_t stride)
{
    if (stride != 3 || raw.size() % 9 != 0)
        throw std::invalid_argument("CreateTriangleMesh: 9 floats per triangle, stride 3");
    auto p = std::make_shared<Primitive>();
    p->soup = std::make_shared<const std::vector<float>>(raw);
    return {p};
}

struct BVH {
    upsp_bvh *h = nullptr;
    BVH(const std::vector<std::shared_ptr<Primitive>> &p, int /*maxPrimsInNode*/ = 4)
    {
        if (p.empty() || !p[0] || !p[0]->soup || p[0]->soup->empty()) return;  // "no primitives!"
        if (upsp_bvh_create(p[0]->soup->data(), p[0]->soup->size() / 9, &h) != UPSP_OK)
            throw std::runtime_error(upsp_last_error());
    }
    ~BVH() { upsp_bvh_destroy(h); }
    BVH(const BVH &) = delete;
    BVH &operator=(const BVH &) = delete;

    // rt::BVH::intersect, pspRT.cpp:359-431
    bool intersect(const Ray &ray, Hit *hit) const
    {
        if (!h) throw std::runtime_error("rt::BVH::intersect(): no nodes!");
        uint8_t any = 0;
        float t, uvw[3], pos[3], nrm[3];
        int32_t prim;
        upsp_hits out{&any, &t, &prim, uvw, pos, nrm};
        const float o[3] = {ray.o.x, ray.o.y, ray.o.z}, d[3] = {ray.d.x, ray.d.y, ray.d.z};
        if (upsp_bvh_intersect_host(h, o, 3, d, 1, &out) != UPSP_OK)
            throw std::runtime_error(upsp_last_error());
        if (any && prim >= 0 && t < hit->t) {
            hit->t = t; hit->u = uvw[0]; hit->v = uvw[1]; hit->w = uvw[2];
            hit->pos = V3f(pos[0], pos[1], pos[2]);
            hit->nrm = V3f(nrm[0], nrm[1], nrm[2]);
            hit->primID = prim;
        }
        return any != 0;
    }

    // batched form: origins / dirs are 3*n floats, outputs may be null
    void intersect_many(const float *org, int org_stride, const float *dir, size_t n, uint8_t *hit,
                        float *t, int32_t *prim, float *pos) const
    {
        upsp_hits out{hit, t, prim, nullptr, pos, nullptr};
        if (upsp_bvh_intersect_host(h, org, org_stride, dir, n, &out) != UPSP_OK)
            throw std::runtime_error(upsp_last_error());
    }
};

inline std::unique_ptr<BVH> CreateBVH(const std::vector<float> &raw, size_t stride)
{
    return std::make_unique<BVH>(CreateTriangleMesh(raw, stride), 4);
}

}  // namespace rt
#endif
