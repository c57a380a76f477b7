// C++ call sites of the reference (psp_process.cpp:44-53, 257-267) compiled against the
// rt:: shim; prints "hit t x y z primID" lines that tests/test_rt_shim_gpu.py checks.
#include <array>
#include <cstdio>
#include "upsp_rt.hpp"

int main()
{
    std::vector<float> tris = {0, 0, 1, 0, 1, 0, 1, 0, 0,  0, 0, 1, 0, 1, 0, 0, 1, 1};
    std::vector<std::shared_ptr<rt::Primitive>> prims = rt::CreateTriangleMesh(tris, 3);
    auto scene = std::make_shared<rt::BVH>(prims, 4);
    std::array<float, 3> orig = {0.25f, 0.25f, 5.0f}, dir = {0.f, 0.f, -1.f};
    rt::Ray ray(orig, dir);
    rt::Hit hitrec;
    bool hit = scene->intersect(ray, &hitrec);
    std::printf("%d %.6f %.6f %.6f %.6f %d\n", (int)hit, hitrec.t, hitrec.pos[0], hitrec.pos[1], hitrec.pos[2], hitrec.primID);
    rt::Ray miss(std::array<float, 3>{5, 5, 5}, std::array<float, 3>{0, 0, 1});
    rt::Hit h2;
    hit = scene->intersect(miss, &h2);
    std::printf("%d %g %d\n", (int)hit, h2.t, h2.primID);
    auto s2 = rt::CreateBVH(tris, 3);
    rt::Hit h3;
    std::printf("%d\n", (int)s2->intersect(ray, &h3));
    return 0;
}
