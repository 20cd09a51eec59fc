/* Sphere.hpp / Square.hpp -- the two analytic primitives with the reference's fields
 * (R/src/Sphere.h:8-42, R/src/Square.h:8-61).  Their intersect() routines are device code
 * (sphere_t / square_t in csrc/rt_device.h) reached through rt_primitive_intersect; a Scene that
 * contains them answers ray queries with the same code. */
#pragma once
#include "Mesh.hpp"

struct RaySphereIntersection {
    bool intersectionExists;
    float t;
    float theta, phi;
    Vec3 intersection;
    Vec3 secondintersection;
    Vec3 normal;
};

class Sphere : public Mesh {
public:
    Vec3 m_center;
    float m_radius = 0.f;
    Sphere() : Mesh() {}
    Sphere(Vec3 c, float r) : Mesh(), m_center(c), m_radius(r) {}
    void build_arrays() override {} /* the reference tessellates a preview mesh here; nothing to do headless */

    /* Sphere::intersect(Ray), R/src/Sphere.h:85-162, on the GPU (batch form first: the single-ray form uploads the
     * sphere, asks one question and frees it again).  secondintersection, which nothing reads, stays zero. */
    void intersect(const Ray* rays, size_t n, RaySphereIntersection* out) const {
        rthost::ensure_device();
        rt_material mat{};
        rt_sphere sp{};
        for (unsigned a = 0; a < 3; ++a) sp.center[a] = m_center[a];
        sp.radius = m_radius;
        rt_scene_desc d{};
        d.n_spheres = 1; d.spheres = &sp; d.n_materials = 1; d.materials = &mat;
        d.kd_max_depth = 15; d.kd_max_leaf = 30; d.kd_epsilon = 0.01f;
        rt_scene_t h;
        rthost::require(rt_scene_create(&d, &h), "rt_scene_create");
        std::vector<float> org(3 * n), dir(3 * n);
        for (size_t i = 0; i < n; ++i)
            for (unsigned a = 0; a < 3; ++a) { org[3 * i + a] = rays[i].origin()[a]; dir[3 * i + a] = rays[i].direction()[a]; }
        std::vector<rt_hit> hits(n);
        rthost::require(rt_primitive_intersect(h, RT_SPHERE, 0, org.data(), dir.data(), n, hits.data()), "rt_primitive_intersect");
        rt_scene_destroy(h);
        for (size_t i = 0; i < n; ++i) {
            const rt_hit& r = hits[i];
            RaySphereIntersection& o = out[i];
            o = RaySphereIntersection();
            o.intersectionExists = r.hit != 0;
            o.t = r.hit ? r.t : FLT_MAX;
            o.theta = o.phi = 0.f;
            o.intersection = Vec3(r.p[0], r.p[1], r.p[2]);
            o.normal = Vec3(r.n[0], r.n[1], r.n[2]);
        }
    }
    RaySphereIntersection intersect(Ray const& ray) const {
        RaySphereIntersection r;
        intersect(&ray, 1, &r);
        return r;
    }
};
