/* Sphere.hpp / Square.hpp -- the two analytic primitives with the reference's fields
 * (R/src/Sphere.h:8-42, R/src/Square.h:8-61).  Their intersect() routines are device code
 * (sphere_t / square_t in csrc/rt_device.h); a Scene that contains them answers ray queries. */
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
};
