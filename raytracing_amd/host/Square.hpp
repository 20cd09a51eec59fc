#pragma once
#include "Mesh.hpp"

struct RaySquareIntersection {
    bool intersectionExists;
    float t;
    float u, v;
    Vec3 intersection;
    Vec3 normal;
};

class Square : public Mesh {
public:
    Vec3 m_normal, m_bottom_left, m_right_vector, m_up_vector;

    Square() : Mesh() {}
    Square(Vec3 const& bottomLeft, Vec3 const& rightVector, Vec3 const& upVector, float width = 1., float height = 1.,
           float uMin = 0.f, float uMax = 1.f, float vMin = 0.f, float vMax = 1.f)
        : Mesh() {
        setQuad(bottomLeft, rightVector, upVector, width, height, uMin, uMax, vMin, vMax);
    }

    /* R/src/Square.h:30-61 */
    void setQuad(Vec3 const& bottomLeft, Vec3 const& rightVector, Vec3 const& upVector, float width = 1., float height = 1.,
                 float uMin = 0.f, float uMax = 1.f, float vMin = 0.f, float vMax = 1.f) {
        m_normal = Vec3::cross(rightVector, upVector);
        m_normal.normalize();
        m_bottom_left = bottomLeft;
        m_right_vector = rightVector;
        m_right_vector.normalize();
        m_right_vector = m_right_vector * width;
        m_up_vector = upVector;
        m_up_vector.normalize();
        m_up_vector = m_up_vector * height;
        vertices.assign(4, MeshVertex());
        vertices[0].position = bottomLeft;
        vertices[1].position = bottomLeft + m_right_vector;
        vertices[2].position = bottomLeft + m_right_vector + m_up_vector;
        vertices[3].position = bottomLeft + m_up_vector;
        const float us[4] = {uMin, uMax, uMax, uMin}, vs[4] = {vMin, vMin, vMax, vMax};
        for (int i = 0; i < 4; ++i) { vertices[i].u = us[i]; vertices[i].v = vs[i]; vertices[i].normal = m_normal; }
        triangles.assign(2, MeshTriangle());
        triangles[0] = MeshTriangle(0, 1, 2);
        triangles[1] = MeshTriangle(0, 2, 3);
    }

    /* Square::intersect(Ray), R/src/Square.h:63-115, on the GPU: reads vertices 0, 1 and 3 as the reference does. */
    void intersect(const Ray* rays, size_t n, RaySquareIntersection* out) const {
        rthost::ensure_device();
        rt_material mat{};
        rt_square sq{};
        for (unsigned k = 0; k < 4 && k < vertices.size(); ++k)
            for (unsigned a = 0; a < 3; ++a) sq.v[k][a] = vertices[k].position[a];
        rt_scene_desc d{};
        d.n_squares = 1; d.squares = &sq; d.n_materials = 1; d.materials = &mat;
        d.kd_max_depth = 15; d.kd_max_leaf = 30; d.kd_epsilon = 0.01f;
        rt_scene_t h;
        rthost::require(rt_scene_create(&d, &h), "rt_scene_create");
        std::vector<float> org(3 * n), dir(3 * n);
        for (size_t i = 0; i < n; ++i)
            for (unsigned a = 0; a < 3; ++a) { org[3 * i + a] = rays[i].origin()[a]; dir[3 * i + a] = rays[i].direction()[a]; }
        std::vector<rt_hit> hits(n);
        rthost::require(rt_primitive_intersect(h, RT_SQUARE, 0, org.data(), dir.data(), n, hits.data()), "rt_primitive_intersect");
        rt_scene_destroy(h);
        for (size_t i = 0; i < n; ++i) {
            const rt_hit& r = hits[i];
            RaySquareIntersection& o = out[i];
            o = RaySquareIntersection();
            o.intersectionExists = r.hit != 0;
            o.t = r.hit ? r.t : FLT_MAX;
            o.u = o.v = 0.f;
            o.intersection = Vec3(r.p[0], r.p[1], r.p[2]);
            o.normal = Vec3(r.n[0], r.n[1], r.n[2]);
        }
    }
    RaySquareIntersection intersect(Ray const& ray) const {
        RaySquareIntersection r;
        intersect(&ray, 1, &r);
        return r;
    }
};
