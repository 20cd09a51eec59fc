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
};
