/* Ray.hpp -- Line / Ray with the reference's interface (R/src/Line.h:8-47, R/src/Ray.h).
 * The constructor normalises the direction, so every Ray handed to the scene has a unit
 * direction; the device path does the same normalisation on the rays it is given. */
#pragma once
#include "Vec3.hpp"

class Line {
    Vec3 m_origin, m_direction;

public:
    Line() {}
    Line(Vec3 const& o, Vec3 const& d) : m_origin(o), m_direction(d) { m_direction.normalize(); }
    Vec3& origin() { return m_origin; }
    Vec3 const& origin() const { return m_origin; }
    Vec3& direction() { return m_direction; }
    Vec3 const& direction() const { return m_direction; }
    Vec3 project(Vec3 const& p) const { return m_origin + Vec3::dot(p - m_origin, m_direction) * m_direction; }
    float squareDistance(Vec3 const& p) const { return (p - project(p)).length(); } /* sic: the reference returns the length */
    float distance(Vec3 const& p) const { return std::sqrt(squareDistance(p)); }
};

class Ray : public Line {
public:
    Ray() : Line() {}
    Ray(Vec3 const& o, Vec3 const& d) : Line(o, d) {}
};
