/* Trackball.hpp -- virtual trackball quaternions with the interface of R/src/Trackball.h
 * (trackball, add_quats, build_rotmatrix, axis_to_quat; algorithm: G. Bell's deformed-sphere
 * trackball, R/src/Trackball.cpp:164-345).  Float state, double intermediates where the
 * reference's C expressions promote. */
#pragma once
#include <cmath>

namespace tb_detail {
const float kBallSize = 0.8f;
inline float project_to_sphere(float r, float x, float y) {
    float d = std::sqrt((double)(x * x + y * y)), z;
    if (d < r * 0.70710678118654752440) z = std::sqrt((double)(r * r - d * d)); /* on the sphere */
    else { float t = r / 1.41421356237309504880; z = t * t / d; }               /* on the hyperbola */
    return z;
}
}  // namespace tb_detail

inline void axis_to_quat(float a[3], float phi, float q[4]) {
    float len = std::sqrt((double)(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]));
    for (int i = 0; i < 3; ++i) a[i] *= 1.0 / len;
    float s = std::sin(phi / 2.0);
    for (int i = 0; i < 3; ++i) q[i] = a[i] * s;
    q[3] = std::cos(phi / 2.0);
}

inline void trackball(float q[4], float p1x, float p1y, float p2x, float p2y) {
    if (p1x == p2x && p1y == p2y) { q[0] = q[1] = q[2] = 0.f; q[3] = 1.f; return; }
    float p1[3] = {p1x, p1y, tb_detail::project_to_sphere(tb_detail::kBallSize, p1x, p1y)};
    float p2[3] = {p2x, p2y, tb_detail::project_to_sphere(tb_detail::kBallSize, p2x, p2y)};
    float axis[3] = {p2[1] * p1[2] - p2[2] * p1[1], p2[2] * p1[0] - p2[0] * p1[2], p2[0] * p1[1] - p2[1] * p1[0]};
    float d[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
    float t = std::sqrt((double)(d[0] * d[0] + d[1] * d[1] + d[2] * d[2])) / (2.0 * tb_detail::kBallSize);
    if (t > 1.0) t = 1.0;
    if (t < -1.0) t = -1.0;
    float phi = 2.0 * std::asin((double)t);
    axis_to_quat(axis, phi, q);
}

/* composes two rotations; renormalises every 97 calls like the reference */
inline void add_quats(float q1[4], float q2[4], float dest[4]) {
    static int count = 0;
    float t1[3], t2[3], t3[3], tf[4];
    for (int i = 0; i < 3; ++i) { t1[i] = q1[i] * q2[3]; t2[i] = q2[i] * q1[3]; }
    t3[0] = q2[1] * q1[2] - q2[2] * q1[1];
    t3[1] = q2[2] * q1[0] - q2[0] * q1[2];
    t3[2] = q2[0] * q1[1] - q2[1] * q1[0];
    for (int i = 0; i < 3; ++i) tf[i] = t3[i] + (t1[i] + t2[i]);
    tf[3] = q1[3] * q2[3] - (q1[0] * q2[0] + q1[1] * q2[1] + q1[2] * q2[2]);
    for (int i = 0; i < 4; ++i) dest[i] = tf[i];
    if (++count > 97) {
        count = 0;
        float mag = std::sqrt((double)(dest[0] * dest[0] + dest[1] * dest[1] + dest[2] * dest[2] + dest[3] * dest[3]));
        for (int i = 0; i < 4; ++i) dest[i] /= mag;
    }
}

/* R/src/Trackball.cpp:323-345 */
inline void build_rotmatrix(float m[4][4], float q[4]) {
    m[0][0] = 1.0 - 2.0 * (q[1] * q[1] + q[2] * q[2]);
    m[0][1] = 2.0 * (q[0] * q[1] - q[2] * q[3]);
    m[0][2] = 2.0 * (q[2] * q[0] + q[1] * q[3]);
    m[1][0] = 2.0 * (q[0] * q[1] + q[2] * q[3]);
    m[1][1] = 1.0 - 2.0 * (q[2] * q[2] + q[0] * q[0]);
    m[1][2] = 2.0 * (q[1] * q[2] - q[0] * q[3]);
    m[2][0] = 2.0 * (q[2] * q[0] - q[1] * q[3]);
    m[2][1] = 2.0 * (q[1] * q[2] + q[0] * q[3]);
    m[2][2] = 1.0 - 2.0 * (q[1] * q[1] + q[0] * q[0]);
    m[0][3] = m[1][3] = m[2][3] = m[3][0] = m[3][1] = m[3][2] = 0.0;
    m[3][3] = 1.0;
}
