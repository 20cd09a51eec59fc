/* AABB.hpp -- axis-aligned box with the reference's fields (R/src/AABB.h:10-28, 76-84).  The
 * slab test itself (intersectRayAABB) runs on the device for every ray; the host only needs the
 * box as data. */
#pragma once
#include <algorithm>
#include <limits>
#include "Vec3.hpp"

struct AABB {
    Vec3 maxCoor, minCoor;
    AABB() {
        const float inf = std::numeric_limits<float>::infinity();
        maxCoor = Vec3(-inf, -inf, -inf);
        minCoor = Vec3(inf, inf, inf);
    }
    void expandByPoint(const Vec3& p) {
        for (unsigned a = 0; a < 3; ++a) {
            minCoor[a] = std::min(minCoor[a], p[a]);
            maxCoor[a] = std::max(maxCoor[a], p[a]);
        }
    }
};
