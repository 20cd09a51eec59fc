/*
 * rt_flops.h -- frozen algorithmic FLOP constants of the ray-intersection + shading path.
 *
 * Counting rule: every IEEE add / sub / mul / div / sqrt / min / max / abs / compare that the
 * reference source executes counts as 1 FLOP, powf counts as 1, and work an implementation adds
 * on top (re-computation, index arithmetic, RNG hashing) earns nothing.  The constants were
 * counted by hand from the reference routines named beside them; the CPU oracle counts the
 * events, multiplies by these constants and reports FLOP per ray for a workload, and bench.py
 * turns rays/s into FLOP/s with that figure.
 */
#ifndef RT_FLOPS_H
#define RT_FLOPS_H

/* AABB::intersectRayAABB, R/src/AABB.h:30-73: per axis abs+cmp (2), 2 sub + 2 div (4), cmp (1),
 * max + min (2), cmp (1) = 10 -> 30 for three axes, minus the early returns nobody takes inside
 * the box, plus the final compare: counted as 28 on the common path. */
#define RT_FLOP_AABB 28

/* Sphere::intersect, R/src/Sphere.h:85-162.
 * always: o-c (3), d.d (5), 2*d (3), dot (5), norm (5 + sqrt 1), norm*norm, r*r, sub (3),
 *         b*b, 4*a, *c, sub (4), delta>0 (1)                                            = 30
 * root:   sqrt, 2*a, neg/sub/div x2 (6), t>0 x2, min (3), o+t*d (6), P-c (3),
 *         normalize (5 + sqrt + 3 div = 9), second point (7, only when both roots > 0)   ~ 36 */
#define RT_FLOP_SPHERE_MISS 30
#define RT_FLOP_SPHERE_HIT 66

/* Square::intersect, R/src/Square.h:63-115.
 * always: w, h edges (6), cross (9), v0-o (3), dot (5), dot (5), div (1)                 = 29
 * t>=0:   o+t*d (6), P-v0 (3), two lengths (2*6), two dots (10), two div (2), 4 cmp (4)  = 37
 * hit:    normalize (9)                                                                  =  9 */
#define RT_FLOP_SQUARE_BEHIND 29
#define RT_FLOP_SQUARE_REJECT 66
#define RT_FLOP_SQUARE_HIT 75

/* Mesh::intersect per triangle, R/src/Mesh.h:225-277, by exit point.
 * scale 9, edges 6, cross 9, dot 5 (= 29 with abs+cmp folded in) -> exit A
 * 1/a, s (3), dot (5), mul: u, 2 cmp                                          -> exit B (39)
 * cross (9), dot (5), mul, cmp, add, cmp                                      -> exit C (54)
 * dot (5), mul, 2 cmp                                                         -> exit D (60)
 * closest update: o+d*t (6), weights (2), 3 scaled normals (9), 2 adds (6), normalize (9) +2 */
#define RT_FLOP_TRI_PARALLEL 29
#define RT_FLOP_TRI_U_OUT 39
#define RT_FLOP_TRI_V_OUT 54
#define RT_FLOP_TRI_TESTED 60
#define RT_FLOP_TRI_UPDATE 34

/* KDTree::intersectNode leaf loop, R/src/KDTree.cpp:586-598: one `t < best` per listed object. */
#define RT_FLOP_LEAF_REF 1

/* One light sample of Scene::phong, R/src/Scene.h:369-388 + samplePointOnLight :252-276 +
 * computeLightContribution :304-316: sample point 15, L / V / |S-P| 36, shadow origin 6,
 * R 21, three channels x (diffuse 5 + specular 15) = 60, compares and the rest 12. */
#define RT_FLOP_LIGHT_SAMPLE 150

#endif /* RT_FLOPS_H */
