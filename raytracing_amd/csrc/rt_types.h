/*
 * rt_types.h -- device-side scene layout shared by the host builder (rt_build.cpp) and the HIP
 * kernels (rt_api.hip).
 *
 * Everything a ray needs that does not depend on the ray is computed once on the host with the
 * same binary32 operations the reference evaluates per ray (host code is built with
 * -ffp-contract=off), so reading it back is bit-identical to recomputing it:
 *   square: edge vectors, their cross product, its normalisation, and the rectangle test turned
 *           into interval bounds (R/src/Square.h:80-84, 98-104, 108-109)
 *   sphere: radius * radius (R/src/Sphere.h:104)
 *   triangle: the three vertices scaled by 1.000001f and the two edges (R/src/Mesh.h:220-234)
 * Records are 16-byte multiples so a wave-uniform record is fetched with scalar dwordx4 loads.
 */
#ifndef RT_TYPES_H
#define RT_TYPES_H

#include <stdint.h>

struct DSphere {  /* 32 B */
    float c[3]; float r;
    float r2; int32_t mat; int32_t pad[2];
};

struct DSquare {  /* 80 B */
    float v0[3]; float w_lo;  /* rectangle test along w: w_lo <= dot(P - v0, w) < w_hi, see below */
    float w[3]; float w_hi;
    float h[3]; int32_t mat;
    float n[3]; float h_lo;   /* n = cross(w, h), not normalised */
    float nn[3]; float h_hi;  /* nn = normalize(n) */
};
/* The reference accepts a plane hit when 0 < x / |w| && x / |w| < |w| with x = dot(P - v0, w)
 * (R/src/Square.h:101-104; same along h).  A correctly rounded division by a positive constant is
 * monotonic in x, so each pair of comparisons is an interval test on x itself: w_lo is the
 * smallest float x with fl(x / |w|) > 0 and w_hi the smallest float x with fl(x / |w|) >= |w|.
 * The host finds both by bisection over the float bit patterns using the real division, which
 * makes the interval test exact and removes two divisions per candidate hit.  A degenerate edge
 * (|w| zero, infinite or NaN) can never pass the reference's test; it gets the empty interval. */

struct DMaterial {  /* 48 B */
    float kd[3]; float shininess;
    float ks[3]; float ior;
    int32_t type; int32_t pad[3];
};

struct DLight {  /* 80 B */
    float pos[3]; int32_t type;
    float color[3]; float pad0;
    float p0[3]; float pad1;
    float p1[3]; float pad2;
    float p2[3]; float pad3;
};

struct DTri {  /* 48 B: one Moller-Trumbore test reads exactly this */
    float v0[3]; uint32_t i0;  /* i0..i2: vertex ids into DeviceScene::normals (global) */
    float e1[3]; uint32_t i1;
    float e2[3]; uint32_t i2;
};

struct DMesh {  /* 64 B */
    uint32_t first_tri, n_tris;
    int32_t mat;
    int32_t bvh_root;  /* index into DeviceScene::bvh, -1 = brute force over the triangle range */
    float bs_c[3];     /* with a BVH: a sphere around the padded root box; a ray whose line misses it, or */
    float bs_r2;       /* that leaves it behind, skips the traversal (squared radius, rounded up)         */
    float lo[3]; int32_t bvh4_root; /* the padded root box itself; bvh4_root: index into DeviceScene::bvh4, the tree the rays walk */
    float hi[3]; float pad1;
};

/* Triangle BVH node (binary, 64 B = one cache-line half, four dwordx4 loads): the boxes of BOTH
 * children, so one fetch decides which of them to visit.  Child k: n[k] == 0 -> inner node c[k];
 * n[k] > 0 -> leaf = triangles [c[k], c[k] + n[k]) of DeviceScene::tris (global positions).
 * Boxes are padded on the host; the reference has no such structure (Mesh::intersect is a
 * brute-force loop, R/src/Mesh.h:225) -- it only has to visit every triangle the loop would accept. */
struct DBvhNode {
    float lo0[3]; int32_t c0;
    float hi0[3]; int32_t n0;
    float lo1[3]; int32_t c1;
    float hi1[3]; int32_t n1;
};

/* The tree the rays walk: four children per node (128 B, eight dwordx4 loads), made from the binary tree by giving every
 * node at an even depth the children of its inner children.  One fetch then decides among four subtrees, and a path
 * from the root has half as many dependent fetches.  Two binary records back to back: children 0, 1 in half[0],
 * children 2, 3 in half[1]; same child encoding; an unused slot is an empty child (c < 0, n == 0) with an empty box. */
struct DBvh4Node {
    DBvhNode half[2];
};

/* The same node in 64 bytes (four dwordx4 loads, two nodes per cache line): what the rays fetch.  The four child boxes are
 * stored relative to the node's own box -- the union of its children -- with 8 bits per bound: bound = lo[a] + q * scale[a],
 * lower bounds rounded down, upper bounds rounded up, so a quantised box encloses the box of the 128-byte node (whose pad it
 * inherits) and can only make a walk visit more, never less.  q[b] holds bound b (lo x, lo y, lo z, hi x, hi y, hi z) of the
 * four children, one byte each (child k in byte k: v_cvt_f32_ubyte<k>).  child[k]: >= 0 inner node, < 0 a leaf, ~child =
 * first triangle * 8 + count (the traversal's stack entries), RT_BVH4Q_EMPTY for an unused slot. */
#define RT_BVH4Q_EMPTY ((int32_t)0x80000000)
struct DBvh4Q {
    float lo[3];
    float scale[3];
    uint32_t q[6];
    int32_t child[4];
};

/* Shadow rays of a quad light against a square, decided without the square's arithmetic.  A sample of the light is
 * S(u, v) = (1 - u - v) p0 + u p1 + v p2 (R/src/Scene.h:252-276), its shadow ray runs from so = P + 0.001 N along S - P for
 * the length |S - P|, i.e. to S + 0.001 N.  The numerator of Square::intersect, dot(v0 - X, n) (R/src/Square.h:86-88), is
 * affine in X, so at the far end it is g(u, v) = h0 + u e1 + v e2 up to 0.001 |n| and rounding; where the ray crosses the plane
 * follows from the values at both ends.  With s = +-1 the side of the plane the origin lies on (s num > a_sign in every lane):
 *   s g > t_miss    both ends on the same side, the far one by a margin: the reference's t is negative, or >= |S - P|
 *   s g < -t_hit    the far end lies beyond the plane by a margin: 0 < t < |S - P|; the crossing point projects between the
 *                   projections of the two ends (the rectangle test is affine too), so when the origin's own projections lie
 *                   inside the accepted intervals (|dot(so - v0, w) - cw| < ow, same along the other edge with ck, ok: rounding
 *                   only) and so do the sample's (|w0 + u w1 + v w2| < rw, |k0 + u k1 + v k2| < rk: rounding and the 0.001 N
 *                   between the sample and the end of its ray), the reference accepts the hit
 * and anything else takes the reference's arithmetic.  The margins cover 0.001 |n|, the rounding of every float involved for
 * points within DeviceScene::cert_radius of the origin, and 1e-5 of the largest numerator (rt_build.cpp derives them).
 * g_min / g_max: g at the light's corners; a square with s g_min > t_miss (s > 0) or g_max < -t_miss (s < 0) is missed by every
 * sample.  All NaN for a point light: no comparison holds and nothing is decided. */
struct DLightSq {  /* 80 B */
    float h0, e1, e2, t_miss;
    float g_min, g_max, a_sign, t_hit;
    float w0, w1, w2, rw;
    float k0, k1, k2, rk;
    float cw, ck, ow, ok;
};

/* Object-level kd-tree node, R/src/KDTree.h:12-29, children by index, leaf refs by range. */
struct DKdNode {  /* 32 B */
    int32_t leaf; int32_t axis; float cut; int32_t left;
    int32_t right; int32_t first_ref; int32_t n_refs; int32_t depth;
};

struct DRef { int32_t type; int32_t object; };

/* kd-tree with internal nodes: everything a leaf's loop needs of reference r in one 96-byte record stored at r -- one fetch per
 * object instead of reference -> record, and one scalar fetch for the whole wave where its lanes are in the same leaf. */
struct DLeafRec {
    int32_t type; int32_t object; int32_t pad[2];
    union { DSphere s; DSquare q; } u; /* the record of a sphere / a square; zeros for a mesh (found through `object`) */
};

struct DeviceScene {
    const DSphere* spheres;
    const DSquare* squares;
    const DMesh* meshes;
    const DTri* tris;           /* per mesh: original order (brute force) or BVH leaf order */
    const uint32_t* tri_orig;   /* mesh-local original index of tris[i]: tIndex and the tie-break */
    const DBvhNode* bvh;        /* binary trees as built (kept for verification) */
    const DBvh4Node* bvh4;      /* their four-wide form with full-precision boxes (made from bvh, kept for verification and refit) */
    const DBvh4Q* bvh4q;        /* the same nodes, same indices, with quantised boxes: what bvh_walk traverses */
    const float* normals;  /* xyz per vertex */
    const DMaterial* materials;
    const DLight* lights;
    const DKdNode* kd;
    const DRef* refs;
    const DLeafRec* leaf_recs;  /* kd-tree with internal nodes only: the record of reference r, stored at r */
    /* per (light l, square q), at [l n_squares + q]: what decides, from the barycentric coordinates of a light sample alone,
     * whether its shadow ray can meet the square (DLightSq; null: every ray tests every square) */
    const DLightSq* light_sq;
    uint32_t n_spheres, n_squares, n_meshes, n_lights;
    uint32_t n_kd, n_refs;
    float box_lo[3], box_hi[3];  /* root box = Scene::computeGlobalAABB */
    float kd_epsilon;
    int32_t root_is_leaf;
    float cert_radius; /* light_sq holds for shadow origins within this distance of the world origin (the root box lies inside) */
};

/* Inverted camera matrices, R/src/matrixUtilities.h:188-217, done once on the host in double. */
struct DCamera {
    double proj_inv[16];
    double mv_inv[16];
    double depth_near;
    float eye[3];
    float pad;
};

#endif
