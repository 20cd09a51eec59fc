/*
 * rt_device.h -- device functions of the intersection + shading path (gfx950, wave64).
 *
 * Numerics contract: every float expression below is evaluated with the operation order of the
 * reference routine cited beside it, with plain IEEE binary32 add / mul / div / sqrt (the file is
 * compiled with -ffp-contract=off and hipcc's default correctly rounded divide / sqrt), because
 * the reference binary is x86-64 SSE code without fused multiply-add.  Hit / miss decisions and
 * the closest object are therefore bit-exact; colours differ only through powf.
 *
 * Execution model: one lane follows one ray / one path.  The scene tables are read with
 * wave-uniform indices (loop counters), so the compiler fetches them through the scalar cache
 * into SGPRs and the vector ALUs only see per-ray data.
 */
#ifndef RT_DEVICE_H
#define RT_DEVICE_H

#include <hip/hip_runtime.h>

#include "../../include/rtamd.h"
#include "rt_types.h"

#define RT_DEV __device__ __forceinline__

/* Attribution switches (RT_EXP_*) leave a part of the work out to see what it costs; every one of them renders WRONG images.
 * They only compile in builds that say so (tools/ab.sh passes -DRT_ATTRIBUTION_BUILD), never in the library proper. */
#if (defined(RT_EXP_NO_MESH_CLOSEST) || defined(RT_EXP_NO_MESH_SHADOW) || defined(RT_EXP_NO_WALK) || defined(RT_EXP_CHEAP_RNG) || \
     defined(RT_EXP_NO_RENORM) || defined(RT_EXP_FLOAT_POW) || defined(RT_EXP_NO_LIGHT_TERM) || defined(RT_EXP_NO_SPHERES) ||      \
     defined(RT_EXP_NO_SQUARES) || defined(RT_EXP_NO_SHADE) || defined(RT_EXP_NO_SHADOW_TESTS) || defined(RT_EXP_SKIP_SQUARE) || defined(RT_EXP_COUNT_OPEN) || defined(RT_EXP_DROP_OPEN)) && !defined(RT_ATTRIBUTION_BUILD)
#error "RT_EXP_* switches change the image: attribution builds only (-DRT_ATTRIBUTION_BUILD, see tools/ab.sh)"
#endif

struct f3 { float x, y, z; };

RT_DEV f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_DEV f3 ld3(const float* p) { return mk3(p[0], p[1], p[2]); }
RT_DEV f3 ld3c(const float* p) { /* same through the constant address space (scene tables) */
    const __attribute__((address_space(4))) float* q = (const __attribute__((address_space(4))) float*)p;
    return mk3(q[0], q[1], q[2]);
}
RT_DEV f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_DEV f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_DEV f3 operator*(float s, f3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
RT_DEV f3 operator/(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
RT_DEV float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RT_DEV f3 cross3(f3 a, f3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
RT_DEV float len3(f3 a) { return __builtin_sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }

/* ---- correctly rounded sqrt / divide without the range scaffolding --------------------------
 * hipcc expands an IEEE fp32 divide into v_div_scale x2, v_rcp, two FMAs refining the reciprocal,
 * a multiply, three FMAs correcting the quotient, v_div_fmas and v_div_fixup; an IEEE sqrt into a
 * 2^32 pre-scale for tiny inputs, v_sqrt, a one-ulp-down / one-ulp-up residual test and a
 * special-value select.  For operands in a "plain" range (|x| in [2^-40, 2^40] below) v_div_scale
 * returns its input unscaled with VCC = 0, v_div_fmas is then an ordinary FMA, v_div_fixup passes
 * the quotient through and the sqrt pre-scale is not taken -- so the helpers below run the compiler's
 * remaining instructions (the division: one of its two quotient corrections, see div_plain) and
 * return bit-identical results, while a normalisation shares one refined reciprocal between its
 * three quotients.  Anything outside the range (zero, subnormal, huge, NaN components) takes the
 * ordinary operators. */
RT_DEV float sqrt_plain(float x) { /* x in [2^-96, FLT_MAX] */
    float s = __builtin_amdgcn_sqrtf(x);
    float s_dn = __int_as_float(__float_as_int(s) - 1), s_up = __int_as_float(__float_as_int(s) + 1);
    float e_dn = __builtin_fmaf(-s_dn, s, x), e_up = __builtin_fmaf(-s_up, s, x);
    s = (0.f >= e_dn) ? s_dn : s;
    s = (0.f < e_up) ? s_up : s;
    return s;
}
RT_DEV float rcp_refined(float b) {
    float r = __builtin_amdgcn_rcpf(b);
    return __builtin_fmaf(__builtin_fmaf(-b, r, 1.0f), r, r);
}
/* a / b, r = rcp_refined(b): the product a * r and ONE correction by the exact remainder.  The compiler's sequence corrects twice;
 * once is enough here: on gfx950 rcp_refined returns the correctly rounded reciprocal of every significand, and with it the
 * single step yields the correctly rounded quotient for every pair of significands -- checked over all 2^23 x 2^23 pairs
 * (7.04e13 quotients, tools/micro/div_check.hip; the uncorrected product is off by one in 27 % of them), and scaling either
 * operand by a power of two scales every intermediate exactly while nothing leaves the plain range. */
RT_DEV float div_plain(float a, float b, float r) {
    float q = a * r;
    return __builtin_fmaf(__builtin_fmaf(-b, q, a), r, q);
}
/* normalize() of R/src/Vec3.h:34: L = sqrt(x^2 + y^2 + z^2), three true divides by L; also returns L */
RT_DEV f3 norm3_len(f3 a, float& L) {
    float q = a.x * a.x + a.y * a.y + a.z * a.z;
    float lo = fminf(fminf(__builtin_fabsf(a.x), __builtin_fabsf(a.y)), __builtin_fabsf(a.z));
    float hi = fmaxf(fmaxf(__builtin_fabsf(a.x), __builtin_fabsf(a.y)), __builtin_fabsf(a.z));
    /* the short sequences run unconditionally (on operands outside the plain range their result is simply
     * discarded): the common case then crosses no taken branch, which costs a wave more than these 25
     * instructions; every component plain => q, L and the quotients are too */
    L = sqrt_plain(q);
    float r = rcp_refined(L);
    f3 res = mk3(div_plain(a.x, L, r), div_plain(a.y, L, r), div_plain(a.z, L, r));
    if (__builtin_expect(!(lo >= 0x1p-40f && hi <= 0x1p40f), 0)) {
        L = __builtin_sqrtf(q);
        res = mk3(a.x / L, a.y / L, a.z / L);
    }
    return res;
}
RT_DEV f3 norm3(f3 a) { float L; return norm3_len(a, L); }
/* normalize() of a vector that is already unit length up to rounding (the Ray constructor's second
 * normalisation, a mirrored or refracted unit direction, a stored normal): same result as norm3, cheaper
 * when q = x^2 + y^2 + z^2 lies within 4096 float steps of 1.  There the correctly rounded square root
 * follows from the bit pattern alone: q = 1 + m 2^-23 has sqrt 1 + (m/2) 2^-23 - m^2 2^-48 + ..., which
 * rounds to 1 + floor(m/2) 2^-23 (the second-order term, < 2^-25, only decides the halfway case, downwards);
 * q = 1 - k 2^-24 has sqrt 1 - (k/2) 2^-24 - k^2 2^-51 - ..., which rounds to 1 - ceil(k/2) 2^-24.  Both
 * are bits(L) = (bits(q) >> 1) + 0x1fc00000 (arithmetic: 0x3f800000 is even, so halving the distance from
 * it with floor rounding is one shift).  The quotients are those of norm3_len.  The device self-test
 * checks every q of that window against the IEEE square root. */
RT_DEV f3 norm3_unit(f3 a) {
    float q = a.x * a.x + a.y * a.y + a.z * a.z;
    float lo = fminf(fminf(__builtin_fabsf(a.x), __builtin_fabsf(a.y)), __builtin_fabsf(a.z));
    float L = __int_as_float((__float_as_int(q) >> 1) + 0x1fc00000);
    float r = rcp_refined(L);
    f3 res = mk3(div_plain(a.x, L, r), div_plain(a.y, L, r), div_plain(a.z, L, r));
    if (__builtin_expect(!((uint32_t)(__float_as_int(q) - (0x3f800000 - 4096)) <= 8192u && lo >= 0x1p-40f), 0)) res = norm3(a);
    return res;
}
RT_DEV float comp(f3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
RT_DEV float max_std(float a, float b) { return (a < b) ? b : a; } /* std::max, NaN behaviour included */
RT_DEV float min_std(float a, float b) { return (b < a) ? b : a; } /* std::min */

#define RT_INF __builtin_inff()

/* Scene tables never change while a kernel runs.  Reading them through the constant address space
 * tells the compiler so: a load whose address is wave-uniform then becomes a scalar load (s_load
 * into SGPRs through the scalar cache) instead of 64 identical vector loads; a divergent address
 * still gets an ordinary vector load. */
template <class T>
RT_DEV T ldc(const T* p) {
    static_assert(sizeof(T) % 4 == 0, "scene records are dword multiples");
    const __attribute__((address_space(4))) uint32_t* q = (const __attribute__((address_space(4))) uint32_t*)p;
    union { T v; uint32_t w[sizeof(T) / 4]; } u;
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 4; ++i) u.w[i] = q[i]; /* merged into dwordx4 / x8 loads */
    return u.v;
}

/* per-lane context: work counters (summed per workgroup at the end) and the base of this lane's
 * slice of the LDS traversal stack (element i lives at stack[i * stack_stride]) */
struct Cnt { uint32_t rays, nodes; uint32_t* stack; uint32_t bvh_nodes, bvh_tris; /* BVH nodes fetched, triangles tested */
             float* frames; uint32_t frame_stride; /* LDS part of the path-frame stack (trace_path), may be null when unused */
             uint32_t skipped; /* of `rays`: shadow queries the reference issues but whose result it discards (skip_shade) */
             uint32_t stack_stride; /* threads of the workgroup: element i of the lane's stack lives at stack[i * stack_stride] */ };
/* kernel variants: bit 0 = the object kd-tree has internal nodes, bit 1 = some mesh has a triangle BVH.
 * Code a scene cannot reach is compiled out of its variant, which keeps the common variant's registers low. */
#define RT_M_KD 1
#ifdef RT_LANE_STATS /* debug build only: how many lanes run each marked region ([2 id] visits by a wave, [2 id + 1] lanes) */
__device__ unsigned long long rt_dbg_lanes[64];
#define RT_LANE_PROBE(id)                                                                                   \
    do {                                                                                                    \
        const unsigned long long probe_m = __ballot(1);                                                     \
        if (__lane_id() == (unsigned)__builtin_ctzll(probe_m)) {                                             \
            atomicAdd(&rt_dbg_lanes[2 * (id)], 1ull); atomicAdd(&rt_dbg_lanes[2 * (id) + 1], (unsigned long long)__popcll(probe_m)); \
        }                                                                                                   \
    } while (0)
#else
#define RT_LANE_PROBE(id) do {} while (0)
#endif
#define RT_M_BVH 2
#ifndef RT_STACK_LDS
#define RT_STACK_LDS 32      /* entries per lane, all in LDS: the host bounds the BVH depth to this (kBvhMaxDepth) */
#endif
#define RT_STACK_STRIDE 256  /* workgroup size of the lane-per-ray kernels: lane-interleaved, conflict-free LDS layout */

/* wave-level helpers: number of set bits of `mask` below `lane`; ordering of this wave's LDS accesses (the LDS
 * unit serves one wave's requests in order: the fences only stop the compiler from moving them) */
RT_DEV uint32_t wave_rank(uint64_t mask, uint32_t lane) {
    return (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
/* The traversal stacks, the pool table and the LDS path frames are reached through these types.  Built with
 * -DRT_LDS_TYPED they say "LDS" and every access is a ds_ instruction; by default they are generic, and where the code
 * chooses between an LDS entry and an overflow entry the compiler merges both into one flat access.  Measured, typed
 * against generic: 100 k-triangle frame 36.1 / 36.7 ms, 10 M-triangle frame 101.2 / 102.0 ms, but the Cornell frame
 * 22.9 / 21.6 ms and an eighth of the 10 M-triangle frame in passes of 16 spp 36 - 41 / 21 - 23 ms: the typed build
 * allocates registers differently and spills 976 instead of 576 bytes per lane in the variant with a BVH. */
#ifdef RT_LDS_TYPED
typedef __attribute__((address_space(3))) uint32_t rt_lds_u32;
typedef __attribute__((address_space(3))) float rt_lds_f32;
#else
typedef uint32_t rt_lds_u32;
typedef float rt_lds_f32;
#endif
RT_DEV rt_lds_u32* lds_words(uint32_t* p) { return (rt_lds_u32*)p; }
RT_DEV void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

/* closest hit so far: enough to rebuild position and normal of the winner afterwards */
struct HitRec {
    float t;    /* +inf while nothing was hit */
    int type;   /* RT_* or -1 */
    int obj;
    int tri;    /* global triangle id for a mesh hit */
    float u, v; /* barycentrics of a mesh hit */
};

struct FullHit {
    bool hit;
    int type, obj, tri;
    float t;
    f3 p, n;
};

/* ---- AABB::intersectRayAABB, R/src/AABB.h:30-73 ---------------------------------------- */
RT_DEV bool box_test(const float lo[3], const float hi[3], f3 o, f3 d, float& t_in, float& t_out) {
    t_in = -RT_INF;
    t_out = RT_INF;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float oo = comp(o, a), dd = comp(d, a);
        if (__builtin_fabsf(dd) < 1e-8f) {
            if (oo < lo[a] || oo > hi[a]) return false;
            continue;
        }
        float t1 = (lo[a] - oo) / dd, t2 = (hi[a] - oo) / dd;
        if (t1 > t2) { float s = t1; t1 = t2; t2 = s; }
        t_in = max_std(t_in, t1);
        t_out = min_std(t_out, t2);
        if (t_in > t_out) return false;
    }
    return t_out >= 0.0f;
}

/* A ray that starts inside the root box (faces included) always passes AABB::intersectRayAABB:
 * on an axis with |d| >= 1e-8 the numerators (lo - o) <= 0 <= (hi - o) are exact, so the two
 * quotients have opposite signs (or are zero), t_entry stays <= 0 and t_exit >= 0, `t_entry >
 * t_exit` never holds and the final `t_exit >= 0` does (-0 >= 0 is true); on an axis with
 * |d| < 1e-8 the origin is within the slab.  Checking the origin replaces the six divisions for
 * every shadow and bounce ray. */
RT_DEV bool inside_box(const float lo[3], const float hi[3], f3 o) {
    return o.x >= lo[0] && o.x <= hi[0] && o.y >= lo[1] && o.y <= hi[1] && o.z >= lo[2] && o.z <= hi[2];
}

/* ---- Sphere::intersect, R/src/Sphere.h:85-162: distance only ---------------------------- */
RT_DEV bool sphere_t(const DSphere& s, f3 o, f3 d, float& t) {
    f3 oc = o - ld3(s.c);
    float a = dot3(d, d);
    float b = dot3(2.f * d, oc);
    float n = len3(oc);
    float c = (n * n) - s.r2;
    float delta = b * b - 4.f * a * c;
    if (delta > 0.f) {
        float sq = __builtin_sqrtf(delta);
        float a2 = 2.f * a;
        float t1 = (-b - sq) / a2, t2 = (-b + sq) / a2;
        if (t1 > 0.f && t2 > 0.f) t = min_std(t1, t2);
        else if (t1 > 0.f) t = t1;
        else if (t2 > 0.f) t = t2;
        else return false;
        return true;
    }
    if (delta == 0.f) {
        t = -b / (2.f * a);
        return t > 0.f;
    }
    return false;
}

/* ---- Square::intersect, R/src/Square.h:63-115: distance only ----------------------------
 * The two `0 < x / len && x / len < len` checks are evaluated as interval tests on x (bounds made
 * exact on the host, see DSquare). */
RT_DEV bool square_in_rect(const DSquare& q, f3 bp) {
    float x = dot3(bp, ld3(q.w)), y = dot3(bp, ld3(q.h));
    return x >= q.w_lo && x < q.w_hi && y >= q.h_lo && y < q.h_hi;
}
RT_DEV bool square_t(const DSquare& q, f3 o, f3 d, float& t) {
    f3 v0 = ld3(q.v0), n = ld3(q.n);
    float tt = dot3(v0 - o, n) / dot3(d, n);
    if (tt >= 0.f && square_in_rect(q, (o + tt * d) - v0)) { t = tt; return true; }
    return false; /* t < 0, rectangle missed, or t is NaN (undefined in the reference) */
}

/* ---- one triangle of Mesh::intersect, R/src/Mesh.h:225-257 ------------------------------ */
RT_DEV bool tri_t(const DTri& tr, f3 o, f3 d, float& t, float& u, float& v) {
    f3 e1 = ld3(tr.e1), e2 = ld3(tr.e2);
    f3 h = cross3(d, e2);
    float a = dot3(e1, h);
    if (__builtin_fabsf(a) < 1e-6f) return false;
    float f = 1.0f / a;
    f3 s = o - ld3(tr.v0);
    u = f * dot3(s, h);
    if (u < 0.0f || u > 1.0f) return false;
    f3 q = cross3(s, e1);
    v = f * dot3(d, q);
    if (v < 0.0f || u + v > 1.0f) return false;
    t = f * dot3(e2, q);
    return true;
}

/* ---- per-mesh triangle BVH traversal -------------------------------------------------------
 * Replaces the brute-force loop of Mesh::intersect (R/src/Mesh.h:225-277) for large meshes without
 * changing its answer: every triangle test is the reference's arithmetic (tri_t), the winner is
 * the smallest t and, on equal t, the smallest original index -- what the loop's strict `<` keeps.
 * The boxes only decide which triangles get tested; they are padded on the host and a subtree is
 * skipped only when its box is missed or starts beyond the best t so far.
 * while-while form: descend through inner nodes until a leaf entry is reached, then test the
 * leaf's triangles; one 64-byte node holds both child boxes.  The stack lives in LDS, lane-interleaved;
 * the host bounds the tree depth to its RT_STACK_LDS entries.
 * The box tests only steer the traversal, so they use v_rcp_f32 (1 ulp) instead of IEEE divisions: the
 * boxes are padded by 1e-4 of the mesh diagonal, four orders of magnitude more than that error.
 * Stack entries: >= 0 inner node index; < 0 leaf, ~entry = first_triangle * 8 + count (count <= 4). */
RT_DEV bool slab(const float lo[3], const float hi[3], f3 o, f3 id, float limit, float& tn) {
    float x1 = (lo[0] - o.x) * id.x, x2 = (hi[0] - o.x) * id.x;
    float y1 = (lo[1] - o.y) * id.y, y2 = (hi[1] - o.y) * id.y;
    float z1 = (lo[2] - o.z) * id.z, z2 = (hi[2] - o.z) * id.z;
    /* fminf / fmaxf drop a NaN operand (0 * inf when the ray runs inside a face plane): that axis
     * then does not constrain the interval, which is the conservative reading */
    tn = fmaxf(fmaxf(fminf(x1, x2), fminf(y1, y2)), fmaxf(fminf(z1, z2), 0.f));
    float tf = fminf(fminf(fmaxf(x1, x2), fmaxf(y1, y2)), fminf(fmaxf(z1, z2), limit));
    return tn <= tf;
}
/* Can the ray come near the mesh at all?  A division-free test against the mesh's bounding sphere (b and c2
 * carry a rounding error of a few 2^-24 * c2, d is unit to 2^-23: the 1e-6 * c2 slack keeps every rejection
 * certain; NaN / inf operands reject nothing), then the padded root box.  limit = FLT_MAX for a closest-hit
 * query, the distance to the light for a shadow ray. */
/* 1 / d per axis for the box tests, kept finite: a direction component of exactly zero would give an infinite factor, and the
 * node test evaluates a bound as code * (scale * id) + (lo - o) * id, where inf - inf is NaN and the axis would stop
 * constraining anything -- such a ray walks every node along that axis (one lane was seen to visit 5 500 nodes where its
 * neighbours visit 180).  With +-1e18 instead, an origin outside the slab still yields a distance beyond every limit. */
RT_DEV float bvh_inv(float x) { return fminf(fmaxf(__builtin_amdgcn_rcpf(x), -1e18f), 1e18f); }
RT_DEV f3 bvh_inv_dir(f3 d) { return mk3(bvh_inv(d.x), bvh_inv(d.y), bvh_inv(d.z)); }
template <bool ANY>
RT_DEV bool bvh_reaches(const DMesh& m, f3 o, f3 d, float limit) {
    const f3 oc = ld3(m.bs_c) - o;
    const float b = dot3(oc, d), c2 = dot3(oc, oc);
    const float r2 = m.bs_r2 + 1e-6f * c2;
    if (c2 - b * b > r2) return false;   /* the line misses the sphere */
    if (b < 0.f && c2 > r2) return false; /* the sphere lies behind the origin */
    if (ANY && b > limit) { const float e = b - limit; if (e * e > r2) return false; } /* ... or beyond the light */
    float tn;
    return slab(m.lo, m.hi, o, bvh_inv_dir(d), limit, tn);
}
/* The traversal proper; the caller has checked bvh_reaches.  It walks the four-wide tree in its 64-byte form (DBvh4Q): one
 * fetch of four dwordx4 decides among four subtrees, the hit ones are visited nearest first.  Stack: RT_STACK_LDS entries in LDS, and what a
 * path could need beyond them (three pushes per level, RT_STACK_LDS / 2 levels at most) in a small per-lane array. */
#define RT_STACK_OVER 16 /* a power of two: the index wraps by masking */
/* The four child boxes of a wide node against a ray: how many the ray reaches within `limit`, and their stack entries
 * (>= 0 inner node, < 0 leaf) ordered by entry distance, e0 the nearest. */
RT_DEV int bvh4_order(const DBvh4Q& n, f3 o, f3 id, float limit, int& e0, int& e1, int& e2, int& e3) {
    /* entry / exit distance along an axis of a bound stored as code q: ((lo + q scale) - o) id = q (scale id) + (lo - o) id */
    const f3 A = mk3(n.scale[0] * id.x, n.scale[1] * id.y, n.scale[2] * id.z);
    const f3 B = mk3((n.lo[0] - o.x) * id.x, (n.lo[1] - o.y) * id.y, (n.lo[2] - o.z) * id.z);
    /* Which face of a box the ray meets first along an axis is the sign of its direction there (A has that sign, or is zero and
     * both faces give B; the factors are finite, see bvh_inv, so no NaN can arise): the near and the far codes of all four
     * children are picked once per axis instead of ordering every pair of distances. */
    const bool px = id.x >= 0.f, py = id.y >= 0.f, pz = id.z >= 0.f;
    const uint32_t nx = px ? n.q[0] : n.q[3], fx = px ? n.q[3] : n.q[0];
    const uint32_t ny = py ? n.q[1] : n.q[4], fy = py ? n.q[4] : n.q[1];
    const uint32_t nz = pz ? n.q[2] : n.q[5], fz = pz ? n.q[5] : n.q[2];
    float tn[4];
    bool h[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        auto code = [&](uint32_t w) { return (float)((w >> (8 * k)) & 255u); }; /* byte k of w as a float: one v_cvt_f32_ubyte<k> */
        tn[k] = fmaxf(fmaxf(__builtin_fmaf(code(nx), A.x, B.x), __builtin_fmaf(code(ny), A.y, B.y)), fmaxf(__builtin_fmaf(code(nz), A.z, B.z), 0.f));
        const float tf = fminf(fminf(__builtin_fmaf(code(fx), A.x, B.x), __builtin_fmaf(code(fy), A.y, B.y)), fminf(__builtin_fmaf(code(fz), A.z, B.z), limit));
        h[k] = tn[k] <= tf && n.child[k] != RT_BVH4Q_EMPTY;
    }
    e0 = n.child[0]; e1 = n.child[1]; e2 = n.child[2]; e3 = n.child[3];
    /* entry distances, +inf for a child that is not visited; sorted ascending by a five-comparator network */
    float k0 = h[0] ? tn[0] : RT_INF, k1 = h[1] ? tn[1] : RT_INF, k2 = h[2] ? tn[2] : RT_INF, k3 = h[3] ? tn[3] : RT_INF;
    auto order = [](float& ka, int& ea, float& kb, int& eb) {
        const bool sw = kb < ka;
        const float kt = sw ? kb : ka; kb = sw ? ka : kb; ka = kt;
        const int et = sw ? eb : ea; eb = sw ? ea : eb; ea = et;
    };
    order(k0, e0, k1, e1); order(k2, e2, k3, e3); order(k0, e0, k2, e2); order(k1, e1, k3, e3); order(k1, e1, k2, e2);
    return (int)h[0] + (int)h[1] + (int)h[2] + (int)h[3];
}
template <bool ANY>
RT_DEV bool bvh_walk(const DeviceScene& sc, const DMesh& m, f3 o, f3 d, float dist, Cnt& cnt, float& best_t, int& best_tri,
                     float& best_u, float& best_v) {
    best_t = dist; /* ANY: the distance to the light; closest hit: FLT_MAX, the reference's initial closest t, or what a hit has to beat */
    const f3 id = bvh_inv_dir(d);
    int sp = 0;
    bool found = false;
    uint32_t best_orig = 0xffffffffu;
    int over[RT_STACK_OVER];
    auto push = [&](int e) {
        if (sp < RT_STACK_LDS) lds_words(cnt.stack)[sp * cnt.stack_stride] = (uint32_t)e; else over[(sp - RT_STACK_LDS) & (RT_STACK_OVER - 1)] = e;
        ++sp;
    };
    auto pop = [&]() -> int {
        --sp;
        if (sp < RT_STACK_LDS) return (int)lds_words(cnt.stack)[sp * cnt.stack_stride];
        return over[(sp - RT_STACK_LDS) & (RT_STACK_OVER - 1)];
    };
    int cur = m.bvh4_root;
    bool done = false;
    auto visit = [&](const DBvh4Q& n) { /* the four child boxes of one node: where to go next, what to keep for later */
        cnt.bvh_nodes += 1; /* counted in 64-byte units */
        int e0, e1, e2, e3;
        const int hits = bvh4_order(n, o, id, best_t, e0, e1, e2, e3);
        if (hits > 3) push(e3);
        if (hits > 2) push(e2);
        if (hits > 1) push(e1);
        if (hits > 0) cur = e0;
        else if (sp > 0) cur = pop();
        else done = true;
    };
    /* the root is the same node for every lane: its address is wave-uniform, so this fetch is a scalar load */
    visit(ldc(&sc.bvh4q[m.bvh4_root]));
    for (;;) {
        while (!done && cur >= 0) visit(ldc(&sc.bvh4q[cur])); /* inner nodes */
        if (done) break;
        { /* leaf */
            int code = ~cur;
            int first = code >> 3, count = code & 7;
            cnt.bvh_tris += (uint32_t)count;
            /* two records per round, both requested before the first is used */
            for (int k0 = 0; k0 < count; k0 += 2) {
                const int k1 = k0 + 1 < count ? k0 + 1 : k0;
                const DTri ta = ldc(&sc.tris[first + k0]);
                const DTri tb = ldc(&sc.tris[first + k1]);
                for (int j = 0; j < 2; ++j) {
                    const int k = j ? k1 : k0;
                    if (j && k1 == k0) break;
                    float t, u, v;
                    if (tri_t(j ? tb : ta, o, d, t, u, v) && t > 1e-6f) {
                        if (ANY) { if (t < dist) return true; }
                        else if (t <= best_t) { /* the original index is only needed to break a tie */
                            uint32_t orig = ldc(&sc.tri_orig[first + k]);
                            if (t < best_t || orig < best_orig) {
                                best_t = t; best_tri = first + k; best_u = u; best_v = v; best_orig = orig; found = true;
                            }
                        }
                    }
                }
            }
            if (sp > 0) cur = pop();
            else break;
        }
    }
    return found;
}

template <bool ANY>
RT_DEV bool bvh_traverse(const DeviceScene& sc, const DMesh& m, f3 o, f3 d, float dist, Cnt& cnt, float& best_t, int& best_tri,
                         float& best_u, float& best_v) {
    best_t = dist;
    if (!bvh_reaches<true>(m, o, d, best_t)) return false; /* the bounds within `dist`, for either kind of query */
    return bvh_walk<ANY>(sc, m, o, d, dist, cnt, best_t, best_tri, best_u, best_v);
}

/* ---- Mesh::intersect closest loop, R/src/Mesh.h:225-277 (brute force over the slice) ----- */
/* `limit`: the caller only uses a hit with t < limit (the closest hit among the objects tested before), so a mesh with
 * a BVH is walked no further than that: a ray inside the glass sphere, or one that ends on the mirror sphere in front of
 * the mesh, fetches no node at all.  A hit the brute-force loop would report beyond the limit is not reported, which
 * the caller cannot tell apart. */
template <int M>
RT_DEV bool mesh_closest(const DeviceScene& sc, const DMesh& m, f3 o, f3 d, Cnt& cnt, float& best_t, int& best_tri, float& best_u, float& best_v,
                         float limit = 3.402823466e+38f) {
    if ((M & RT_M_BVH) && m.bvh_root >= 0) return bvh_traverse<false>(sc, m, o, d, limit, cnt, best_t, best_tri, best_u, best_v);
    bool found = false;
    best_t = 3.402823466e+38f; /* FLT_MAX */
    for (uint32_t i = 0; i < m.n_tris; ++i) {
        float t, u, v;
        if (tri_t(ldc(&sc.tris[m.first_tri + i]), o, d, t, u, v) && t > 1e-6f && t < best_t) {
            best_t = t; best_tri = (int)(m.first_tri + i); best_u = u; best_v = v; found = true;
        }
    }
    return found;
}
template <int M>
RT_DEV bool mesh_any(const DeviceScene& sc, const DMesh& m, f3 o, f3 d, float dist, Cnt& cnt) {
    if ((M & RT_M_BVH) && m.bvh_root >= 0) {
        float t, u, v;
        int tri;
        return bvh_traverse<true>(sc, m, o, d, dist, cnt, t, tri, u, v);
    }
    for (uint32_t i = 0; i < m.n_tris; ++i) {
        float t, u, v;
        if (tri_t(ldc(&sc.tris[m.first_tri + i]), o, d, t, u, v) && t > 1e-6f && t < dist) return true;
    }
    return false;
}

/* Certain misses of the two analytic tests, decided before their square root / division (leaves of a kd-tree with
 * internal nodes test up to 30 objects per ray, most of them missed).
 * Sphere: the reference's c is fl(|oc|)^2 - r^2, within 4e-7 |oc|^2 of dot(oc, oc) - r^2; the discriminant formed
 * from the latter is off by less than 1e-6 (b^2 + 4a (|oc|^2 + r^2)), so below minus ten times that the exact one
 * is negative.  Square: num * rcp(den) is within 1.5 * 2^-23 of the exact quotient, so below -1e-30 the correctly
 * rounded t is negative (the rule shadow_batch uses).  NaN operands fail both comparisons and take the exact path. */
RT_DEV bool sphere_surely_missed(const DSphere& s, f3 o, f3 d) {
    f3 oc = o - ld3(s.c);
    float a = dot3(d, d), b = dot3(2.f * d, oc), q = dot3(oc, oc);
    float four_ac = 4.f * a * (q - s.r2);
    return b * b - four_ac < -1e-5f * (b * b + __builtin_fabsf(four_ac) + 4.f * a * (q + s.r2));
}
RT_DEV bool square_surely_behind(const DSquare& q, f3 o, f3 d) {
    f3 n = ld3(q.n);
    return dot3(ld3(q.v0) - o, n) * __builtin_amdgcn_rcpf(dot3(d, n)) < -1e-30f;
}

/* ---- leaf loop of KDTree::intersectNode, R/src/KDTree.cpp:586-598 ------------------------
 * The references of a leaf keep the order of KDTree::buildKDTree's list -- spheres, squares, meshes (every split is a stable
 * partition) -- so a leaf is walked in two parts: its spheres and squares (by each lane for itself, or by the wave together
 * from scalar-loaded records), then its meshes from `meshes_at` on, in one place for both (kd_closest). */
RT_DEV void leaf_keep(HitRec& best, int type, int object, float t, int tri, float u, float v) {
    if (best.type < 0 || t < best.t) { best.t = t; best.type = type; best.obj = object; best.tri = tri; best.u = u; best.v = v; }
}
/* Every lane walks the references of its own leaf (the lanes of the wave may be in different leaves: vector loads). */
RT_DEV bool leaf_closest(const DeviceScene& sc, int first, int count, f3 o, f3 d, HitRec& best, int& meshes_at) {
    bool any = false;
    int r = first;
    for (; r < first + count; ++r) {
        const DLeafRec* rec = &sc.leaf_recs[r];
        const int type = ldc(&rec->type), object = ldc(&rec->object);
        float t = 0.f;
        bool ok;
        if (type == RT_SPHERE) { const DSphere s = ldc(&rec->u.s); ok = !sphere_surely_missed(s, o, d) && sphere_t(s, o, d, t); }
        else if (type == RT_SQUARE) { const DSquare q = ldc(&rec->u.q); ok = !square_surely_behind(q, o, d) && square_t(q, o, d, t); }
        else break;
        if (ok) { any = true; leaf_keep(best, type, object, t, -1, 0.f, 0.f); }
    }
    meshes_at = r;
    return any;
}
/* The same loop for a scene whose root is a leaf (every built-in scene): the leaf list is all
 * spheres, then all squares, then all meshes -- the order KDTree::buildKDTree collects them in -- so the
 * three tables are walked directly, and each record is requested one iteration before it is used:
 * the scalar load then completes behind the arithmetic of the current object instead of stalling
 * the wave. */
RT_DEV uint32_t next_index(uint32_t i, uint32_t n) { return i + 1 < n ? i + 1 : i; }
/* Calls f(record, index) for every record of a table.  Two record buffers alternate: the scalar
 * load of the next record is issued into the buffer that is not in use, before the arithmetic on
 * the current one starts, so its latency overlaps that arithmetic.  (With a single buffer the
 * register allocator reuses the SGPRs and the load can only start once the current record is dead.) */
template <class Rec, class F>
RT_DEV void for_each_record(const Rec* tab, uint32_t n, F&& f) {
    if (!n) return;
    Rec a = ldc(&tab[0]);
    for (uint32_t i = 0; i < n; i += 2) {
        const Rec b = ldc(&tab[next_index(i, n)]);
        f(a, i);
        a = ldc(&tab[i + 2 < n ? i + 2 : n - 1]);
        if (i + 1 < n) f(b, i + 1);
    }
}
template <int M>
RT_DEV void root_closest(const DeviceScene& sc, f3 o, f3 d, Cnt& cnt, HitRec& best) {
    for_each_record(sc.spheres, sc.n_spheres, [&](const DSphere& rec, uint32_t i) {
        float t;
        if (sphere_t(rec, o, d, t) && (best.type < 0 || t < best.t)) { best.t = t; best.type = RT_SPHERE; best.obj = (int)i; }
    });
    for_each_record(sc.squares, sc.n_squares, [&](const DSquare& rec, uint32_t i) {
        float t;
        if (square_t(rec, o, d, t) && (best.type < 0 || t < best.t)) { best.t = t; best.type = RT_SQUARE; best.obj = (int)i; }
    });
#ifdef RT_EXP_NO_MESH_CLOSEST /* attribution builds: what the meshes cost in the closest-hit queries (wrong images) */
    if (0)
#endif
    for (uint32_t i = 0; i < sc.n_meshes; ++i) {
        float t, u, v;
        int tri;
        if (mesh_closest<M>(sc, ldc(&sc.meshes[i]), o, d, cnt, t, tri, u, v, best.type < 0 ? 3.402823466e+38f : best.t) && (best.type < 0 || t < best.t)) {
            best.t = t; best.type = RT_MESH; best.obj = (int)i; best.tri = tri; best.u = u; best.v = v;
        }
    }
}
/* shadow query on a leaf: "closest hit exists and its t < dist" == "some object reports t < dist" */
template <int M>
RT_DEV bool leaf_any(const DeviceScene& sc, int first, int count, f3 o, f3 d, float dist, Cnt& cnt) {
    for (int r = first; r < first + count; ++r) {
        DRef ref = ldc(&sc.refs[r]);
        float t;
        if (ref.type == RT_SPHERE) {
            if (sphere_t(ldc(&sc.spheres[ref.object]), o, d, t) && t < dist) return true;
        } else if (ref.type == RT_SQUARE) {
            if (square_t(ldc(&sc.squares[ref.object]), o, d, t) && t < dist) return true;
        }
        else { if (mesh_any<M>(sc, ldc(&sc.meshes[ref.object]), o, d, dist, cnt)) return true; }
    }
    return false;
}
/* The leaf loop for lanes that are all in the same leaf (first, count wave-uniform): the records arrive by scalar loads, each
 * requested while the one before it is tested, and the branch on the object's kind is the same for all lanes. */
RT_DEV bool leaf_closest_uniform(const DeviceScene& sc, int first, int count, f3 o, f3 d, HitRec& best, int& meshes_at) {
    bool any = false;
    int at = first + count;
    for_each_record(sc.leaf_recs + first, (uint32_t)count, [&](const DLeafRec& rec, uint32_t i) {
        float t = 0.f;
        bool ok;
        if (rec.type == RT_SPHERE) ok = !sphere_surely_missed(rec.u.s, o, d) && sphere_t(rec.u.s, o, d, t);
        else if (rec.type == RT_SQUARE) ok = !square_surely_behind(rec.u.q, o, d) && square_t(rec.u.q, o, d, t);
        else { at = min(at, first + (int)i); return; }
        if (ok) { any = true; leaf_keep(best, rec.type, rec.object, t, -1, 0.f, 0.f); }
    });
    meshes_at = at;
    return any;
}

/* ---- KDTree::intersectNode for internal nodes, R/src/KDTree.cpp:600-676 -------------------
 * The recursion is unrolled into a loop with one frame per "ray crosses the plane" level.  The
 * boolean each call returns (something in that subtree was hit) is kept, because the reference
 * uses it to decide whether the far child is visited. */
#define RT_KD_MAX_DEPTH 16
/* The lanes of a wave run this together.  Every lane keeps its own traversal state; what is shared is the leaf loop: per trip
 * the leaf of the first waiting lane is taken, and when at least half of the waiting lanes are in that very leaf (rays of one
 * pixel, shadow rays of one point) they test its objects together from scalar-loaded records while the others wait for a
 * later trip; otherwise every waiting lane walks its own leaf at once.  Which lanes share a trip changes no lane's sequence
 * of tests. */
template <int M>
RT_DEV bool kd_closest(const DeviceScene& sc, f3 o, f3 d, float t_in, float t_out, Cnt& cnt, HitRec& best) {
    /* frame = far child id * 4 + state, plane distance, exit distance of the level.
     * state 0: near subtree running; 1 / 2: far subtree running after near returned false / true */
    int f_node[RT_KD_MAX_DEPTH];
    float f_split[RT_KD_MAX_DEPTH], f_out[RT_KD_MAX_DEPTH];
    int sp = 0, node = 0;
    bool ret = false, done = false;
    for (;;) {
        int first = 0, count = 0;
        while (!done) { /* descend until a leaf */
            DKdNode nd = ldc(&sc.kd[node]);
            if (nd.leaf) { first = nd.first_ref; count = nd.n_refs; break; }
            float da = comp(d, nd.axis), oa = comp(o, nd.axis);
            float t_split = (__builtin_fabsf(da) < sc.kd_epsilon) ? RT_INF : (nd.cut - oa) / da;
            int near_id = (da >= 0.f) ? nd.left : nd.right;
            int far_id = (da >= 0.f) ? nd.right : nd.left;
            if (t_split <= t_in) node = far_id;
            else if (t_split >= t_out) node = near_id;
            else {
                if (sp >= RT_KD_MAX_DEPTH) { done = true; break; } /* unreachable: the host rejects deeper trees */
                f_node[sp] = far_id * 4; f_split[sp] = t_split; f_out[sp] = t_out; ++sp;
                node = near_id; t_out = t_split;
            }
        }
        const uint64_t waiting = __ballot(!done);
        if (waiting == 0ull) return ret;
        const int lead = (int)__builtin_ctzll(waiting);
        const int lead_node = __builtin_amdgcn_readlane(node, lead);
        const int lead_first = __builtin_amdgcn_readlane(first, lead), lead_count = __builtin_amdgcn_readlane(count, lead);
        const bool with_lead = !done && node == lead_node;
        bool tested;
        int meshes_at = 0;
        if (2 * __popcll(__ballot(with_lead)) >= __popcll(waiting)) {
            tested = with_lead;
            if (with_lead) ret = leaf_closest_uniform(sc, lead_first, lead_count, o, d, best, meshes_at);
        } else {
            tested = !done;
            if (!done) ret = leaf_closest(sc, first, count, o, d, best, meshes_at);
        }
        if (!tested) continue;
        if (sc.n_meshes) /* the leaf's meshes, last in its list */
            for (int r = meshes_at; r < first + count; ++r) {
                const int object = ldc(&sc.leaf_recs[r].object);
                float t = 0.f, u = 0.f, v = 0.f;
                int tri = -1;
                if (mesh_closest<M>(sc, ldc(&sc.meshes[object]), o, d, cnt, t, tri, u, v)) { ret = true; leaf_keep(best, RT_MESH, object, t, tri, u, v); }
            }
        bool resumed = false;
        while (sp > 0) { /* hand `ret` back to the waiting levels */
            int k = sp - 1;
            int state = f_node[k] & 3;
            if (state == 0) {
                bool hit_near = ret;
                if (!hit_near || best.t > f_split[k]) {
                    node = f_node[k] >> 2; t_in = f_split[k]; t_out = f_out[k];
                    f_node[k] |= hit_near ? 2 : 1;
                    resumed = true;
                    break;
                }
                --sp; /* this level returns hit_near || false */
            } else {
                ret = (state == 2) || ret;
                --sp;
            }
        }
        if (!resumed) done = true;
    }
}

/* ---- Scene::computeIntersection, R/src/Scene.h:169-205 via KDTree::intersect :679-687 ----- */
template <int M>
RT_DEV HitRec closest_rec(const DeviceScene& sc, f3 o, f3 d, Cnt& cnt) {
    RT_LANE_PROBE(9);
    ++cnt.rays;
    HitRec best;
    best.t = RT_INF; best.type = -1; best.obj = -1; best.tri = -1; best.u = 0.f; best.v = 0.f;
    float t_in, t_out;
    if (sc.n_kd == 0) return best;
    if (!(M & RT_M_KD)) { /* the root is a leaf (every built-in scene): only pass / fail of the box matters */
        if (!inside_box(sc.box_lo, sc.box_hi, o) && !box_test(sc.box_lo, sc.box_hi, o, d, t_in, t_out)) return best;
        root_closest<M>(sc, o, d, cnt, best);
    } else {
        if (!box_test(sc.box_lo, sc.box_hi, o, d, t_in, t_out)) return best;
        kd_closest<M>(sc, o, d, t_in, t_out, cnt, best);
    }
    return best;
}

/* position / normal of the winner, exactly what its own intersect routine would have stored */
RT_DEV FullHit finish_hit(const DeviceScene& sc, const HitRec& r, f3 o, f3 d) {
    FullHit h;
    h.hit = r.type >= 0; h.type = r.type; h.obj = r.obj; h.tri = r.tri; h.t = r.t;
    h.p = mk3(0.f, 0.f, 0.f); h.n = mk3(0.f, 0.f, 0.f);
    if (!h.hit) return h;
    h.p = o + r.t * d;
    if (r.type == RT_SPHERE) h.n = norm3(h.p - ld3c(sc.spheres[r.obj].c));                  /* Sphere.h:120-122 */
    else if (r.type == RT_SQUARE) h.n = ld3c(sc.squares[r.obj].nn);                         /* Square.h:108-109 */
    else {                                                                                  /* Mesh.h:267-272 */
        DTri tr = ldc(&sc.tris[r.tri]);
        f3 n0 = ld3c(sc.normals + 3 * (size_t)tr.i0), n1 = ld3c(sc.normals + 3 * (size_t)tr.i1), n2 = ld3c(sc.normals + 3 * (size_t)tr.i2);
        h.n = norm3(((1.0f - r.u - r.v) * n0 + r.u * n1) + r.v * n2);
    }
    return h;
}

/* shadow test of Scene::phong, R/src/Scene.h:379-384: closest hit exists and t < dist */
template <int M>
RT_DEV bool occluded(const DeviceScene& sc, f3 o, f3 d, float dist, Cnt& cnt) {
    ++cnt.rays;
    float t_in, t_out;
    if (sc.n_kd == 0 || !box_test(sc.box_lo, sc.box_hi, o, d, t_in, t_out)) return false;
    if (!(M & RT_M_KD)) return leaf_any<M>(sc, 0, (int)sc.n_refs, o, d, dist, cnt);
    HitRec best;
    best.t = RT_INF; best.type = -1; best.obj = -1; best.tri = -1; best.u = 0.f; best.v = 0.f;
    return kd_closest<M>(sc, o, d, t_in, t_out, cnt, best) && best.type >= 0 && best.t < dist;
}

/* Shadow queries of one shaded point, SB at a time (root-leaf scenes).  All rays of a batch leave the
 * same origin, so everything that depends on (origin, object) only -- the sphere's o - c and
 * |o - c|^2 - r^2, the square's dot(v0 - o, n), a triangle's o - v0 and cross(o - v0, e1) -- is
 * computed once per object and reused by every direction of the batch, each object record is fetched
 * once per batch, and the independent dependency chains give the scheduler work to interleave.
 * Every value is still produced by the reference's operations in the reference's order; the
 * answer per ray is "the closest hit exists and its t < dist", i.e. some object reports t < dist. */
/* The constant term of Sphere::intersect's quadratic, R/src/Sphere.h: (o - c).norm() * (o - c).norm() - r * r (square root,
 * then square).  All shadow rays of a shaded point share their origin, so shade computes it once per point for the first
 * RT_SPH_KEPT spheres instead of once per batch. */
#define RT_SPH_KEPT 2
RT_DEV float sphere_c(const DSphere& s, f3 o) {
    const float n = len3(o - ld3(s.c));
    return (n * n) - s.r2;
}
#ifndef RT_SB
#define RT_SB 4 /* rays per batch; the render variant of scenes without kd nodes or BVH takes RT_SB0 (see k_render) */
#endif
#ifndef RT_SB0
#define RT_SB0 2
#endif
struct SquareRules { /* per shaded point and light, the same in every lane: which squares its shadow rays still test, and how (shade) */
    uint32_t test;       /* bit q clear: no ray of this point can meet square q, it is not even fetched */
    uint32_t by_sample;  /* bit q: DLightSq decides most rays of square q from the light sample's (u, v) */
    uint32_t flipped;    /* ... with the origin on the negative side of the plane */
    uint32_t can_hit;    /* ... and certain hits may be taken (every lane's origin projects well inside the rectangle) */
    const DLightSq* rec; /* the light's row of DeviceScene::light_sq */
};
/* The squares of rules.by_sample against the light samples (u, v) of a batch, before their rays even exist: fused arithmetic
 * against margins (DLightSq), no quotient.  Sets hidden[j] when a square is known to hide ray j and returns the squares that
 * some ray of the wave still has to test with their own arithmetic -- which then runs for every ray of the wave and finds
 * again what the rules knew.  Records are requested one ahead, like the object records of the tests proper; the side of the
 * plane goes into the sign bits of the record's three coefficients (the same in every lane: scalar instructions). */
template <int SB>
RT_DEV uint32_t sample_rules(const SquareRules& rules, const float (&bu)[SB], const float (&bv)[SB], bool (&hidden)[SB]) {
    uint32_t test = rules.test;
#pragma unroll
    for (int j = 0; j < SB; ++j) hidden[j] = false;
    uint32_t left = (uint32_t)__builtin_amdgcn_readfirstlane((int)rules.by_sample);
    if (!left) return test;
    auto one = [&](const DLightSq& t, uint32_t k) {
        RT_LANE_PROBE(2);
        const int flip = ((rules.flipped >> k) & 1u) ? (int)0x80000000 : 0;
        const float h0 = __int_as_float(__float_as_int(t.h0) ^ flip), e1 = __int_as_float(__float_as_int(t.e1) ^ flip),
                    e2 = __int_as_float(__float_as_int(t.e2) ^ flip);
        const bool can_hit = ((rules.can_hit >> k) & 1u) != 0u;
        bool any_open = false;
#pragma unroll
        for (int j = 0; j < SB; ++j) {
            const float g = __builtin_fmaf(bu[j], e1, __builtin_fmaf(bv[j], e2, h0));
            bool decided = g > t.t_miss;
            if (can_hit) {
                const bool hit = g < -t.t_hit && __builtin_fabsf(__builtin_fmaf(bu[j], t.w1, __builtin_fmaf(bv[j], t.w2, t.w0))) < t.rw &&
                                 __builtin_fabsf(__builtin_fmaf(bu[j], t.k1, __builtin_fmaf(bv[j], t.k2, t.k0))) < t.rk;
                hidden[j] = hidden[j] || hit;
                decided = decided || hit;
            }
            any_open = any_open || !decided;
        }
        if (!__ballot(any_open)) test &= ~(1u << k);
    };
    uint32_t k = (uint32_t)__builtin_ctz(left);
    DLightSq a = ldc(&rules.rec[k]);
    for (;;) {
        left &= left - 1u;
        if (!left) { one(a, k); break; }
        const uint32_t k2 = (uint32_t)__builtin_ctz(left);
        const DLightSq b = ldc(&rules.rec[k2]);
        one(a, k);
        a = b; k = k2;
    }
    return test;
}
template <int M, int SB>
RT_DEV void shadow_batch(const DeviceScene& sc, f3 o, bool o_inside, const f3 (&d)[SB], const float (&dist)[SB], bool (&occ)[SB], uint32_t& near,
                         Cnt& cnt, uint32_t sq_mask, const float (&sph_c)[RT_SPH_KEPT]) {
    /* sq_mask (the same in every lane): the squares some ray of the wave still has to test; occ[j] comes in true for a ray a
     * square is already known to hide (sample_rules) */
    bool act[SB];
    cnt.rays += SB;
#pragma unroll
    for (int j = 0; j < SB; ++j) act[j] = true;
    if (__builtin_expect(!o_inside, 0)) { /* the origin left the root box: the slab test decides, ray by ray */
#pragma unroll
        for (int j = 0; j < SB; ++j) {
            float t_in, t_out;
            act[j] = box_test(sc.box_lo, sc.box_hi, o, d[j], t_in, t_out);
        }
    }
    /* tables walked directly with one-ahead record fetches, as in root_closest */
    /* Per object the cheap part runs for all rays of the batch first, in straight-line code, and the rare
     * expensive part sits behind ONE wave-level branch per object instead of one per ray. */
#ifdef RT_EXP_NO_SPHERES /* attribution: the sphere tests of the shadow rays */
    if (0)
#endif
    for_each_record(sc.spheres, sc.n_spheres, [&](const DSphere& s, uint32_t si) {
        RT_LANE_PROBE(3);
        f3 oc = o - ld3(s.c);
        float c; /* |o - c|^2 - r^2 depends on the origin only: the caller computed it for the first spheres (sphere_c) */
        if (si < RT_SPH_KEPT) {
            c = sph_c[0];
#pragma unroll
            for (uint32_t k = 1; k < RT_SPH_KEPT; ++k) c = si == k ? sph_c[k] : c;
        } else c = sphere_c(s, o);
        float b[SB], delta[SB], bb[SB], ac4[SB];
        bool some = false;
#pragma unroll
        for (int j = 0; j < SB; ++j) {
            float a = dot3(d[j], d[j]); /* terms of the direction alone: the compiler moves them out of the loop over the spheres */
            b[j] = dot3(2.f * d[j], oc);
            bb[j] = b[j] * b[j];
            ac4[j] = 4.f * a * c;
            delta[j] = bb[j] - ac4[j];
            some = some || delta[j] >= 0.f;
        }
        if (!some) return; /* every line of the batch misses the sphere */
        RT_LANE_PROBE(4);
#ifndef RT_NO_SPHERE_CERT
        /* A line that meets the sphere is not yet a ray that meets it before the light: rays that leave the sphere they start
         * on, and rays whose line reaches a sphere only behind their origin or beyond the light, are most of the lines that get
         * here.  Both cases are certain before any square root or division (a = d . d lies within 3e-7 of 1):
         *   behind: b > 0 and 4ac > 1e-5 b^2.  Then c > 0, the float discriminant is at most b^2 (1 - 1e-5 + 2^-23), its root
         *           is smaller than b, both numerators -b -+ sq are negative and neither root is > 0: the reference reports
         *           no intersection (for a zero discriminant -b / 2a < 0 says the same).
         *   beyond: b < 0, c > 0 and, with m = -b and D = 1.001 dist: m > 2.001 D (the vertex of the quadratic lies beyond D),
         *           c > D (m - 0.999 D) (the quadratic is positive at D) and m < 1000 dist (which bounds the cancellation in
         *           m - sq to a relative 1e-4): the smaller root computed in floats is larger than dist, so is the one the
         *           reference picks, and `t < dist` fails.
         * Only a batch with a ray that is neither goes on to the roots. */
        bool need = false;
#pragma unroll
        for (int j = 0; j < SB; ++j) {
            const float m = -b[j], D = 1.001f * dist[j];
            const bool behind = b[j] > 0.f && ac4[j] > 1e-5f * bb[j];
            const bool beyond = m > 2.001f * D && c * 0.999f > D * (m - 0.999f * D) && m < 1000.f * dist[j] && c > 0.f;
            need = need || (delta[j] >= 0.f && !behind && !beyond);
        }
        if (!need) return;
#endif
#pragma unroll
        for (int j = 0; j < SB; ++j) {
            float a = dot3(d[j], d[j]);
            float t = RT_INF; /* stays +inf (never < dist) when the sphere is missed */
            if (delta[j] > 0.f) {
                RT_LANE_PROBE(5);
                float sq = __builtin_sqrtf(delta[j]);
                float a2 = 2.f * a;
                float t1 = (-b[j] - sq) / a2, t2 = (-b[j] + sq) / a2;
                if (t1 > 0.f && t2 > 0.f) t = min_std(t1, t2);
                else if (t1 > 0.f) t = t1;
                else if (t2 > 0.f) t = t2;
            } else if (delta[j] == 0.f) {
                float t0 = -b[j] / (2.f * a);
                if (t0 > 0.f) t = t0;
            }
            occ[j] = occ[j] || (t < dist[j]);
        }
    });
    /* Squares: the plane distance decides "0 <= t < dist" long before its last bit matters.  num * rcp(den)
     * is within 1.5 * 2^-23 of the exact quotient (v_rcp_f32: 1 ulp, one rounding of the product), so a
     * value below -1e-30 means the correctly rounded t is negative, and a value above dist * (1 + 2^-20)
     * means it is >= dist: both are misses whatever the rectangle test would say.  Everything else
     * (including NaN, zero and denormal results) takes the reference's division. */
    float dist_hi[SB];
#pragma unroll
    for (int j = 0; j < SB; ++j) dist_hi[j] = dist[j] * 1.000001f;
    auto square = [&](const DSquare& q, uint32_t) {
        RT_LANE_PROBE(6);
        f3 v0 = ld3(q.v0), n = ld3(q.n);
        float num = dot3(v0 - o, n);
        bool maybe[SB], some = false;
#pragma unroll
        for (int j = 0; j < SB; ++j) {
            float ta = num * __builtin_amdgcn_rcpf(dot3(d[j], n));
            maybe[j] = !(ta < -1e-30f) && !(ta > dist_hi[j]);
            some = some || maybe[j];
        }
        if (!some) return;
#pragma unroll
        for (int j = 0; j < SB; ++j)
            if (maybe[j]) {
                RT_LANE_PROBE(7);
                float tt = num / dot3(d[j], n);
                if (tt >= 0.f && tt < dist[j] && square_in_rect(q, (o + tt * d[j]) - v0)) occ[j] = true;
            }
    };
    /* sq_mask (the same in every lane): bit q clear = no ray of this shaded point can meet square q (shade decides that
     * once per point), and the square is not even fetched.  Scenes with more than 32 squares test them all. */
#ifdef RT_EXP_NO_SQUARES /* attribution: the square tests of the shadow rays */
    if (0) {}
    else
#endif
    if (sc.n_squares > 32u || sq_mask == 0xffffffffu) for_each_record(sc.squares, sc.n_squares, square);
    else if (sq_mask) {
        uint32_t left = (uint32_t)__builtin_amdgcn_readfirstlane((int)sq_mask);
        uint32_t k = (uint32_t)__builtin_ctz(left);
        DSquare a = ldc(&sc.squares[k]);
        for (;;) { /* the record of the next set bit is requested before the current one is used */
            left &= left - 1u;
            if (!left) { square(a, k); break; }
            const uint32_t k2 = (uint32_t)__builtin_ctz(left);
            const DSquare b = ldc(&sc.squares[k2]);
            square(a, k);
            a = b; k = k2;
        }
    }
    for (uint32_t mi = 0; mi < sc.n_meshes; ++mi) {
        const DMesh m = ldc(&sc.meshes[mi]);
        if ((M & RT_M_BVH) && m.bvh_root >= 0) continue; /* large meshes: after everything else, see below */
        for_each_record(sc.tris + m.first_tri, m.n_tris, [&](const DTri& tr, uint32_t) {
            f3 e1 = ld3(tr.e1), e2 = ld3(tr.e2);
            f3 sv = o - ld3(tr.v0);
            f3 qv = cross3(sv, e1);
            float e2q = dot3(e2, qv);
#pragma unroll
            for (int j = 0; j < SB; ++j) {
                f3 h = cross3(d[j], e2);
                float a = dot3(e1, h);
                float f = 1.0f / a;
                float u = f * dot3(sv, h);
                float v = f * dot3(d[j], qv);
                float t = f * e2q;
                bool ok = !(__builtin_fabsf(a) < 1e-6f) && !(u < 0.0f || u > 1.0f) && !(v < 0.0f || u + v > 1.0f) && t > 1e-6f && t < dist[j];
                occ[j] = occ[j] || ok;
            }
        });
    }
#pragma unroll
    for (int j = 0; j < SB; ++j) occ[j] = occ[j] && act[j];
    /* Large meshes are not walked here: few rays come near one, and a wave would pay a whole traversal per batch
     * for a handful of them.  The rays still undecided that reach the bounds of some mesh with a BVH are reported
     * in `near` and walked later, all light samples of the shaded point together (walk_deferred). */
    near = 0;
#ifndef RT_EXP_NO_MESH_SHADOW /* attribution builds: -DRT_EXP_NO_MESH_SHADOW / _NO_WALK / _NO_MESH_CLOSEST leave a part of the */
    if (M & RT_M_BVH)         /* mesh handling out (wrong images) to see what it costs: the numbers quoted with the pooled walks  */
#else
    if (0)
#endif
        for (uint32_t mi = 0; mi < sc.n_meshes; ++mi) {
            const DMesh m = ldc(&sc.meshes[mi]);
            if (m.bvh_root < 0) continue;
#pragma unroll
            for (int j = 0; j < SB; ++j)
                if (!occ[j] && act[j] && bvh_reaches<true>(m, o, d[j], dist[j])) near |= 1u << j;
        }
}

/* ---- random numbers ---------------------------------------------------------------------- */
struct Rng {
    /* kg = k * 0x9E3779B9 mod 2^32 for the stream position k: draw k is rt_mix32(key + kg) >> 1 = rt_rand31(key, k), and
     * moving on in the stream is one addition instead of a multiplication per draw */
    uint32_t key, kg;
    RT_DEV void at(uint32_t k) { kg = k * 0x9E3779B9U; }
    RT_DEV void skip(uint32_t n) { kg += n * 0x9E3779B9U; }
    /* (float)rand() / RAND_MAX with RAND_MAX converted to float = 2^31 (R/src/Scene.h:259-260) */
#ifdef RT_EXP_CHEAP_RNG /* attribution: what the two integer multiplies and six shift / xor steps of a draw cost */
    RT_DEV float unit() { const float r = (float)(int)(((key + kg) ^ 0x5bd1e995U) >> 1) / 2147483648.f; kg += 0x9E3779B9U; return r; }
#else
    RT_DEV float unit() { const float r = (float)(int)(rt_mix32(key + kg) >> 1) / 2147483648.f; kg += 0x9E3779B9U; return r; }
#endif
};

/* Light sample i of a shaded point (Scene::samplePointOnLight, R/src/Scene.h:252-276): calls k + 2 i and
 * k + 2 i + 1 of the sample's random stream, wherever the caller is in that stream. */
RT_DEV f3 light_sample(const Rng& at_first, uint32_t i, f3 p0, f3 p1, f3 p2) {
    Rng r = at_first;
    r.skip(2u * i);
    float u = r.unit();
    float v = r.unit();
    if (u + v > 1.0f) { u = 1.0f - u; v = 1.0f - v; }
    return ((1.f - u - v) * p0 + u * p1) + v * p2;
}

/* The shadow rays of one shaded point that came near a mesh with a BVH (bit i of `near` = light sample i), walked
 * for all 16 light samples at once.  Such rays are rare (a few per hundred), which lanes hold them is uneven, and their
 * walks differ in length by an order of magnitude: walked where they arise, or dealt out in rounds of one ray per lane,
 * the wave waits for its longest walk again and again (measured on the 100 k-triangle scene: 330 rays per call, the
 * longest of a round several times the mean).  So the wave pools them: rays are numbered lane by lane, and every lane
 * of the wave keeps taking the next unanswered ray, whoever it belongs to, until none is left; who walks a ray does
 * not change its answer.  A ray is rebuilt from the random stream where it is needed (light_sample and the two
 * normalisations of shade) instead of being kept in registers.
 * Returns the bits of the rays that hit a triangle before the light. */
#ifdef RT_WALK_STATS /* debug build only: the deferred walks of a frame */
__device__ unsigned long long rt_dbg_walk[128];
#endif
#define RT_POOL_ROWS 11
#define RT_POOL_LDS (RT_STACK_LDS - RT_POOL_ROWS) /* stack entries a pooled walk keeps in LDS */
#define RT_POOL_OVER 32                           /* ... and per lane beyond them: 21 + 32 >= three pushes per level, 16 levels */
template <int M>
RT_DEV uint32_t walk_deferred(const DeviceScene& sc, f3 P, f3 o, const Rng& at_first, f3 p0, f3 p1, f3 p2, uint32_t near, Cnt& cnt) {
    const uint64_t lanes = __ballot(1);
    const uint32_t n_lanes = (uint32_t)__popcll(lanes);
    const uint32_t own = (uint32_t)__popc(near);
    uint32_t start = 0, total = 0, longest = 0; /* rays of the lanes below this one, of the wave, of its busiest lane */
#pragma unroll
    for (int b = 4; b >= 0; --b) {
        const uint64_t bit = __ballot((own >> b) & 1u);
        start += wave_rank(bit, 0) << b;
        total += (uint32_t)__popcll(bit) << b;
        const uint32_t t = longest | (1u << b);
        if (__ballot(own >= t)) longest = t;
    }
    if (total == 0) return 0;
#ifdef RT_WALK_STATS
    const uint32_t nodes_before = cnt.bvh_nodes;
    struct Report { /* at every return: rays and lanes of the call, the longest and the summed walk (64-byte node units) */
        uint32_t before, total, n_lanes; const Cnt& c; unsigned long long t0; uint32_t iters, refills; uint32_t last_kind; unsigned long long last_t; unsigned long long k_cycles[3]; uint32_t k_trips[3], k_lanes[3]; unsigned long long wait_cycles; uint32_t waits;
        __device__ ~Report() {
            const unsigned long long dt = __builtin_readcyclecounter() - t0;
            const uint32_t mine = c.bvh_nodes - before;
            uint32_t mx = mine, sm = mine;
            for (int off = 32; off > 0; off >>= 1) { mx = max(mx, (uint32_t)__shfl_xor((int)mx, off, 64)); sm += (uint32_t)__shfl_xor((int)sm, off, 64); }
            if (wave_rank(__ballot(1), __lane_id()) == 0u) {
                atomicAdd(&rt_dbg_walk[0], 1ull); atomicAdd(&rt_dbg_walk[1], (unsigned long long)total); atomicAdd(&rt_dbg_walk[2], (unsigned long long)mx);
                atomicAdd(&rt_dbg_walk[3], (unsigned long long)sm); atomicAdd(&rt_dbg_walk[4], (unsigned long long)n_lanes);
                for (int k = 0; k < 3; ++k) {
                    atomicAdd(&rt_dbg_walk[114 + k], (unsigned long long)k_trips[k]); atomicAdd(&rt_dbg_walk[117 + k], (unsigned long long)k_lanes[k]);
                    atomicAdd(&rt_dbg_walk[120 + k], k_cycles[k]);
                }
                atomicAdd(&rt_dbg_walk[123], wait_cycles); atomicAdd(&rt_dbg_walk[124], (unsigned long long)waits);
                atomicAdd(&rt_dbg_walk[5], dt); atomicAdd(&rt_dbg_walk[6], (unsigned long long)iters); atomicAdd(&rt_dbg_walk[7], (unsigned long long)refills);
                atomicAdd(&rt_dbg_walk[8 + min(total / 8u, 39u)], 1ull);        /* rays per call, buckets of 8 */
                atomicAdd(&rt_dbg_walk[48 + min(mx / 16u, 39u)], 1ull);         /* longest walk, buckets of 16 units */
                atomicAdd(&rt_dbg_walk[88 + min(total / 16u, 25u)], (unsigned long long)mx); /* longest walk summed per rays bucket */
            }
        }
    } report{nodes_before, total, n_lanes, cnt, __builtin_readcyclecounter(), 0u, 0u, 3u, 0ull, {0ull, 0ull, 0ull}, {0u, 0u, 0u}, {0u, 0u, 0u}, 0ull, 0u};
#endif
    auto ray = [&](uint32_t i, f3& d, float& dist) {
        f3 SP = light_sample(at_first, i, p0, p1, p2) - P;
        d = norm3_unit(norm3_len(SP, dist));
    };
    auto any_mesh = [&](f3 ro, f3 rd, float rdist) {
        bool hit = false;
        for (uint32_t mi = 0; mi < sc.n_meshes; ++mi) {
            const DMesh m = ldc(&sc.meshes[mi]);
            if (m.bvh_root < 0) continue;
            float t, u, v;
            int tri;
            if (!hit && bvh_reaches<true>(m, ro, rd, rdist)) hit = bvh_walk<true>(sc, m, ro, rd, rdist, cnt, t, tri, u, v);
        }
        return hit;
    };
    if (longest <= 1) { /* at most one ray per lane: nothing to even out */
        uint32_t hit = 0;
        for (uint32_t bits = near; bits; bits &= bits - 1) {
            const uint32_t i = (uint32_t)__builtin_ctz(bits);
            f3 d;
            float dist;
            ray(i, d, dist);
            if (any_mesh(o, d, dist)) hit |= 1u << i;
        }
        return hit;
    }
    /* The pool.  Every lane leaves what a ray of its is rebuilt from in a table (the upper RT_POOL_ROWS entries of the
     * wave's stack words; the walks keep RT_POOL_LDS entries and a longer per-lane overflow): rays before it in the wave's
     * numbering, its mask, shaded point, origin, random stream.  Then one loop runs until every ray is answered: a lane
     * without a ray takes the next number from a counter, finds the owner (binary search over the first row) and the
     * sample (the k-th set bit), rebuilds the ray and starts at the root; a lane with a ray does one step, a wide node and /
     * or a leaf.  A hit sets the sample's bit in the upper half of the owner's mask word. */
    const uint32_t lane = __lane_id();
    const uint32_t my = wave_rank(lanes, lane); /* position among the participating lanes */
    rt_lds_u32* const xb = lds_words(cnt.stack) - lane;
    const uint32_t stride = cnt.stack_stride;
    enum { R_START = 0, R_NEAR, R_PX, R_PY, R_PZ, R_OX, R_OY, R_OZ, R_KEY, R_K, R_NEXT };
    auto row = [&](uint32_t r, uint32_t c) -> rt_lds_u32& { return xb[(RT_POOL_LDS + r) * stride + c]; };
    row(R_START, my) = start; row(R_NEAR, my) = near;
    row(R_PX, my) = __float_as_uint(P.x); row(R_PY, my) = __float_as_uint(P.y); row(R_PZ, my) = __float_as_uint(P.z);
    row(R_OX, my) = __float_as_uint(o.x); row(R_OY, my) = __float_as_uint(o.y); row(R_OZ, my) = __float_as_uint(o.z);
    row(R_KEY, my) = at_first.key; row(R_K, my) = at_first.kg;
    if (my == 0) row(R_NEXT, 0) = n_lanes; /* rays 0 .. n_lanes - 1 are handed out by position */
    wave_lds_sync();
    bool want = true, out_of_rays = false, first = true;
    uint32_t g = my, owner = 0, mi = 0;
    f3 ro = o, rd = o, id = o;
    float rdist = 0.f;
    int cur = 0, sp = 0;
    int over[RT_POOL_OVER];
    auto push = [&](int e) {
        if (sp < RT_POOL_LDS) lds_words(cnt.stack)[sp * stride] = (uint32_t)e; else over[(sp - RT_POOL_LDS) & (RT_POOL_OVER - 1)] = e;
        ++sp;
    };
    auto pop = [&]() -> int {
        --sp;
        if (sp < RT_POOL_LDS) return (int)lds_words(cnt.stack)[sp * stride];
        return over[(sp - RT_POOL_LDS) & (RT_POOL_OVER - 1)];
    };
    auto next_mesh = [&](uint32_t from) { /* the first mesh with a BVH from `from` on that the ray comes near: its root, or the ray is done */
        for (mi = from; mi < sc.n_meshes; ++mi) {
            const DMesh m = ldc(&sc.meshes[mi]);
            if (m.bvh_root >= 0 && bvh_reaches<true>(m, ro, rd, rdist)) { cur = m.bvh4_root; sp = 0; return true; }
        }
        return false;
    };
    /* One kind of work per trip -- fetch a ray, visit a wide node, test a leaf's triangles -- and always the kind most lanes
     * are waiting for: a lane at a leaf waits while most others still descend, and the other way round, so that each pass
     * over the code of a kind serves many lanes (doing whatever each lane needs in every trip cost 8.5 k cycles per trip
     * with 42 triangle tests in the 256 places of its leaf part). */
    for (;;) {
        const uint64_t idle = __ballot(want && !out_of_rays), inner = __ballot(!want && cur >= 0), leaf = __ballot(!want && cur < 0);
        if (!(idle | inner | leaf)) break;
        const uint32_t n_idle = (uint32_t)__popcll(idle), n_inner = (uint32_t)__popcll(inner), n_leaf = (uint32_t)__popcll(leaf);
#ifdef RT_WALK_STATS
        ++report.iters;
        if (n_idle > n_inner && n_idle > n_leaf) ++report.refills;
        { /* cycles and lanes served per kind of trip: [16 + 3 kind] trips, [17 + 3 kind] lanes, [18 + 3 kind] cycles (of the trip before) */
            const unsigned long long now = __builtin_readcyclecounter();
            const uint32_t kind = (n_idle > n_inner && n_idle > n_leaf) ? 0u : (n_inner >= n_leaf ? 1u : 2u);
            if (report.last_kind < 3u) report.k_cycles[report.last_kind] += now - report.last_t;
            report.k_trips[kind] += 1u;
            report.k_lanes[kind] += kind == 0u ? n_idle : (kind == 1u ? n_inner : n_leaf);
            report.last_kind = kind; report.last_t = now;
        }
#endif
        if (n_idle > n_inner && n_idle > n_leaf) {
            if (want && !out_of_rays) {
                if (!first) g = __hip_atomic_fetch_add(&row(R_NEXT, 0), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                first = false;
                if (g >= total) out_of_rays = true;
                else {
                    uint32_t c = 0;
#pragma unroll
                    for (uint32_t step = 32; step; step >>= 1) {
                        const uint32_t c2 = c + step;
                        if (c2 < n_lanes && row(R_START, c2) <= g) c = c2;
                    }
                    uint32_t bits = row(R_NEAR, c) & 0xffffu;
                    for (uint32_t k = g - row(R_START, c); k; --k) bits &= bits - 1u;
                    const uint32_t i = (uint32_t)__builtin_ctz(bits);
                    owner = c | (i << 8);
                    const f3 Pc = mk3(__uint_as_float(row(R_PX, c)), __uint_as_float(row(R_PY, c)), __uint_as_float(row(R_PZ, c)));
                    ro = mk3(__uint_as_float(row(R_OX, c)), __uint_as_float(row(R_OY, c)), __uint_as_float(row(R_OZ, c)));
                    Rng r0; r0.key = row(R_KEY, c); r0.kg = row(R_K, c);
                    rd = norm3_unit(norm3_len(light_sample(r0, i, p0, p1, p2) - Pc, rdist));
                    id = bvh_inv_dir(rd);
                    want = !next_mesh(0); /* a ray that comes near no mesh after all is answered: not hit */
                }
            }
        } else if (n_inner >= n_leaf) {
            if (!want && cur >= 0) { /* a wide node */
#ifdef RT_WALK_STATS
                const unsigned long long w0 = __builtin_readcyclecounter();
#endif
                const DBvh4Q n = ldc(&sc.bvh4q[cur]);
#ifdef RT_WALK_STATS
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                report.wait_cycles += __builtin_readcyclecounter() - w0; /* of the lane that reports: issue to arrival of a node */
                ++report.waits;
#endif
                cnt.bvh_nodes += 1;
                int e0, e1, e2, e3;
                const int hits = bvh4_order(n, ro, id, rdist, e0, e1, e2, e3);
                if (hits > 3) push(e3);
                if (hits > 2) push(e2);
                if (hits > 1) push(e1);
                if (hits > 0) cur = e0;
                else if (sp > 0) cur = pop();
                else want = !next_mesh(mi + 1u);
            }
        } else if (!want && cur < 0) { /* a leaf */
            const int code = ~cur, first_tri = code >> 3, count = code & 7;
            bool hit_now = false;
            cnt.bvh_tris += (uint32_t)count;
            /* two records per round, both requested before the first is used, and tri_t's arithmetic without its early
             * returns (the same operations on the same values; a test that tri_t leaves early fails here in its condition) */
            auto hits = [&](const DTri& tr) {
                const f3 e1 = ld3(tr.e1), e2 = ld3(tr.e2);
                const f3 h = cross3(rd, e2);
                const float a = dot3(e1, h);
                const float f = 1.0f / a;
                const f3 sv = ro - ld3(tr.v0);
                const float u = f * dot3(sv, h);
                const f3 q = cross3(sv, e1);
                const float v = f * dot3(rd, q);
                const float t = f * dot3(e2, q);
                return !(__builtin_fabsf(a) < 1e-6f) && !(u < 0.0f || u > 1.0f) && !(v < 0.0f || u + v > 1.0f) && t > 1e-6f && t < rdist;
            };
            for (int k = 0; k < count; k += 2) {
                const DTri ta = ldc(&sc.tris[first_tri + k]);
                const DTri tb = ldc(&sc.tris[first_tri + (k + 1 < count ? k + 1 : k)]);
                const bool ha = hits(ta), hb = hits(tb);
                hit_now = hit_now || ha || hb;
            }
            if (hit_now) { __hip_atomic_fetch_or(&row(R_NEAR, owner & 0xffu), 1u << (16u + (owner >> 8)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); want = true; }
            else if (sp > 0) cur = pop();
            else want = !next_mesh(mi + 1u);
        }
    }
    wave_lds_sync();
    const uint32_t hit = row(R_NEAR, my) >> 16;
    wave_lds_sync(); /* every lane has its answer before the words are stack entries again */
    return hit;
}

/* powf(x, y) for x >= 0 through double: x^16 (the Cornell shininess) by four exact-enough double
 * squarings, anything else through the double pow; rounded once to float. */
__device__ __attribute__((noinline)) float pow_general(float x, float y) { return (float)pow((double)x, (double)y); }
RT_DEV float pow_shiny(float x, float y) {
#ifdef RT_EXP_FLOAT_POW /* attribution: the double squarings */
    if (y == 16.0f) { float a = x; a *= a; a *= a; a *= a; a *= a; return a; }
#endif
    if (y == 16.0f) { double a = (double)x; a *= a; a *= a; a *= a; a *= a; return (float)a; }
    if (y == 0.0f) return 1.0f;
    return pow_general(x, y); /* kept out of line: the double pow is large and rarely runs */
}

/* Scene::computeLightContribution, R/src/Scene.h:304-316 */
/* all16 (the same in every lane): every lane's material has shininess 16, so the exponent needs no per-lane branches.
 * The light term feeds nothing but the colour -- no hit / miss decision, no branch of the path, no draw of the random
 * stream depends on it -- so it is the one place where the arithmetic is not the reference's operation for operation: fused
 * multiply-adds, and the normalisation of the mirrored direction through v_rsq_f32 with one Newton step instead of a square
 * root and three divisions.  Each of its results lies within a few units in the last place of the reference's (whose own
 * unfused arithmetic is that far from the exact value): the colours agree to 1e-6 where they agreed to 1e-7, against a
 * tolerance of 40 dB.  RT_EXACT_LIGHT_TERM builds the reference's sequence instead (19.2 -> 18.0 ms on the Cornell frame
 * without it). */
#ifdef RT_EXACT_LIGHT_TERM
RT_DEV void light_term(f3& sum, const DLight& lt, const DMaterial& m, f3 N, f3 V, f3 L, bool all16 = false) {
    f3 R = norm3_unit((2.f * dot3(N, L)) * N - L);
    float nl = max_std(0.f, dot3(L, N));
    float sp;
    if (all16) sp = pow_shiny(max_std(0.f, dot3(R, V)), 16.0f);
    else sp = pow_shiny(max_std(0.f, dot3(R, V)), m.shininess);
    sum.x += lt.color[0] * m.kd[0] * nl; sum.x += lt.color[0] * m.ks[0] * sp;
    sum.y += lt.color[1] * m.kd[1] * nl; sum.y += lt.color[1] * m.ks[1] * sp;
    sum.z += lt.color[2] * m.kd[2] * nl; sum.z += lt.color[2] * m.ks[2] * sp;
}
#else
RT_DEV float fdot3(f3 a, f3 b) { return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x)); }
RT_DEV void light_term(f3& sum, const DLight& lt, const DMaterial& m, f3 N, f3 V, f3 L, bool all16 = false) {
    const float nl0 = fdot3(N, L), two = 2.f * nl0;
    const f3 R = mk3(__builtin_fmaf(two, N.x, -L.x), __builtin_fmaf(two, N.y, -L.y), __builtin_fmaf(two, N.z, -L.z));
    const float rr = fdot3(R, R);
    float inv = __builtin_amdgcn_rsqf(rr);
    inv = inv * __builtin_fmaf(-0.5f * rr, inv * inv, 1.5f); /* one Newton step: 1 / |R| to the last bit or two */
    const float rv = max_std(0.f, fdot3(R, V) * inv);
    const float nl = max_std(0.f, nl0);
    float sp;
    if (all16) sp = pow_shiny(rv, 16.0f);
    else sp = pow_shiny(rv, m.shininess);
    sum.x = __builtin_fmaf(lt.color[0] * m.kd[0], nl, sum.x); sum.x = __builtin_fmaf(lt.color[0] * m.ks[0], sp, sum.x);
    sum.y = __builtin_fmaf(lt.color[1] * m.kd[1], nl, sum.y); sum.y = __builtin_fmaf(lt.color[1] * m.ks[1], sp, sum.y);
    sum.z = __builtin_fmaf(lt.color[2] * m.kd[2], nl, sum.z); sum.z = __builtin_fmaf(lt.color[2] * m.ks[2], sp, sum.z);
}
#endif

RT_DEV int material_index(const DeviceScene& sc, int type, int obj) {
    return type == RT_SPHERE ? ldc(&sc.spheres[obj].mat) : (type == RT_SQUARE ? ldc(&sc.squares[obj].mat) : ldc(&sc.meshes[obj].mat));
}

#ifdef RT_SHADE_HIST /* debug build only: how many lanes of a wave shade a node together */
__device__ unsigned long long rt_dbg_hist[65];
#endif
/* ---- Scene::phong, R/src/Scene.h:319-422 (+ samplePointOnLight :252-276) ------------------ */
template <int M, int SB = RT_SB>
RT_DEV f3 shade(const DeviceScene& sc, const FullHit& h, Rng& rng, Cnt& cnt) {
    static_assert(16 % SB == 0, "the 16 light samples of a quad light come in whole batches");
    RT_LANE_PROBE(0);
    ++cnt.nodes;
#ifdef RT_SHADE_HIST
    { /* the predicate depends on per-lane data so that the ballot cannot be moved out of the (divergent) path loop */
        const uint64_t act = __ballot(cnt.nodes != 0u);
        if (wave_rank(act, __lane_id()) == 0u) atomicAdd(&rt_dbg_hist[__popcll(act)], 1ull);
    }
#endif
    const DMaterial m = ldc(&sc.materials[material_index(sc, h.type, h.obj)]);
    f3 ambient = 0.35f * mk3(0.2f, 0.2f, 0.2f);
    f3 diffuse = mk3(0.f, 0.f, 0.f), specular = mk3(0.f, 0.f, 0.f);
    const f3 P = h.p;
    const f3 N = norm3_unit(h.n);
    const f3 V = norm3(-1.f * P);              /* camera assumed at the world origin */
    const f3 so = P + 0.001f * N;              /* shadow-ray origin */
    const bool all16 = __ballot(m.shininess != 16.0f) == 0ull; /* decided once per shaded point for the whole wave */
    for (uint32_t li = 0; li < sc.n_lights; ++li) {
        const DLight lt = ldc(&sc.lights[li]);
        if (lt.type == RT_LIGHT_QUAD && !(M & RT_M_KD)) {
            f3 sum = mk3(0.f, 0.f, 0.f);
            f3 p0 = ld3(lt.p0), p1 = ld3(lt.p1), p2 = ld3(lt.p2);
            {
                const bool so_inside = inside_box(sc.box_lo, sc.box_hi, so);
                const Rng at_first = rng; /* the stream position of light sample 0 */
                /* What this point's shadow rays still have to test of every square (SquareRules, DLightSq).  The side of a
                 * square's plane the origin lies on must be the same, by the margin a_sign, in every lane of the wave; then a
                 * square with all three corners of the light on that side beyond t_miss is missed by every sample -- walls seen
                 * from inside the box -- and for the others the samples decide ray by ray.  NaN records (point lights) and
                 * origins outside the root box decide nothing: every ray then tests every square. */
                SquareRules rules;
                rules.test = 0xffffffffu; rules.by_sample = 0u; rules.flipped = 0u; rules.can_hit = 0u;
                rules.rec = sc.light_sq ? sc.light_sq + (size_t)li * sc.n_squares : nullptr;
                if (sc.n_squares <= 32u && sc.light_sq && __ballot(!so_inside) == 0ull) {
                    rules.test = 0u;
                    for_each_record(sc.squares, sc.n_squares, [&](const DSquare& q, uint32_t k) {
                        const DLightSq t = ldc(&rules.rec[k]);
                        const f3 vo = ld3(q.v0) - so;
                        const float num = dot3(vo, ld3(q.n));
                        const bool all_pos = __ballot(!(num > t.a_sign)) == 0ull, all_neg = __ballot(!(num < -t.a_sign)) == 0ull;
                        bool keep = true;
                        if (all_pos || all_neg) {
                            if (all_pos ? t.g_min > t.t_miss : t.g_max < -t.t_miss) keep = false;
                            else {
                                rules.by_sample |= 1u << k;
                                if (all_neg) rules.flipped |= 1u << k;
                                const bool in = __builtin_fabsf(-dot3(vo, ld3(q.w)) - t.cw) < t.ow && __builtin_fabsf(-dot3(vo, ld3(q.h)) - t.ck) < t.ok;
                                const bool beyond = all_pos ? t.g_min < -t.t_hit : t.g_max > t.t_hit; /* some sample can lie on the far side */
                                if (beyond && __ballot(!in) == 0ull) rules.can_hit |= 1u << k;
                            }
                        }
                        if (keep) rules.test |= 1u << k;
                    });
                }
#ifdef RT_EXP_SKIP_SQUARE /* attribution: the shadow tests of one square */
                rules.test &= ~(1u << RT_EXP_SKIP_SQUARE);
#endif
                float sph_c[RT_SPH_KEPT];
#pragma unroll
                for (uint32_t k = 0; k < RT_SPH_KEPT; ++k) sph_c[k] = k < sc.n_spheres ? sphere_c(ldc(&sc.spheres[k]), so) : 0.f;
                uint32_t dark = 0, near = 0; /* bit i: light sample i is occluded / still has to be walked through a BVH */
                /* With a BVH in the scene a batch whose rays are all decided (no lane of the wave has one that comes near a
                 * mesh: the usual case away from the meshes) adds its light terms right away, like the variants without a
                 * BVH.  From the first batch that leaves a ray open the terms wait -- the sum runs in sample order -- and
                 * are added after the walks, their light directions rebuilt from the random stream. */
                uint32_t first_open = 16; /* wave-uniform: the first sample whose light term is still to be added */
                for (int i0 = 0; i0 < 16; i0 += SB) {
                    RT_LANE_PROBE(1);
                    f3 L[SB], Ld[SB];
                    float dist[SB], bu[SB], bv[SB];
                    bool occ[SB];
                    uint32_t near4;
#pragma unroll
                    for (int j = 0; j < SB; ++j) {
                        float u = rng.unit();
                        float v = rng.unit();
                        if (u + v > 1.0f) { u = 1.0f - u; v = 1.0f - v; }
                        bu[j] = u; bv[j] = v;
                    }
#if defined(RT_EXP_COUNT_OPEN) /* attribution: how often a square with sample rules still needs its own arithmetic (counts land in rt_stats) */
                    const uint32_t sq_test = sample_rules<SB>(rules, bu, bv, occ);
                    cnt.bvh_tris += (sq_test >> RT_EXP_COUNT_OPEN) & 1u; cnt.bvh_nodes += (rules.by_sample >> RT_EXP_COUNT_OPEN) & 1u; cnt.nodes += 1000u * ((rules.can_hit >> RT_EXP_COUNT_OPEN) & 1u);
#elif defined(RT_EXP_DROP_OPEN) /* attribution: the cost of the rules alone (squares they leave open are not tested) */
                    const uint32_t sq_test = sample_rules<SB>(rules, bu, bv, occ) & ~rules.by_sample;
#else
                    const uint32_t sq_test = sample_rules<SB>(rules, bu, bv, occ);
#endif
#pragma unroll
                    for (int j = 0; j < SB; ++j) {
                        const float u = bu[j], v = bv[j];
                        f3 S = ((1.f - u - v) * p0 + u * p1) + v * p2;
                        f3 SP = S - P;
                        L[j] = norm3_len(SP, dist[j]); /* dist = |S - P|, the same sqrt the normalisation takes */
#ifdef RT_EXP_NO_RENORM /* attribution: the second normalisation of every shadow ray */
                        Ld[j] = L[j];
#else
                        Ld[j] = norm3_unit(L[j]); /* Ray() normalises L again */
#endif
                    }
#ifdef RT_EXP_NO_SHADOW_TESTS /* attribution: light samples, both normalisations and light terms without the object tests */
                    near4 = 0;
#pragma unroll
                    for (int j = 0; j < SB; ++j) occ[j] = Ld[j].x + dist[j] > 1e30f;
#else
                    shadow_batch<M, SB>(sc, so, so_inside, Ld, dist, occ, near4, cnt, sq_test, sph_c);
#endif
                    bool now = true; /* every ray of the batch is decided: add the light terms in sample order right away */
                    if (M & RT_M_BVH) {
                        if (first_open == 16 && __ballot(near4 != 0u) != 0ull) first_open = (uint32_t)i0;
                        now = first_open == 16;
#pragma unroll
                        for (int j = 0; j < SB; ++j) dark |= (occ[j] ? 1u : 0u) << (i0 + j);
                        near |= near4 << i0;
                    }
#ifdef RT_EXP_NO_LIGHT_TERM /* attribution: the light terms of the unoccluded samples */
                    now = false;
#pragma unroll
                    for (int j = 0; j < SB; ++j) sum.x += occ[j] ? 0.f : L[j].x;
#endif
                    if (now) {
#pragma unroll
                        for (int j = 0; j < SB; ++j)
                            if (!occ[j]) { RT_LANE_PROBE(8); light_term(sum, lt, m, N, V, L[j], all16); }
                    }
                }
                if ((M & RT_M_BVH) && first_open < 16) {
                    /* the undecided rays of all samples in one go, then the rest of the sum in sample order; a light direction
                     * is cheaper to rebuild from the random stream than to keep over the traversals */
#ifndef RT_EXP_NO_WALK /* attribution builds (see RT_EXP_NO_MESH_SHADOW): the rays near a mesh count as unoccluded */
                    if (__ballot(near != 0)) dark |= walk_deferred<M>(sc, P, so, at_first, p0, p1, p2, near, cnt);
#endif
                    for (uint32_t i = first_open; i < 16; ++i) {
                        if ((dark >> i) & 1u) continue;
                        float len;
                        light_term(sum, lt, m, N, V, norm3_len(light_sample(at_first, i, p0, p1, p2) - P, len));
                    }
                }
            }
            sum = sum / 16.f;
            diffuse = diffuse + sum;
            specular = specular + sum; /* the reference adds the quad-light term to both sums */
        } else if (!(M & RT_M_KD)) { /* a point light */
            f3 SP = ld3(lt.pos) - P;
            f3 L = norm3(SP);
            if (occluded<M>(sc, so, norm3_unit(L), len3(SP), cnt)) continue;
            light_term(diffuse, lt, m, N, V, L);
        } else {
            /* kd-tree with internal nodes: one shadow query at a time, through one place in the code (the traversal is large):
             * a point light (its term goes straight into the diffuse sum), or the 16 samples of a quad light (sum / 16 into both). */
            const bool quad = lt.type == RT_LIGHT_QUAD;
            const f3 p0 = ld3(lt.p0), p1 = ld3(lt.p1), p2 = ld3(lt.p2);
            f3 acc = quad ? mk3(0.f, 0.f, 0.f) : diffuse;
            for (int i = 0; i < (quad ? 16 : 1); ++i) {
                f3 S = ld3(lt.pos);
                if (quad) {
                    float u = rng.unit();
                    float v = rng.unit();
                    if (u + v > 1.0f) { u = 1.0f - u; v = 1.0f - v; }
                    S = ((1.f - u - v) * p0 + u * p1) + v * p2;
                }
                f3 SP = S - P;
                f3 L = norm3(SP);
                if (occluded<M>(sc, so, norm3_unit(L), len3(SP), cnt)) continue; /* Ray() normalises L again */
                light_term(acc, lt, m, N, V, L);
            }
            if (quad) {
                acc = acc / 16.f;
                diffuse = diffuse + acc;
                specular = specular + acc; /* the reference adds the quad-light term to both sums */
            } else diffuse = acc;
        }
    }
    return (ambient + diffuse) + specular;
}

/* Scene::phong for a node whose colour the reference never uses: a glass sphere hit with bounces left
 * (R/src/Scene.h:626 computes it, :705 overwrites it with F * reflected + (1 - F) * refracted).  Only what the
 * rest of the sample can observe is kept: the position in the random stream (32 draws per quad light,
 * R/src/Scene.h:369-370 via :259-260; none for a point light) and the work counters, which keep counting what
 * the reference does -- the shadow queries not traced are reported separately in `skipped`. */
RT_DEV void skip_shade(const DeviceScene& sc, Rng& rng, Cnt& cnt) {
    RT_LANE_PROBE(11);
    ++cnt.nodes;
    for (uint32_t li = 0; li < sc.n_lights; ++li) {
        const uint32_t n = ldc(&sc.lights[li].type) == RT_LIGHT_QUAD ? 16u : 1u;
        if (n == 16u) rng.skip(32u);
        cnt.rays += n;
        cnt.skipped += n;
    }
}

/* The rays a hit on a mirror or glass sphere sends on, R/src/Scene.h:641-702.  kind 0: mirror, (ro, rd) the reflected ray;
 * kind 1: glass, reflected ray (ro, rd) and refracted ray (to, td), F the Fresnel weight of the reflected colour;
 * kind 2: glass without a refracted ray (total internal reflection, or F rounds to 1).  mtype is RT_MAT_MIRROR or RT_MAT_GLASS. */
struct Bounce { f3 ro, rd, to, td; float F; uint32_t kind; };
RT_DEV Bounce bounce_rays(const DeviceScene& sc, int mtype, int mat, f3 P, f3 N, f3 I) {
    RT_LANE_PROBE(10);
    Bounce b;
    b.to = b.td = mk3(0.f, 0.f, 0.f);
    b.F = 0.f;
    if (mtype == RT_MAT_MIRROR) {
        f3 R = I - (2.0f * dot3(I, N)) * N;
        b.ro = P + 0.001f * N; b.rd = norm3_unit(R);
        b.kind = 0u;
        return b;
    }
    float n1 = 1.0f, n2 = ldc(&sc.materials[mat].ior);
    if (dot3(I, N) > 0.0f) { float s = n1; n1 = n2; n2 = s; N = -1.0f * N; }
    float cosI = __builtin_fabsf(dot3(I, N));
    float eta = n1 / n2;
    float sinT2 = eta * eta * (1.0f - cosI * cosI);
    float F;
    if (sinT2 > 1.0f) F = 1.0f;
    else {
        float cosT = __builtin_sqrtf(1.0f - sinT2);
        float rs = (n1 * cosI - n2 * cosT) / (n1 * cosI + n2 * cosT);
        float rp = (n1 * cosT - n2 * cosI) / (n1 * cosT + n2 * cosI);
        /* pow(x, 2) in double = exact square rounded once = the float product */
        F = (rs * rs + rp * rp) / 2.0f;
    }
    f3 R = I - (2.0f * dot3(I, N)) * N;
    b.F = F;
    if (F < 1.0f) {
        f3 T = eta * (I - cosI * N) - __builtin_sqrtf(1.0f - sinT2) * N;
        b.to = P - 0.001f * N; b.td = norm3_unit(T);
        b.kind = 1u;
    } else b.kind = 2u;
    b.ro = P + 0.001f * N; b.rd = norm3_unit(R);
    return b;
}

/* ---- Scene::rayTraceRecursive, R/src/Scene.h:596-713, as a loop --------------------------
 * The recursion (depth <= 11, reflected child before refracted child) becomes a loop over
 * "calls" with a small per-lane frame stack; the colour arithmetic on the way back up is the
 * reference's, in the reference's order, so the result does not depend on the rewrite.
 * A frame is RT_FRAME_WORDS words; what kind it is lives in two bits per level of one register:
 *   kind 0 mirror : w0..2 = phong colour of the node, w3 = material index (specular_material is read again)
 *   kind 1 glass, reflected child running : w0..2 = refracted origin, w3..5 = refracted direction, w6 = F
 *   kind 2 glass, reflected child running, no refracted ray (total internal reflection): w6 = F
 *   kind 3 glass, refracted child running : w0..2 = reflected colour, w6 = F
 * The first L levels of the stack live in LDS (word w of level l of this lane at cnt.frames[(l * RT_FRAME_WORDS
 * + w) * cnt.frame_stride]: lane-interleaved, conflict-free), deeper ones in a per-lane array.  A per-lane array
 * indexed by the stack pointer is scratch memory: every push and pop of it is a round trip through the memory
 * hierarchy, 64 lanes wide, and most path trees are shallow.
 * PROBE = true walks the same tree without shading it (no light samples, no shadow rays) and only
 * counts its nodes in cnt.nodes, up to RT_PROBE_CAP: the tree's shape does not depend on the shading. */
#define RT_MAX_BOUNCES 10
#define RT_PROBE_CAP 24
#define RT_FRAME_WORDS 7
template <int L>
struct FrameStack {
    rt_lds_f32* lds; uint32_t stride; /* the LDS levels, typed as LDS (see lds_words) */
    float deep[(RT_MAX_BOUNCES - L > 0 ? RT_MAX_BOUNCES - L : 1) * RT_FRAME_WORDS];
    /* N words of level l, from / into v[first .. first + N) */
    template <int FIRST, int N>
    RT_DEV void put(int l, const float (&v)[RT_FRAME_WORDS]) {
        if (L >= RT_MAX_BOUNCES || l < L) {
#pragma unroll
            for (int w = FIRST; w < FIRST + N; ++w) lds[(uint32_t)(l * RT_FRAME_WORDS + w) * stride] = v[w];
        } else {
#pragma unroll
            for (int w = FIRST; w < FIRST + N; ++w) deep[(l - L) * RT_FRAME_WORDS + w] = v[w];
        }
    }
    template <int FIRST, int N>
    RT_DEV void get(int l, float (&v)[RT_FRAME_WORDS]) const {
        if (L >= RT_MAX_BOUNCES || l < L) {
#pragma unroll
            for (int w = FIRST; w < FIRST + N; ++w) v[w] = lds[(uint32_t)(l * RT_FRAME_WORDS + w) * stride];
        } else {
#pragma unroll
            for (int w = FIRST; w < FIRST + N; ++w) v[w] = deep[(l - L) * RT_FRAME_WORDS + w];
        }
    }
};
template <int M, bool PROBE = false, int L = 0, int SB = RT_SB>
RT_DEV f3 trace_path(const DeviceScene& sc, f3 o, f3 d, Rng& rng, Cnt& cnt) {
    FrameStack<L> stk;
    stk.lds = (rt_lds_f32*)cnt.frames; stk.stride = cnt.frame_stride;
    int sp = 0;
    uint32_t kinds = 0; /* bits 2 l, 2 l + 1: kind of the frame at level l */
    f3 ret = mk3(0.f, 0.f, 0.f);
    for (;;) {
#ifdef RT_SHADE_HIST
        if (!PROBE) { const uint64_t act = __ballot(sp >= 0); if (wave_rank(act, __lane_id()) == 0u) atomicAdd(&rt_dbg_hist[0], 1ull); }
#endif
        HitRec rec = closest_rec<M>(sc, o, d, cnt);
        bool descended = false;
        if (rec.type < 0) ret = mk3(0.f, 0.f, 0.f);
        else {
            FullHit h = finish_hit(sc, rec, o, d);
            f3 col = mk3(0.f, 0.f, 0.f);
            const bool bounces = sp < RT_MAX_BOUNCES && h.type == RT_SPHERE;
            int mat = 0, mtype = RT_MAT_DIFFUSE;
            if (bounces) { mat = ldc(&sc.spheres[h.obj].mat); mtype = ldc(&sc.materials[mat].type); }
            if (PROBE) { if (++cnt.nodes >= RT_PROBE_CAP) return ret; }
#ifdef RT_EXP_NO_SHADE /* attribution: everything but Scene::phong */
            else if (true) { skip_shade(sc, rng, cnt); col = mk3(0.07f, 0.07f, 0.07f); }
#endif
            else if (mtype == RT_MAT_GLASS) skip_shade(sc, rng, cnt);
            else col = shade<M, SB>(sc, h, rng, cnt);
            ret = col;
            if (bounces && mtype != RT_MAT_DIFFUSE) {
                const Bounce bn = bounce_rays(sc, mtype, mat, h.p, h.n, d);
                float f[RT_FRAME_WORDS];
                if (bn.kind == 0u) {
                    f[0] = col.x; f[1] = col.y; f[2] = col.z; f[3] = __int_as_float(mat);
                    stk.template put<0, 4>(sp, f);
                } else if (bn.kind == 1u) {
                    f[0] = bn.to.x; f[1] = bn.to.y; f[2] = bn.to.z; f[3] = bn.td.x; f[4] = bn.td.y; f[5] = bn.td.z; f[6] = bn.F;
                    stk.template put<0, 7>(sp, f);
                } else {
                    f[6] = bn.F;
                    stk.template put<6, 1>(sp, f);
                }
                kinds = (kinds & ~(3u << (2 * sp))) | (bn.kind << (2 * sp));
                ++sp;
                o = bn.ro; d = bn.rd;
                descended = true;
            }
        }
        if (descended) continue;
        bool resumed = false;
        while (sp > 0) {
            const int l = sp - 1;
            const uint32_t kind = (kinds >> (2 * l)) & 3u;
            float f[RT_FRAME_WORDS];
            if (kind == 0u) {
                stk.template get<0, 4>(l, f);
                const f3 ks = ld3c(sc.materials[__float_as_int(f[3])].ks);
                ret = mk3(f[0] + ret.x * ks.x, f[1] + ret.y * ks.y, f[2] + ret.z * ks.z);
                --sp;
            } else if (kind == 1u) {
                stk.template get<0, 6>(l, f);
                o = mk3(f[0], f[1], f[2]); d = mk3(f[3], f[4], f[5]);
                f[0] = ret.x; f[1] = ret.y; f[2] = ret.z;
                stk.template put<0, 3>(l, f);
                kinds |= 3u << (2 * l);
                resumed = true;
                break;
            } else if (kind == 2u) {
                stk.template get<6, 1>(l, f);
                ret = f[6] * ret + (1.0f - f[6]) * mk3(0.f, 0.f, 0.f);
                --sp;
            } else {
                stk.template get<0, 3>(l, f);
                stk.template get<6, 1>(l, f);
                ret = f[6] * mk3(f[0], f[1], f[2]) + (1.0f - f[6]) * ret;
                --sp;
            }
        }
        if (!resumed) return ret;
    }
}

/* ---- screen_space_to_world_space_ray, R/src/matrixUtilities.h:198-222 --------------------
 * double arithmetic exactly as the reference; the direction comes back normalised once. */
RT_DEV void primary_ray(const DCamera& cam, float u, float v, f3& o, f3& d) {
    double x = (double)2.f * (double)u - (double)1.f;
    double y = -((double)2.f * (double)v - (double)1.f);
    double z = cam.depth_near, w = 1.0;
    const double* A = cam.proj_inv;
    double a0 = A[0] * x + A[4] * y + A[8] * z + A[12] * w;
    double a1 = A[1] * x + A[5] * y + A[9] * z + A[13] * w;
    double a2 = A[2] * x + A[6] * y + A[10] * z + A[14] * w;
    double a3 = A[3] * x + A[7] * y + A[11] * z + A[15] * w;
    const double* B = cam.mv_inv;
    double b0 = B[0] * a0 + B[4] * a1 + B[8] * a2 + B[12] * a3;
    double b1 = B[1] * a0 + B[5] * a1 + B[9] * a2 + B[13] * a3;
    double b2 = B[2] * a0 + B[6] * a1 + B[10] * a2 + B[14] * a3;
    double b3 = B[3] * a0 + B[7] * a1 + B[11] * a2 + B[15] * a3;
    f3 world = mk3((float)(b0 / b3), (float)(b1 / b3), (float)(b2 / b3));
    o = ld3(cam.eye);
    d = norm3(world - o);
}

#endif /* RT_DEVICE_H */
