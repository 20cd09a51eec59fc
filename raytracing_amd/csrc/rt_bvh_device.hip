/*
 * rt_bvh_device.hip -- per-mesh triangle BVH built on the GPU (gfx950).
 *
 * The reference has no such structure: Mesh::intersect loops over every triangle (R/src/Mesh.h:225).  A
 * BVH only has to visit every triangle that loop would accept; which tree is used does not change a hit
 * record (ties between equal t go to the lower original index, kept in tri_orig).  The host builder
 * (rt_build.cpp, binned SAH) makes the better tree but needs ~0.1 us per triangle per level on one core;
 * this one orders the triangles along a 63-bit Morton curve and cuts the sorted sequence top-down, all of
 * it in device memory:
 *
 *   k_tri_keys      one thread per triangle: box centre -> 21 bits per axis (cubic cells) -> Morton key
 *   rocprim sort    (key, triangle) pairs
 *   k_permute       triangle records into sorted order (= leaf order), tri_orig
 *   k_split         one launch per tree level, one thread per node of that level: the node's triangles are
 *                   a range of the sorted sequence; cut it where the highest differing key bit flips
 *                   (binary search), or at the median when the keys are equal or when
 *                   depth + 1 + ceil(log2 count) would exceed the stack budget -- the host builder's rule,
 *                   which keeps every root-to-leaf path within the 32 entries of the traversal stack.
 *                   Ranges of <= 4 triangles become leaves; the others are queued for the next level.
 *   rocprim scan    a node is named by its cut position (unique in a tree over a sequence): flags -> ranks
 *                   give a numbering that does not depend on the order the queue was filled in
 *   k_fit           levels in reverse: child boxes from the leaf triangles or from the finished child node,
 *                   padded exactly as the host builder pads them
 */
#include "rt_bvh_device.h"

#include <algorithm>
#include <cstring>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

namespace rtd {
namespace {

const uint32_t kLeafMax = 4;

struct Item {          /* one inner node while the tree is being cut */
    uint32_t lo, hi;   /* its triangles: positions [lo, hi) of the sorted sequence */
    uint32_t mid;      /* cut position = the node's name */
    uint32_t kid[2];   /* queue positions of the children that are inner nodes */
};

__device__ inline uint64_t spread3(uint32_t v) { /* 21 bits -> every third bit */
    uint64_t x = v & 0x1fffffu;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

struct F3 { float x, y, z; };

__device__ inline void tri_box(const DTri& t, float lo[3], float hi[3]) {
    for (int a = 0; a < 3; ++a) {
        float p0 = t.v0[a], p1 = t.v0[a] + t.e1[a], p2 = t.v0[a] + t.e2[a];
        lo[a] = fminf(fminf(lo[a], p0), fminf(p1, p2));
        hi[a] = fmaxf(fmaxf(hi[a], p0), fmaxf(p1, p2));
    }
}

__global__ void __launch_bounds__(256) k_tri_keys(const DTri* tris, uint32_t count, F3 lo, float cells_per_unit, uint64_t* keys,
                                                  uint32_t* ids) {
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= count) return;
    float blo[3] = {INFINITY, INFINITY, INFINITY}, bhi[3] = {-INFINITY, -INFINITY, -INFINITY};
    tri_box(tris[i], blo, bhi);
    const float org[3] = {lo.x, lo.y, lo.z};
    uint32_t q[3];
    for (int a = 0; a < 3; ++a) {
        float c = 0.5f * (blo[a] + bhi[a]);
        float f = (c - org[a]) * cells_per_unit;
        q[a] = (uint32_t)fminf(fmaxf(f, 0.f), 2097151.f); /* NaN -> 0 */
    }
    keys[i] = spread3(q[0]) << 2 | spread3(q[1]) << 1 | spread3(q[2]);
    ids[i] = i;
}

__global__ void __launch_bounds__(256) k_permute(const DTri* src, const uint32_t* ids, uint32_t count, DTri* dst, uint32_t* tri_orig) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    __shared__ float4 s_rec[768];
    if (i < count) {
        const uint32_t id = ids[i];
        const float4* s = (const float4*)(src + id);
        s_rec[3 * threadIdx.x] = s[0]; s_rec[3 * threadIdx.x + 1] = s[1]; s_rec[3 * threadIdx.x + 2] = s[2];
        tri_orig[i] = id;
    }
    __syncthreads();
    float4* out4 = (float4*)dst + 3ull * blockIdx.x * 256u;
    const uint64_t left = 3ull * count - 3ull * blockIdx.x * 256u;
#pragma unroll
    for (uint32_t k = 0; k < 3; ++k) {
        const uint32_t j = threadIdx.x + 256u * k;
        if (j < left) out4[j] = s_rec[j];
    }
}

/* items [level[depth], level[depth + 1]) are the nodes of this level; *n_items is the end of the queue.  k_level_end
 * notes where the next level ends once a level is cut, so the host launches all levels without waiting for any. */
__global__ void k_level_end(uint32_t* level, int depth, const uint32_t* n_items, uint32_t capacity) {
    const uint32_t n = *n_items;
    level[depth + 2] = n < capacity ? n : capacity;
}

__global__ void __launch_bounds__(256) k_split(Item* items, const uint32_t* level, int depth, int budget, const uint64_t* keys,
                                               uint32_t* flags, uint32_t* n_items, uint32_t capacity) {
    const uint32_t begin = level[depth], end = level[depth + 1]; /* the grid is sized for the most a level can hold */
    uint32_t i = begin + blockIdx.x * 256u + threadIdx.x;
    if (i >= end) return;
    Item it = items[i];
    const uint32_t cnt = it.hi - it.lo;
    const uint64_t first = keys[it.lo], last = keys[it.hi - 1];
    int log2c = 32 - __clz((int)(cnt - 1)); /* ceil(log2 cnt), cnt > 4 */
    uint32_t mid = it.lo + cnt / 2;
    if (first != last && depth + 1 + log2c <= budget) {
        const int bit = 63 - __clzll((long long)(first ^ last));
        const uint64_t pivot = (last >> bit) << bit; /* the common prefix, this bit set, zeros below */
        uint32_t a = it.lo + 1, b = it.hi - 1;      /* smallest position with key >= pivot lies in [a, b] */
        while (a < b) {
            uint32_t m = a + (b - a) / 2;
            if (keys[m] >= pivot) b = m; else a = m + 1;
        }
        mid = a;
    }
    it.mid = mid;
    it.kid[0] = it.kid[1] = 0xffffffffu;
    flags[mid] = 1u;
    const uint32_t lo[2] = {it.lo, mid}, hi[2] = {mid, it.hi};
    for (int k = 0; k < 2; ++k) {
        if (hi[k] - lo[k] <= kLeafMax) continue;
        uint32_t at = atomicAdd(n_items, 1u);
        if (at < capacity) { /* cannot overflow: a sequence of n has at most n - 1 cuts */
            Item ch; ch.lo = lo[k]; ch.hi = hi[k]; ch.mid = 0; ch.kid[0] = ch.kid[1] = 0xffffffffu;
            items[at] = ch;
            it.kid[k] = at;
        }
    }
    items[i] = it;
}

__global__ void __launch_bounds__(256) k_fit(const Item* items, uint32_t begin, uint32_t end, const uint32_t* rank, const DTri* tris,
                                             uint32_t first, int32_t node_base, float pad, DBvhNode* nodes) {
    uint32_t i = begin + blockIdx.x * 256u + threadIdx.x;
    if (i >= end) return;
    const Item it = items[i];
    const uint32_t lo[2] = {it.lo, it.mid}, hi[2] = {it.mid, it.hi};
    DBvhNode out;
    float* olo[2] = {out.lo0, out.lo1};
    float* ohi[2] = {out.hi0, out.hi1};
    int32_t c[2], n[2];
    for (int k = 0; k < 2; ++k) {
        if (it.kid[k] == 0xffffffffu) { /* leaf: box of its triangles, then the pad */
            float blo[3] = {INFINITY, INFINITY, INFINITY}, bhi[3] = {-INFINITY, -INFINITY, -INFINITY};
            for (uint32_t t = lo[k]; t < hi[k]; ++t) tri_box(tris[t], blo, bhi);
            for (int a = 0; a < 3; ++a) { olo[k][a] = blo[a] - pad; ohi[k][a] = bhi[a] + pad; }
            c[k] = (int32_t)(first + lo[k]);
            n[k] = (int32_t)(hi[k] - lo[k]);
        } else { /* inner child, finished by the previous launch: union of its (padded) child boxes */
            const uint32_t r = rank[items[it.kid[k]].mid];
            const DBvhNode ch = nodes[r];
            for (int a = 0; a < 3; ++a) { olo[k][a] = fminf(ch.lo0[a], ch.lo1[a]); ohi[k][a] = fmaxf(ch.hi0[a], ch.hi1[a]); }
            c[k] = node_base + (int32_t)r;
            n[k] = 0;
        }
    }
    out.c0 = c[0]; out.n0 = n[0]; out.c1 = c[1]; out.n1 = n[1];
    nodes[rank[it.mid]] = out;
}

/* floats as unsigned integers in the same order (-inf lowest), for atomicMin / atomicMax */
__host__ __device__ inline uint32_t ordered(float f) {
    uint32_t b;
    memcpy(&b, &f, 4);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__host__ __device__ inline float unordered(uint32_t u) {
    uint32_t b = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    float f;
    memcpy(&f, &b, 4);
    return f;
}

__global__ void __launch_bounds__(256) k_make_tris(const float* positions, const uint32_t* triangles, uint32_t first_vertex, uint32_t n_vertices,
                                                   uint32_t n_triangles, DTri* tris, uint32_t* tri_orig, uint32_t* box) {
    __shared__ float s_box[4][6];
    __shared__ float4 s_rec[768];
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    /* a bounded grid walks the triangles in strides: the box then costs six atomics per workgroup of a few thousand,
     * not of forty thousand (atomics on one address are served one after the other) */
    for (uint64_t base = (uint64_t)blockIdx.x * 256u; base < n_triangles; base += (uint64_t)gridDim.x * 256u) {
        const uint64_t t = base + threadIdx.x;
        if (t < n_triangles) {
            const uint32_t i0 = triangles[3 * t], i1 = triangles[3 * t + 1], i2 = triangles[3 * t + 2];
            DTri tr;
            if (i0 >= n_vertices || i1 >= n_vertices || i2 >= n_vertices) {
                atomicOr(&box[6], 1u);
                memset(&tr, 0, sizeof(tr));
            } else {
                const float scaling = 1.000001f; /* R/src/Mesh.h:220 */
                const float* p0 = positions + 3 * (size_t)i0;
                const float* p1 = positions + 3 * (size_t)i1;
                const float* p2 = positions + 3 * (size_t)i2;
                for (int a = 0; a < 3; ++a) {
                    const float v0 = scaling * p0[a], v1 = scaling * p1[a], v2 = scaling * p2[a];
                    tr.v0[a] = v0; tr.e1[a] = v1 - v0; tr.e2[a] = v2 - v0;
                }
                tr.i0 = first_vertex + i0; tr.i1 = first_vertex + i1; tr.i2 = first_vertex + i2;
                tri_box(tr, lo, hi);
            }
            const float4* s = (const float4*)&tr;
            s_rec[3 * threadIdx.x] = s[0]; s_rec[3 * threadIdx.x + 1] = s[1]; s_rec[3 * threadIdx.x + 2] = s[2];
            tri_orig[t] = (uint32_t)t;
        }
        __syncthreads();
        { /* the 48-byte records leave through LDS so that neighbouring threads store neighbouring 16 bytes */
            float4* out4 = (float4*)tris + 3ull * base;
            const uint64_t left = 3ull * n_triangles - 3ull * base; /* float4s from this chunk's start to the end */
#pragma unroll
            for (uint32_t k = 0; k < 3; ++k) {
                const uint32_t i = threadIdx.x + 256u * k;
                if (i < left) out4[i] = s_rec[i];
            }
        }
        __syncthreads();
    }
    for (int a = 0; a < 3; ++a)
        for (int off = 32; off > 0; off >>= 1) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], off));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off));
        }
    if ((threadIdx.x & 63u) == 0)
        for (int a = 0; a < 3; ++a) { s_box[threadIdx.x >> 6][a] = lo[a]; s_box[threadIdx.x >> 6][3 + a] = hi[a]; }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int k = (int)threadIdx.x;
        float v = s_box[0][k];
        for (int w = 1; w < 4; ++w) v = k < 3 ? fminf(v, s_box[w][k]) : fmaxf(v, s_box[w][k]);
        if (k < 3) atomicMin(&box[k], ordered(v)); else atomicMax(&box[k], ordered(v));
    }
}

/* ---- binary tree -> four-wide tree ------------------------------------------------------------------------ */
__device__ inline bool child_inner(int32_t c, int32_t n) { return n == 0 && c >= 0; }
__device__ inline bool child_empty(int32_t c, int32_t n) { return n == 0 && c < 0; }

__global__ void __launch_bounds__(256) k_parents(const DBvhNode* nodes, uint32_t n2, int32_t* parent) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n2) return;
    const DBvhNode n = nodes[i];
    if (child_inner(n.c0, n.n0)) parent[n.c0] = (int32_t)i;
    if (child_inner(n.c1, n.n1)) parent[n.c1] = (int32_t)i;
}

/* keep[i] = 1 for the nodes at an even depth of their tree (the roots have no parent: depth 0) */
__global__ void __launch_bounds__(256) k_even_depth(const int32_t* parent, uint32_t n2, uint32_t* keep) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n2) return;
    uint32_t depth = 0;
    for (int32_t p = parent[i]; p >= 0 && depth < 64u; p = parent[p]) ++depth; /* trees are at most 32 deep */
    keep[i] = (depth & 1u) ? 0u : 1u;
}

__device__ inline void put_child(DBvh4Node& o, int slot, const float* lo, const float* hi, int32_t c, int32_t n) {
    DBvhNode& h = o.half[slot >> 1];
    float* dlo = (slot & 1) ? h.lo1 : h.lo0;
    float* dhi = (slot & 1) ? h.hi1 : h.hi0;
    for (int a = 0; a < 3; ++a) { dlo[a] = lo[a]; dhi[a] = hi[a]; }
    if (slot & 1) { h.c1 = c; h.n1 = n; } else { h.c0 = c; h.n0 = n; }
}

/* place: where each wide node (numbered by rank) goes in the table, or null for rank order */
__global__ void __launch_bounds__(256) k_collapse(const DBvhNode* nodes, uint32_t n2, const uint32_t* keep, const uint32_t* rank, const uint32_t* place, DBvh4Node* out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n2 || !keep[i]) return;
    const DBvhNode n = nodes[i];
    DBvh4Node o;
    const float none_lo[3] = {INFINITY, INFINITY, INFINITY}, none_hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int s = 0; s < 4; ++s) put_child(o, s, none_lo, none_hi, -1, 0);
    int slot = 0;
    const float* lo[2] = {n.lo0, n.lo1};
    const float* hi[2] = {n.hi0, n.hi1};
    const int32_t c[2] = {n.c0, n.c1}, cnt[2] = {n.n0, n.n1};
    for (int k = 0; k < 2; ++k) {
        if (child_empty(c[k], cnt[k])) continue;
        if (!child_inner(c[k], cnt[k])) { put_child(o, slot++, lo[k], hi[k], c[k], cnt[k]); continue; } /* a leaf stays where it is */
        const DBvhNode m = nodes[c[k]]; /* an inner child (odd depth) hands over its own children (even depth, or leaves) */
        const float* mlo[2] = {m.lo0, m.lo1};
        const float* mhi[2] = {m.hi0, m.hi1};
        const int32_t mc[2] = {m.c0, m.c1}, mn[2] = {m.n0, m.n1};
        for (int j = 0; j < 2; ++j) {
            if (child_empty(mc[j], mn[j])) continue;
            int32_t code = mc[j];
            if (child_inner(mc[j], mn[j])) code = (int32_t)(place ? place[rank[mc[j]]] : rank[mc[j]]);
            put_child(o, slot++, mlo[j], mhi[j], code, mn[j]);
        }
    }
    out[place ? place[rank[i]] : rank[i]] = o;
}

/* DBvh4Node -> DBvh4Q, one thread per node.  The node's box is the union of its children's; scale[a] is chosen so that the
 * largest code, 255, reaches the upper face; every code is then checked against the very expression the traversal evaluates
 * (fmaf(q, scale, lo)) and moved outwards until the decoded bound encloses the child's. */
__global__ void __launch_bounds__(256) k_quantise(const DBvh4Node* in, uint32_t n4, DBvh4Q* out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n4) return;
    const DBvh4Node n = in[i];
    const float* lo[4] = {n.half[0].lo0, n.half[0].lo1, n.half[1].lo0, n.half[1].lo1};
    const float* hi[4] = {n.half[0].hi0, n.half[0].hi1, n.half[1].hi0, n.half[1].hi1};
    const int32_t c[4] = {n.half[0].c0, n.half[0].c1, n.half[1].c0, n.half[1].c1};
    const int32_t cnt[4] = {n.half[0].n0, n.half[0].n1, n.half[1].n0, n.half[1].n1};
    DBvh4Q o;
    bool used[4];
    for (int k = 0; k < 4; ++k) {
        used[k] = !child_empty(c[k], cnt[k]);
        o.child[k] = !used[k] ? RT_BVH4Q_EMPTY : (cnt[k] > 0 ? ~(c[k] * 8 + cnt[k]) : c[k]);
    }
    for (int a = 0; a < 3; ++a) {
        float nlo = INFINITY, nhi = -INFINITY;
        for (int k = 0; k < 4; ++k) if (used[k]) { nlo = fminf(nlo, lo[k][a]); nhi = fmaxf(nhi, hi[k][a]); }
        if (!(nlo <= nhi)) { nlo = 0.f; nhi = 0.f; }
        float sc = (nhi - nlo) / 255.0f;
        if (!(sc > 0.f)) sc = 1e-30f;
        while (fmaf(255.0f, sc, nlo) < nhi) sc = __int_as_float(__float_as_int(sc) + 1); /* a few steps at most */
        o.lo[a] = nlo; o.scale[a] = sc;
        uint32_t wlo = 0, whi = 0;
        for (int k = 0; k < 4; ++k) {
            int ql = 255, qh = 0; /* an unused slot gets the empty box */
            if (used[k]) {
                ql = (int)floorf((lo[k][a] - nlo) / sc); ql = ql < 0 ? 0 : (ql > 255 ? 255 : ql);
                while (ql > 0 && fmaf((float)ql, sc, nlo) > lo[k][a]) --ql;
                qh = (int)ceilf((hi[k][a] - nlo) / sc); qh = qh < 0 ? 0 : (qh > 255 ? 255 : qh);
                while (qh < 255 && fmaf((float)qh, sc, nlo) < hi[k][a]) ++qh;
            }
            wlo |= (uint32_t)ql << (8 * k); whi |= (uint32_t)qh << (8 * k);
        }
        o.q[a] = wlo; o.q[3 + a] = whi;
    }
    out[i] = o;
}

/* ---- refit: new vertex positions, same triangles, same trees ------------------------------------------------- */
/* exact box of the raw positions of one mesh's vertices (what Scene::computeGlobalAABB grows the root box by) */
__global__ void __launch_bounds__(256) k_vertex_box(const float* positions, uint32_t count, uint32_t* box) {
    __shared__ float s_box[4][6];
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint64_t v = (uint64_t)blockIdx.x * 256u + threadIdx.x; v < count; v += (uint64_t)gridDim.x * 256u)
        for (int a = 0; a < 3; ++a) { const float p = positions[3 * v + a]; lo[a] = fminf(lo[a], p); hi[a] = fmaxf(hi[a], p); }
    for (int a = 0; a < 3; ++a)
        for (int off = 32; off > 0; off >>= 1) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], off)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off)); }
    if ((threadIdx.x & 63u) == 0)
        for (int a = 0; a < 3; ++a) { s_box[threadIdx.x >> 6][a] = lo[a]; s_box[threadIdx.x >> 6][3 + a] = hi[a]; }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int k = (int)threadIdx.x;
        float v = s_box[0][k];
        for (int w = 1; w < 4; ++w) v = k < 3 ? fminf(v, s_box[w][k]) : fmaxf(v, s_box[w][k]);
        if (k < 3) atomicMin(&box[k], ordered(v)); else atomicMax(&box[k], ordered(v));
    }
}

/* The records of one mesh again, in the order they have (leaf order), from the vertex ids each record carries: the
 * arithmetic of k_make_tris on the new positions (the whole vertex pool: the ids are global), and the box of the corners. */
__global__ void __launch_bounds__(256) k_remake_tris(const float* positions, uint32_t n_vertices, DTri* tris, uint32_t count, uint32_t* box) {
    __shared__ float s_box[4][6];
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint64_t t = (uint64_t)blockIdx.x * 256u + threadIdx.x; t < count; t += (uint64_t)gridDim.x * 256u) {
        DTri tr = tris[t];
        if (tr.i0 >= n_vertices || tr.i1 >= n_vertices || tr.i2 >= n_vertices) { atomicOr(&box[6], 1u); continue; }
        const float scaling = 1.000001f; /* R/src/Mesh.h:220 */
        const float* p0 = positions + 3 * (size_t)tr.i0;
        const float* p1 = positions + 3 * (size_t)tr.i1;
        const float* p2 = positions + 3 * (size_t)tr.i2;
        for (int a = 0; a < 3; ++a) {
            const float v0 = scaling * p0[a], v1 = scaling * p1[a], v2 = scaling * p2[a];
            tr.v0[a] = v0; tr.e1[a] = v1 - v0; tr.e2[a] = v2 - v0;
        }
        tri_box(tr, lo, hi);
        tris[t] = tr;
    }
    for (int a = 0; a < 3; ++a)
        for (int off = 32; off > 0; off >>= 1) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], off)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off)); }
    if ((threadIdx.x & 63u) == 0)
        for (int a = 0; a < 3; ++a) { s_box[threadIdx.x >> 6][a] = lo[a]; s_box[threadIdx.x >> 6][3 + a] = hi[a]; }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int k = (int)threadIdx.x;
        float v = s_box[0][k];
        for (int w = 1; w < 4; ++w) v = k < 3 ? fminf(v, s_box[w][k]) : fmaxf(v, s_box[w][k]);
        if (k < 3) atomicMin(&box[k], ordered(v)); else atomicMax(&box[k], ordered(v));
    }
}

/* depth of every node below its root, and which of the given roots that is (255: none of them) */
__global__ void __launch_bounds__(256) k_depth_owner(const int32_t* parent, uint32_t n2, const int32_t* roots, uint32_t n_roots, uint8_t* depth, uint8_t* owner) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n2) return;
    uint32_t d = 0;
    int32_t top = (int32_t)i;
    for (int32_t p = parent[i]; p >= 0 && d < 255u; p = parent[p]) { ++d; top = p; }
    uint32_t o = 255u;
    for (uint32_t k = 0; k < n_roots && k < 255u; ++k) if (roots[k] == top) o = k;
    depth[i] = (uint8_t)d; owner[i] = (uint8_t)o;
}

/* the two child boxes of every node of one level, as k_fit makes them: a leaf's triangles and the pad, or the union of
 * an inner child's (already refitted, padded) boxes */
__global__ void __launch_bounds__(256) k_refit_level(DBvhNode* nodes, uint32_t n2, const uint8_t* depth, const uint8_t* owner, uint32_t level,
                                                     const DTri* tris, const float* pads) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n2 || depth[i] != level || owner[i] == 255u) return;
    DBvhNode n = nodes[i];
    const float pad = pads[owner[i]];
    float* olo[2] = {n.lo0, n.lo1};
    float* ohi[2] = {n.hi0, n.hi1};
    const int32_t c[2] = {n.c0, n.c1}, cnt[2] = {n.n0, n.n1};
    for (int k = 0; k < 2; ++k) {
        if (child_empty(c[k], cnt[k])) continue;
        if (child_inner(c[k], cnt[k])) {
            const DBvhNode ch = nodes[c[k]];
            for (int a = 0; a < 3; ++a) { olo[k][a] = fminf(ch.lo0[a], ch.lo1[a]); ohi[k][a] = fmaxf(ch.hi0[a], ch.hi1[a]); }
        } else {
            float blo[3] = {INFINITY, INFINITY, INFINITY}, bhi[3] = {-INFINITY, -INFINITY, -INFINITY};
            for (int32_t t = c[k]; t < c[k] + cnt[k]; ++t) tri_box(tris[t], blo, bhi);
            for (int a = 0; a < 3; ++a) { olo[k][a] = blo[a] - pad; ohi[k][a] = bhi[a] + pad; }
        }
    }
    nodes[i] = n;
}

struct Temps {
    std::vector<void*> ptrs;
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    ~Temps() {
        for (void* p : ptrs) (void)hipFree(p);
        for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e);
    }
    template <class T>
    hipError_t get(T** out, size_t n) {
        void* p = nullptr;
        hipError_t e = hipMalloc(&p, (n ? n : 1) * sizeof(T));
        if (e == hipSuccess) ptrs.push_back(p);
        *out = (T*)p;
        return e;
    }
};

inline unsigned blocks(uint32_t n) { return (n + 255u) / 256u; }

}  // namespace

#define BVH_TRY(expr)                                                     \
    do {                                                                  \
        hipError_t e_ = (expr);                                           \
        if (e_ != hipSuccess) { err = std::string("device BVH build: ") + #expr + ": " + hipGetErrorString(e_); return e_; } \
    } while (0)

void box_init(uint32_t box[8]) {
    for (int a = 0; a < 3; ++a) { box[a] = ordered(INFINITY); box[3 + a] = ordered(-INFINITY); }
    box[6] = box[7] = 0;
}

void box_decode(const uint32_t box[8], float lo[3], float hi[3], bool& bad_index) {
    for (int a = 0; a < 3; ++a) { lo[a] = unordered(box[a]); hi[a] = unordered(box[3 + a]); }
    bad_index = box[6] != 0;
}

hipError_t make_mesh_tris(const float* d_positions, const uint32_t* d_triangles, uint32_t first_vertex, uint32_t n_vertices,
                          uint32_t n_triangles, DTri* d_tris, uint32_t* d_tri_orig, uint32_t* d_box, hipStream_t stream) {
    if (n_triangles == 0) return hipSuccess;
    const unsigned grid = blocks(n_triangles) < 4096u ? blocks(n_triangles) : 4096u;
    hipLaunchKernelGGL(k_make_tris, dim3(grid), dim3(256), 0, stream, d_positions, d_triangles, first_vertex, n_vertices, n_triangles,
                       d_tris, d_tri_orig, d_box);
    return hipGetLastError();
}

hipError_t build_mesh_bvh(DTri* d_tris, uint32_t* d_tri_orig, uint32_t first, uint32_t count, const float lo[3],
                          const float hi[3], float pad, int32_t node_base, int max_levels, hipStream_t stream,
                          DeviceBvh& out, std::string& err) {
    out = DeviceBvh();
    if (count <= kLeafMax || count > 0x7fffffffu) { err = "device BVH build: triangle count outside (4, 2^31)"; return hipErrorInvalidValue; }
    Temps tmp;
    for (hipEvent_t& e : tmp.ev) BVH_TRY(hipEventCreate(&e));
    uint64_t *keys_a, *keys_b;
    uint32_t *ids_a, *ids_b, *flags, *rank, *n_items;
    DTri* sorted;
    Item* items;
    BVH_TRY(tmp.get(&keys_a, count)); BVH_TRY(tmp.get(&keys_b, count));
    BVH_TRY(tmp.get(&ids_a, count)); BVH_TRY(tmp.get(&ids_b, count));
    BVH_TRY(tmp.get(&sorted, count));
    BVH_TRY(tmp.get(&items, count));
    BVH_TRY(tmp.get(&flags, (size_t)count + 1)); BVH_TRY(tmp.get(&rank, (size_t)count + 1));
    BVH_TRY(tmp.get(&n_items, 1));
    size_t sort_bytes = 0, scan_bytes = 0;
    BVH_TRY(rocprim::radix_sort_pairs(nullptr, sort_bytes, keys_a, keys_b, ids_a, ids_b, (size_t)count, 0u, 63u, stream));
    BVH_TRY(rocprim::exclusive_scan(nullptr, scan_bytes, flags, rank, 0u, (size_t)count + 1, rocprim::plus<uint32_t>(), stream));
    char *sort_tmp, *scan_tmp;
    BVH_TRY(tmp.get(&sort_tmp, sort_bytes)); BVH_TRY(tmp.get(&scan_tmp, scan_bytes));

    /* keys: cubic cells over the longest edge of the mesh box, 2^21 along it */
    float ext = fmaxf(fmaxf(hi[0] - lo[0], hi[1] - lo[1]), hi[2] - lo[2]);
    float cells_per_unit = (ext > 0.f && ext < INFINITY) ? 2097152.f / ext : 0.f;
    DTri* mesh_tris = d_tris + first;
    BVH_TRY(hipEventRecord(tmp.ev[0], stream));
    hipLaunchKernelGGL(k_tri_keys, dim3(blocks(count)), dim3(256), 0, stream, (const DTri*)mesh_tris, count, F3{lo[0], lo[1], lo[2]}, cells_per_unit, keys_a, ids_a);
    BVH_TRY(hipGetLastError());
    BVH_TRY(rocprim::radix_sort_pairs(sort_tmp, sort_bytes, keys_a, keys_b, ids_a, ids_b, (size_t)count, 0u, 63u, stream));
    hipLaunchKernelGGL(k_permute, dim3(blocks(count)), dim3(256), 0, stream, (const DTri*)mesh_tris, (const uint32_t*)ids_b, count, sorted, d_tri_orig + first);
    BVH_TRY(hipGetLastError());
    BVH_TRY(hipMemcpyAsync(mesh_tris, sorted, (size_t)count * sizeof(DTri), hipMemcpyDeviceToDevice, stream));
    BVH_TRY(hipEventRecord(tmp.ev[1], stream));

    /* hierarchy, one launch per level */
    BVH_TRY(hipMemsetAsync(flags, 0, ((size_t)count + 1) * 4, stream));
    Item root; root.lo = 0; root.hi = count; root.mid = 0; root.kid[0] = root.kid[1] = 0xffffffffu;
    uint32_t one = 1;
    BVH_TRY(hipMemcpyAsync(items, &root, sizeof(Item), hipMemcpyHostToDevice, stream));
    BVH_TRY(hipMemcpyAsync(n_items, &one, 4, hipMemcpyHostToDevice, stream));
    BVH_TRY(hipStreamSynchronize(stream)); /* root and one live on this stack frame */
    uint32_t* level; /* level d = items [level[d], level[d + 1]) */
    BVH_TRY(tmp.get(&level, (size_t)max_levels + 2));
    const uint32_t first_levels[2] = {0u, 1u};
    BVH_TRY(hipMemcpyAsync(level, first_levels, 8, hipMemcpyHostToDevice, stream));
    BVH_TRY(hipStreamSynchronize(stream)); /* first_levels lives on this stack frame */
    for (int depth = 0; depth < max_levels; ++depth) {
        /* a level holds at most 2^depth nodes, and at most one per kLeafMax + 1 triangles */
        const uint64_t most = std::min<uint64_t>(depth < 31 ? (1ull << depth) : (1ull << 31), (uint64_t)count / (kLeafMax + 1) + 1);
        hipLaunchKernelGGL(k_split, dim3(blocks((uint32_t)most)), dim3(256), 0, stream, items, (const uint32_t*)level, depth, max_levels + 2,
                           (const uint64_t*)keys_b, flags, n_items, count);
        hipLaunchKernelGGL(k_level_end, dim3(1), dim3(1), 0, stream, level, depth, (const uint32_t*)n_items, count);
        BVH_TRY(hipGetLastError());
    }
    std::vector<uint32_t> level_host((size_t)max_levels + 2);
    uint32_t total = 0;
    BVH_TRY(hipMemcpyAsync(level_host.data(), level, level_host.size() * 4, hipMemcpyDeviceToHost, stream));
    BVH_TRY(hipMemcpyAsync(&total, n_items, 4, hipMemcpyDeviceToHost, stream));
    BVH_TRY(hipStreamSynchronize(stream));
    if (total > count) { err = "device BVH build: node queue overflow"; return hipErrorInvalidValue; }
    if (level_host[(size_t)max_levels + 1] > level_host[(size_t)max_levels]) { err = "device BVH build: tree deeper than the traversal stack"; return hipErrorInvalidValue; }
    std::vector<uint32_t> level_end; /* ends of the levels that hold nodes */
    for (int depth = 0; depth < max_levels && level_host[(size_t)depth + 1] > level_host[(size_t)depth]; ++depth) level_end.push_back(level_host[(size_t)depth + 1]);
    const uint32_t end = level_end.empty() ? 0u : level_end.back();
    const uint32_t n_nodes = end;
    BVH_TRY(rocprim::exclusive_scan(scan_tmp, scan_bytes, flags, rank, 0u, (size_t)count + 1, rocprim::plus<uint32_t>(), stream));
    DBvhNode* nodes = nullptr;
    BVH_TRY(hipMalloc((void**)&nodes, (size_t)n_nodes * sizeof(DBvhNode)));
    hipError_t e = hipSuccess;
    for (int d = (int)level_end.size() - 1; d >= 0 && e == hipSuccess; --d) {
        const uint32_t b = d ? level_end[(size_t)d - 1] : 0u, en = level_end[(size_t)d];
        hipLaunchKernelGGL(k_fit, dim3(blocks(en - b)), dim3(256), 0, stream, (const Item*)items, b, en, (const uint32_t*)rank, (const DTri*)mesh_tris, first,
                           node_base, pad, nodes);
        e = hipGetLastError();
    }
    Item root_done;
    uint32_t root_rank = 0;
    if (e == hipSuccess) e = hipEventRecord(tmp.ev[2], stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&root_done, items, sizeof(Item), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e == hipSuccess) e = hipMemcpy(&root_rank, rank + root_done.mid, 4, hipMemcpyDeviceToHost);
    if (e != hipSuccess) {
        (void)hipFree(nodes);
        err = std::string("device BVH build: box pass: ") + hipGetErrorString(e);
        return e;
    }
    out.nodes = nodes;
    out.n_nodes = n_nodes;
    out.root = node_base + (int32_t)root_rank;
    out.max_depth = (int)level_end.size();
    (void)hipEventElapsedTime(&out.sort_ms, tmp.ev[0], tmp.ev[1]);
    (void)hipEventElapsedTime(&out.tree_ms, tmp.ev[1], tmp.ev[2]);
    return hipSuccess;
}

hipError_t collapse_to_bvh4(const DBvhNode* d_bvh2, uint32_t n2, const int32_t* roots, uint32_t n_roots, hipStream_t stream,
                            DBvh4Node** d_bvh4, uint32_t* n4, int32_t* roots4, std::string& err, DBvh4Node* existing, uint32_t n_existing,
                            const uint32_t* d_place) {
    *d_bvh4 = nullptr; *n4 = 0;
    if (n2 == 0) return hipSuccess;
    Temps tmp;
    int32_t* parent;
    uint32_t *keep, *rank;
    void* scan_tmp = nullptr;
    size_t scan_bytes = 0;
    BVH_TRY(tmp.get(&parent, n2));
    BVH_TRY(tmp.get(&keep, (size_t)n2 + 1));
    BVH_TRY(tmp.get(&rank, (size_t)n2 + 1));
    BVH_TRY(hipMemsetAsync(parent, 0xff, (size_t)n2 * 4, stream));
    BVH_TRY(hipMemsetAsync(keep, 0, ((size_t)n2 + 1) * 4, stream));
    hipLaunchKernelGGL(k_parents, dim3(blocks(n2)), dim3(256), 0, stream, d_bvh2, n2, parent);
    hipLaunchKernelGGL(k_even_depth, dim3(blocks(n2)), dim3(256), 0, stream, (const int32_t*)parent, n2, keep);
    BVH_TRY(hipGetLastError());
    BVH_TRY(rocprim::exclusive_scan(nullptr, scan_bytes, keep, rank, 0u, (size_t)n2 + 1, rocprim::plus<uint32_t>(), stream));
    BVH_TRY(tmp.get((char**)&scan_tmp, scan_bytes));
    BVH_TRY(rocprim::exclusive_scan(scan_tmp, scan_bytes, keep, rank, 0u, (size_t)n2 + 1, rocprim::plus<uint32_t>(), stream));
    uint32_t total = 0;
    BVH_TRY(hipMemcpyAsync(&total, rank + n2, 4, hipMemcpyDeviceToHost, stream));
    for (uint32_t k = 0; k < n_roots; ++k) {
        if (roots[k] < 0 || (uint32_t)roots[k] >= n2) { err = "four-wide BVH: root outside the node table"; return hipErrorInvalidValue; }
        BVH_TRY(hipMemcpyAsync(&roots4[k], rank + roots[k], 4, hipMemcpyDeviceToHost, stream));
    }
    BVH_TRY(hipStreamSynchronize(stream));
    if (d_place)
        for (uint32_t k = 0; k < n_roots; ++k) BVH_TRY(hipMemcpy(&roots4[k], d_place + roots4[k], 4, hipMemcpyDeviceToHost));
    DBvh4Node* out = existing;
    if (existing && n_existing != total) { err = "four-wide BVH: the table to refresh has another size"; return hipErrorInvalidValue; }
    if (!existing) BVH_TRY(hipMalloc((void**)&out, (size_t)(total ? total : 1) * sizeof(DBvh4Node)));
    hipLaunchKernelGGL(k_collapse, dim3(blocks(n2)), dim3(256), 0, stream, d_bvh2, n2, (const uint32_t*)keep, (const uint32_t*)rank, d_place, out);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) { if (!existing) (void)hipFree(out); err = std::string("four-wide BVH: ") + hipGetErrorString(e); return e; }
    *d_bvh4 = out; *n4 = total;
    return hipSuccess;
}

/* Table order by treelets: depth first over the trees, and at every node the inner children of that node get consecutive
 * places before any of them is descended into -- the two to four nodes a ray may visit next lie within 256 bytes, and the
 * subtree below a node is one contiguous run of the table. */
hipError_t treelet_places(const DBvh4Q* d_nodes, uint32_t n4, const int32_t* roots4, uint32_t n_roots, uint32_t** d_place, std::string& err) {
    *d_place = nullptr;
    if (n4 == 0) return hipSuccess;
    std::vector<DBvh4Q> nodes(n4);
    BVH_TRY(hipMemcpy(nodes.data(), d_nodes, (size_t)n4 * sizeof(DBvh4Q), hipMemcpyDeviceToHost));
    std::vector<uint32_t> place(n4, 0xffffffffu), todo;
    uint32_t next = 0;
    for (uint32_t k = 0; k < n_roots; ++k) {
        if (roots4[k] < 0 || (uint32_t)roots4[k] >= n4 || place[roots4[k]] != 0xffffffffu) { err = "four-wide BVH: bad root"; return hipErrorInvalidValue; }
        place[roots4[k]] = next++;
        todo.push_back((uint32_t)roots4[k]);
        while (!todo.empty()) {
            const uint32_t at = todo.back();
            todo.pop_back();
            const DBvh4Q& n = nodes[at];
            for (int s = 0; s < 4; ++s) {
                const int32_t c = n.child[s];
                if (c < 0) continue; /* leaf or unused */
                if ((uint32_t)c >= n4 || place[c] != 0xffffffffu) { err = "four-wide BVH: a node has two parents"; return hipErrorInvalidValue; }
                place[c] = next++;
            }
            for (int s = 3; s >= 0; --s) if (n.child[s] >= 0) todo.push_back((uint32_t)n.child[s]);
        }
    }
    for (uint32_t i = 0; i < n4; ++i) if (place[i] == 0xffffffffu) place[i] = next++; /* nodes no root reaches keep a slot */
    if (next != n4) { err = "four-wide BVH: order does not cover the table"; return hipErrorInvalidValue; }
    uint32_t* d = nullptr;
    BVH_TRY(hipMalloc((void**)&d, (size_t)n4 * 4));
    hipError_t e = hipMemcpy(d, place.data(), (size_t)n4 * 4, hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(d); err = std::string("four-wide BVH: ") + hipGetErrorString(e); return e; }
    *d_place = d;
    return hipSuccess;
}

hipError_t quantise_bvh4(const DBvh4Node* d_bvh4, uint32_t n4, DBvh4Q* d_out, hipStream_t stream) {
    if (n4 == 0) return hipSuccess;
    hipLaunchKernelGGL(k_quantise, dim3(blocks(n4)), dim3(256), 0, stream, d_bvh4, n4, d_out);
    return hipGetLastError();
}

hipError_t vertex_box(const float* d_positions, uint32_t count, uint32_t* d_box, hipStream_t stream) {
    if (count == 0) return hipSuccess;
    const unsigned grid = blocks(count) < 1024u ? blocks(count) : 1024u;
    hipLaunchKernelGGL(k_vertex_box, dim3(grid), dim3(256), 0, stream, d_positions, count, d_box);
    return hipGetLastError();
}

hipError_t remake_mesh_tris(const float* d_positions, uint32_t n_vertices, DTri* d_tris, uint32_t count, uint32_t* d_box, hipStream_t stream) {
    if (count == 0) return hipSuccess;
    const unsigned grid = blocks(count) < 4096u ? blocks(count) : 4096u;
    hipLaunchKernelGGL(k_remake_tris, dim3(grid), dim3(256), 0, stream, d_positions, n_vertices, d_tris, count, d_box);
    return hipGetLastError();
}

hipError_t refit_bvh2(DBvhNode* d_bvh2, uint32_t n2, const int32_t* roots, const float* pads, uint32_t n_roots, const DTri* d_tris, hipStream_t stream,
                      std::string& err) {
    if (n2 == 0 || n_roots == 0) return hipSuccess;
    if (n_roots > 254u) { err = "BVH refit: more than 254 meshes with a BVH"; return hipErrorInvalidValue; }
    Temps tmp;
    int32_t *parent, *d_roots;
    uint8_t *depth, *owner;
    float* d_pads;
    BVH_TRY(tmp.get(&parent, n2)); BVH_TRY(tmp.get(&depth, n2)); BVH_TRY(tmp.get(&owner, n2));
    BVH_TRY(tmp.get(&d_roots, n_roots)); BVH_TRY(tmp.get(&d_pads, n_roots));
    BVH_TRY(hipMemcpyAsync(d_roots, roots, (size_t)n_roots * 4, hipMemcpyHostToDevice, stream));
    BVH_TRY(hipMemcpyAsync(d_pads, pads, (size_t)n_roots * 4, hipMemcpyHostToDevice, stream));
    BVH_TRY(hipMemsetAsync(parent, 0xff, (size_t)n2 * 4, stream));
    hipLaunchKernelGGL(k_parents, dim3(blocks(n2)), dim3(256), 0, stream, (const DBvhNode*)d_bvh2, n2, parent);
    hipLaunchKernelGGL(k_depth_owner, dim3(blocks(n2)), dim3(256), 0, stream, (const int32_t*)parent, n2, (const int32_t*)d_roots, n_roots, depth, owner);
    BVH_TRY(hipGetLastError());
    /* deepest level first; a tree has at most 32 inner nodes on a path (the traversal stack), the table says nothing shorter */
    for (int level = 32; level >= 0; --level) {
        hipLaunchKernelGGL(k_refit_level, dim3(blocks(n2)), dim3(256), 0, stream, d_bvh2, n2, (const uint8_t*)depth, (const uint8_t*)owner, (uint32_t)level,
                           d_tris, (const float*)d_pads);
        BVH_TRY(hipGetLastError());
    }
    BVH_TRY(hipStreamSynchronize(stream)); /* roots / pads / the temporaries live until here */
    return hipSuccess;
}

}  // namespace rtd
