/*
 * rt_build.cpp -- host side of rt_scene_create: validates a flat scene, precomputes the
 * ray-independent terms, builds the object-level kd-tree the way the reference does on the host
 * (R/src/KDTree.cpp:363-519) and inverts the camera matrices (R/src/matrixUtilities.h:6-135).
 *
 * Built with -ffp-contract=off: every float operation here must round exactly like the
 * reference's per-ray evaluation of the same expression.
 */
#include "rt_build.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <limits>
#include <thread>

namespace rtb {

namespace {

const float kInf = std::numeric_limits<float>::infinity();

struct F3 { float x, y, z; };
inline F3 ld(const float* p) { return {p[0], p[1], p[2]}; }
inline void st(float* p, F3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
inline F3 operator-(F3 a, F3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline F3 operator*(float s, F3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline F3 cross(F3 a, F3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float len(F3 a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
inline float fmax_std(float a, float b) { return (a < b) ? b : a; }  /* std::max */
inline float fmin_std(float a, float b) { return (b < a) ? b : a; }  /* std::min */

struct Box { float lo[3] = {kInf, kInf, kInf}, hi[3] = {-kInf, -kInf, -kInf}; };
inline void grow(Box& b, const float* p) {  /* AABB::expandByPoint, R/src/AABB.h:76-84 */
    for (int a = 0; a < 3; ++a) { b.lo[a] = fmin_std(b.lo[a], p[a]); b.hi[a] = fmax_std(b.hi[a], p[a]); }
}

/* box the kd builder assigns to one object, R/src/KDTree.cpp:393-416 */
Box object_box(const rt_scene_desc& d, const DRef& r) {
    Box b;
    if (r.type == RT_SPHERE) {
        const rt_sphere& s = d.spheres[r.object];
        for (int a = 0; a < 3; ++a) { b.lo[a] = s.center[a] - s.radius; b.hi[a] = s.center[a] + s.radius; }
    } else if (r.type == RT_SQUARE) {
        const rt_square& q = d.squares[r.object];
        for (int a = 0; a < 3; ++a) b.lo[a] = b.hi[a] = q.v[0][a];
        for (int v = 1; v < 3; ++v)  /* vertices 1 and 2 only, as the reference loops */
            for (int a = 0; a < 3; ++a) {
                if (q.v[v][a] < b.lo[a]) b.lo[a] = q.v[v][a];
                if (q.v[v][a] > b.hi[a]) b.hi[a] = q.v[v][a];
            }
    } else {
        const rt_mesh& m = d.meshes[r.object];
        for (int a = 0; a < 3; ++a) { b.lo[a] = m.aabb_min[a]; b.hi[a] = m.aabb_max[a]; }
    }
    return b;
}

/* KDTree::buildNode (V3), R/src/KDTree.cpp:363-473, on the index range [first, first+count) of
 * out.refs.  The reference copies the list into two children in list order; a stable partition of
 * the range is the same order, so every leaf is a contiguous slice. */
int32_t build_node(const rt_scene_desc& d, Built& out, const Box& box, int32_t first, int32_t count, int32_t depth) {
    int32_t id = (int32_t)out.kd.size();
    out.kd.push_back(DKdNode{0, -1, 0.f, -1, -1, first, 0, depth});
    auto leaf = [&]() { out.kd[id].leaf = 1; out.kd[id].n_refs = count; return id; };
    if (count <= d.kd_max_leaf || depth >= d.kd_max_depth) return leaf();
    float ex = box.hi[0] - box.lo[0], ey = box.hi[1] - box.lo[1], ez = box.hi[2] - box.lo[2];
    if (fmax_std(fmax_std(ex, ey), ez) < d.kd_epsilon) return leaf();
    int axis = 0;
    if (ey >= ex && ey >= ez) axis = 1;
    else if (ez >= ex && ez >= ey) axis = 2;
    float cut = 0.f;
    for (int32_t i = 0; i < count; ++i) {
        Box b = object_box(d, out.refs[first + i]);
        cut += (b.lo[axis] + b.hi[axis]) * 0.5f;
    }
    cut /= (float)(size_t)count;
    auto goes_left = [&](const DRef& r) {
        Box b = object_box(d, r);
        return 0.5f * (b.lo[axis] + b.hi[axis]) <= cut;
    };
    auto mid = std::stable_partition(out.refs.begin() + first, out.refs.begin() + first + count, goes_left);
    int32_t n_left = (int32_t)(mid - (out.refs.begin() + first));
    out.kd[id].axis = axis;
    out.kd[id].cut = cut;
    if (n_left == 0 || n_left == count) return leaf();  /* list order is untouched in that case */
    Box bl = box, br = box;
    bl.hi[axis] = cut;
    br.lo[axis] = cut;
    int32_t l = build_node(d, out, bl, first, n_left, depth + 1);
    int32_t r = build_node(d, out, br, first + n_left, count - n_left, depth + 1);
    out.kd[id].left = l;
    out.kd[id].right = r;
    return id;
}

bool fail(std::string& err, const char* what) { err = what; return false; }

/* smallest positive float x (possibly +inf) for which pred(x) holds; pred must be monotonic */
template <class Pred>
float first_float_where(Pred pred) {
    uint32_t lo = 1, hi = 0x7f800000u; /* smallest subnormal .. +inf, ordered like the floats */
    auto as_float = [](uint32_t b) { float f; std::memcpy(&f, &b, 4); return f; };
    if (!pred(as_float(hi))) return std::numeric_limits<float>::quiet_NaN(); /* nothing passes */
    while (lo < hi) {
        uint32_t mid = lo + (hi - lo) / 2;
        if (pred(as_float(mid))) hi = mid; else lo = mid + 1;
    }
    return as_float(lo);
}
/* interval [lo, hi) of x = dot(P - v0, edge) that the reference's `0 < x/len && x/len < len` accepts */
void edge_interval(float len, float& lo, float& hi) {
    lo = std::numeric_limits<float>::infinity(); hi = 0.f; /* empty */
    if (!(len > 0.f) || std::isinf(len)) return;
    volatile float L = len; /* keep the real division: no reciprocal tricks by the host compiler */
    float a = first_float_where([&](float x) { return x / L > 0.f; });
    float b = first_float_where([&](float x) { return x / L >= L; });
    if (std::isnan(a)) return;
    lo = a;
    hi = std::isnan(b) ? std::numeric_limits<float>::infinity() : b;
}


/* ---- triangle BVH (per mesh) ---------------------------------------------------------------
 * Binned surface-area-heuristic build over the scaled triangles the device tests (v0, v0+e1, v0+e2).
 * Leaves hold at most 4 triangles.  A node at depth D with c triangles splits by SAH only while
 * D + 1 + ceil(log2 c) <= kBvhMaxDepth + 2, otherwise at the median: median splits below any node
 * then end within kBvhMaxDepth inner levels, which is the size of the device's traversal stack.  Leaf triangles are
 * written back in leaf order, with their original indices kept in tri_orig. */
struct BBox {
    float lo[3] = {kInf, kInf, kInf}, hi[3] = {-kInf, -kInf, -kInf};
    void grow(const float* p) { for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], p[a]); hi[a] = std::max(hi[a], p[a]); } }
    void grow(const BBox& b) { grow(b.lo); grow(b.hi); }
    float area() const {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return (dx < 0 || dy < 0 || dz < 0) ? 0.f : 2.f * (dx * dy + dy * dz + dz * dx);
    }
};
struct TriInfo { BBox box; float c[3]; uint32_t id; };

struct BvhBuilder {
    std::vector<TriInfo>& prims;
    std::vector<DBvhNode>& nodes;
    std::vector<uint32_t> order; /* triangle ids in leaf order */
    float pad;
    int max_depth = 0; /* inner nodes on the longest path */
    /* When `defer` is set, subtrees of at most `grain` triangles are not built but queued (in left-to-right
     * order) and their parent gets a placeholder child kDeferred - task: other builders finish them in parallel. */
    struct Task { uint32_t first, count; int depth; };
    std::vector<Task>* defer = nullptr;
    uint32_t grain = 0;
    static constexpr int32_t kDeferred = -0x40000000;
    BvhBuilder(std::vector<TriInfo>& p, std::vector<DBvhNode>& n, float pad_) : prims(p), nodes(n), pad(pad_) {}

    struct Child { bool leaf; int32_t index; int32_t count; BBox box; };

    Child build(uint32_t first, uint32_t count, int depth) {
        BBox box, cbox;
        for (uint32_t i = first; i < first + count; ++i) { box.grow(prims[i].box); cbox.grow(prims[i].c); }
        if (count <= 4) return make_leaf(first, count, box);
        if (defer && count <= grain) {
            defer->push_back(Task{first, count, depth});
            return Child{false, kDeferred - (int32_t)(defer->size() - 1), 0, box};
        }
        int axis = 0;
        float ext[3] = {cbox.hi[0] - cbox.lo[0], cbox.hi[1] - cbox.lo[1], cbox.hi[2] - cbox.lo[2]};
        if (ext[1] > ext[axis]) axis = 1;
        if (ext[2] > ext[axis]) axis = 2;
        uint32_t mid = first + count / 2;
        bool split_ok = false;
        int log2c = 0;
        while ((1ull << log2c) < count) ++log2c;
        max_depth = std::max(max_depth, depth + 1);
        if (depth + 1 + log2c <= kBvhMaxDepth + 2 && ext[axis] > 0.f) { /* binned SAH */
            const int B = 16;
            BBox bb[B]; uint32_t bn[B] = {0};
            float k = B * (1.f - 1e-6f) / ext[axis];
            auto bin_of = [&](const TriInfo& t) { int b = (int)((t.c[axis] - cbox.lo[axis]) * k); return b < 0 ? 0 : (b >= B ? B - 1 : b); };
            for (uint32_t i = first; i < first + count; ++i) { int b = bin_of(prims[i]); bb[b].grow(prims[i].box); ++bn[b]; }
            float right_area[B]; BBox acc; uint32_t nacc = 0;
            for (int b = B - 1; b > 0; --b) { acc.grow(bb[b]); right_area[b] = acc.area(); }
            BBox left; float best = kInf; int best_b = -1; uint32_t nl = 0;
            for (int b = 0; b < B - 1; ++b) {
                left.grow(bb[b]); nl += bn[b];
                uint32_t nr = count - nl;
                if (nl == 0 || nr == 0) continue;
                float cost = left.area() * nl + right_area[b + 1] * nr;
                if (cost < best) { best = cost; best_b = b; }
            }
            (void)nacc;
            if (best_b >= 0) {
                auto it = std::partition(prims.begin() + first, prims.begin() + first + count, [&](const TriInfo& t) { return bin_of(t) <= best_b; });
                mid = (uint32_t)(it - prims.begin());
                split_ok = mid > first && mid < first + count;
            }
        }
        if (!split_ok) { /* median split along the widest centroid axis: halves the range */
            mid = first + count / 2;
            std::nth_element(prims.begin() + first, prims.begin() + mid, prims.begin() + first + count,
                             [&](const TriInfo& a, const TriInfo& b) { return a.c[axis] < b.c[axis]; });
        }
        int32_t id = (int32_t)nodes.size();
        nodes.emplace_back();
        Child l = build(first, mid - first, depth + 1);
        Child r = build(mid, first + count - mid, depth + 1);
        DBvhNode& n = nodes[id];
        store(n.lo0, n.hi0, n.c0, n.n0, l);
        store(n.lo1, n.hi1, n.c1, n.n1, r);
        return Child{false, id, 0, box};
    }
    Child make_leaf(uint32_t first, uint32_t count, const BBox& box) {
        int32_t start = (int32_t)order.size();
        for (uint32_t i = first; i < first + count; ++i) order.push_back(prims[i].id);
        return Child{true, start, (int32_t)count, box};
    }
    void store(float* lo, float* hi, int32_t& c, int32_t& n, const Child& ch) {
        for (int a = 0; a < 3; ++a) { lo[a] = ch.box.lo[a] - pad; hi[a] = ch.box.hi[a] + pad; }
        c = ch.index; n = ch.leaf ? ch.count : 0;
    }
};

/* Padded box and bounding sphere of a mesh's (scaled) triangles: what a ray has to come near before any of its
 * triangles can report a hit.  The pad covers the rounding of the float ray / triangle test near the faces;
 * it is returned because the node boxes of the mesh's BVH take the same one. */
float set_mesh_bounds(const Built& out, uint32_t first, uint32_t count, DMesh& mesh) {
    BBox all;
    for (uint32_t i = 0; i < count; ++i) {
        const DTri& t = out.tris[first + i];
        float v1[3], v2[3];
        for (int a = 0; a < 3; ++a) { v1[a] = t.v0[a] + t.e1[a]; v2[a] = t.v0[a] + t.e2[a]; }
        all.grow(t.v0); all.grow(v1); all.grow(v2);
    }
    return mesh_bounds_from_box(all.lo, all.hi, count, mesh);
}

}  // namespace

float mesh_bounds_from_box(const float lo[3], const float hi[3], uint32_t count, DMesh& mesh) {
    if (count == 0 || !(lo[0] <= hi[0])) { /* nothing to hit: an empty box and a sphere no line reaches */
        for (int a = 0; a < 3; ++a) { mesh.lo[a] = kInf; mesh.hi[a] = -kInf; mesh.bs_c[a] = 0.f; }
        mesh.bs_r2 = -1.f;
        return 0.f;
    }
    float diag = std::sqrt((hi[0] - lo[0]) * (hi[0] - lo[0]) + (hi[1] - lo[1]) * (hi[1] - lo[1]) + (hi[2] - lo[2]) * (hi[2] - lo[2]));
    float pad = 1e-4f * diag + 1e-6f;
    for (int a = 0; a < 3; ++a) { mesh.lo[a] = lo[a] - pad; mesh.hi[a] = hi[a] + pad; }
    /* bounding sphere of the padded box, in double, every rounding outward */
    double c[3], r2 = 0.0;
    for (int a = 0; a < 3; ++a) { mesh.bs_c[a] = (float)(0.5 * ((double)lo[a] + (double)hi[a])); c[a] = mesh.bs_c[a]; }
    for (int a = 0; a < 3; ++a) {
        double e = std::max(std::fabs((double)lo[a] - pad - c[a]), std::fabs((double)hi[a] + pad - c[a]));
        r2 += e * e;
    }
    float r2f = (float)(r2 * 1.0001);
    mesh.bs_r2 = std::isfinite(r2f) ? std::nextafter(r2f, kInf) : kInf;
    return pad;
}

namespace {

/* Builds the BVH of the triangles [first, first+count) of out.tris, reorders them into leaf order
 * and returns the root node index (a mesh with one leaf still gets a root node). */
int32_t build_mesh_bvh(Built& out, uint32_t first, uint32_t count, DMesh& mesh) {
    std::vector<TriInfo> prims(count);
    BBox all;
    for (uint32_t i = 0; i < count; ++i) {
        const DTri& t = out.tris[first + i];
        float v1[3], v2[3];
        for (int a = 0; a < 3; ++a) { v1[a] = t.v0[a] + t.e1[a]; v2[a] = t.v0[a] + t.e2[a]; }
        TriInfo& p = prims[i];
        p.box.grow(t.v0); p.box.grow(v1); p.box.grow(v2);
        for (int a = 0; a < 3; ++a) p.c[a] = 0.5f * (p.box.lo[a] + p.box.hi[a]);
        p.id = i;
        all.grow(p.box);
    }
    float diag = std::sqrt((all.hi[0] - all.lo[0]) * (all.hi[0] - all.lo[0]) + (all.hi[1] - all.lo[1]) * (all.hi[1] - all.lo[1]) +
                           (all.hi[2] - all.lo[2]) * (all.hi[2] - all.lo[2]));
    float pad = 1e-4f * diag + 1e-6f; /* covers the rounding of the float ray / triangle test near box faces */
    int32_t root = (int32_t)out.bvh.size();
    BvhBuilder b(prims, out.bvh, pad);
    /* Large meshes: the top of the tree is split here, the subtrees below it by a pool of threads, each into
     * its own node and order arrays (the triangle ranges are disjoint); they are appended afterwards in
     * left-to-right order and the placeholders in the top nodes patched. */
    std::vector<BvhBuilder::Task> tasks;
    unsigned n_threads = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    if (count >= 200000 && n_threads > 1) { b.defer = &tasks; b.grain = std::max<uint32_t>(4096, count / (8 * n_threads)); }
    BvhBuilder::Child top = b.build(0, count, 0);
    if (top.leaf) return -1; /* cannot happen for count > 4; the caller then keeps the brute-force loop */
    if (!tasks.empty()) {
        struct Sub { std::vector<DBvhNode> nodes; std::vector<uint32_t> order; BvhBuilder::Child top; int max_depth = 0; };
        std::vector<Sub> subs(tasks.size());
        std::atomic<size_t> next{0};
        auto worker = [&]() {
            for (size_t t; (t = next.fetch_add(1)) < tasks.size();) {
                BvhBuilder sb(prims, subs[t].nodes, pad);
                subs[t].top = sb.build(tasks[t].first, tasks[t].count, tasks[t].depth);
                subs[t].order.swap(sb.order);
                subs[t].max_depth = sb.max_depth;
            }
        };
        std::vector<std::thread> pool;
        for (unsigned i = 0; i + 1 < n_threads; ++i) pool.emplace_back(worker);
        worker();
        for (std::thread& th : pool) th.join();
        const size_t top_nodes = out.bvh.size();
        std::vector<BvhBuilder::Child> done(tasks.size());
        for (size_t t = 0; t < tasks.size(); ++t) {
            const int32_t node_base = (int32_t)out.bvh.size(), order_base = (int32_t)b.order.size();
            for (DBvhNode n : subs[t].nodes) { /* sub-builder indices are relative to its own arrays */
                if (n.n0 > 0) n.c0 += order_base; else if (n.c0 >= 0) n.c0 += node_base;
                if (n.n1 > 0) n.c1 += order_base; else if (n.c1 >= 0) n.c1 += node_base;
                out.bvh.push_back(n);
            }
            done[t] = subs[t].top;
            done[t].index += done[t].leaf ? order_base : node_base;
            b.order.insert(b.order.end(), subs[t].order.begin(), subs[t].order.end());
            b.max_depth = std::max(b.max_depth, subs[t].max_depth);
        }
        for (size_t k = (size_t)root; k < top_nodes; ++k) {
            DBvhNode& n = out.bvh[k];
            if (n.n0 == 0 && n.c0 <= BvhBuilder::kDeferred) { const BvhBuilder::Child& c = done[(size_t)(BvhBuilder::kDeferred - n.c0)]; n.c0 = c.index; n.n0 = c.leaf ? c.count : 0; }
            if (n.n1 == 0 && n.c1 <= BvhBuilder::kDeferred) { const BvhBuilder::Child& c = done[(size_t)(BvhBuilder::kDeferred - n.c1)]; n.c1 = c.index; n.n1 = c.leaf ? c.count : 0; }
        }
    }
    if (b.max_depth > kBvhMaxDepth) return -2; /* cannot happen either (see the split rule) */
    /* leaf indices are relative to this mesh: make them global, and permute the triangles */
    std::vector<DTri> sorted(count);
    for (uint32_t i = 0; i < count; ++i) { sorted[i] = out.tris[first + b.order[i]]; out.tri_orig[first + i] = b.order[i]; }
    std::copy(sorted.begin(), sorted.end(), out.tris.begin() + first);
    for (size_t k = (size_t)root; k < out.bvh.size(); ++k) {
        DBvhNode& n = out.bvh[k];
        if (n.n0 > 0) n.c0 += (int32_t)first;
        if (n.n1 > 0) n.c1 += (int32_t)first;
    }
    return root;
}

}  // namespace

bool verify_bvh(const Built& b, int& max_depth, std::string& err) {
    max_depth = 0;
    std::vector<uint8_t> seen(b.tris.size(), 0);
    struct Item { int32_t node; int depth; float lo[3], hi[3]; };
    for (size_t mi = 0; mi < b.meshes.size(); ++mi) {
        const DMesh& m = b.meshes[mi];
        if (m.bvh_root < 0) continue;
        std::vector<Item> todo;
        Item root{m.bvh_root, 1, {m.lo[0], m.lo[1], m.lo[2]}, {m.hi[0], m.hi[1], m.hi[2]}};
        todo.push_back(root);
        size_t leaves_tris = 0;
        while (!todo.empty()) {
            Item it = todo.back(); todo.pop_back();
            if (it.node < 0 || (size_t)it.node >= b.bvh.size()) return fail(err, "bvh: node index out of range");
            if (it.depth > kBvhMaxDepth) return fail(err, "bvh: path longer than the traversal stack");
            max_depth = std::max(max_depth, it.depth);
            const DBvhNode& n = b.bvh[(size_t)it.node];
            const float* lo[2] = {n.lo0, n.lo1}; const float* hi[2] = {n.hi0, n.hi1};
            const int32_t c[2] = {n.c0, n.c1}, cnt[2] = {n.n0, n.n1};
            for (int k = 0; k < 2; ++k) {
                if (cnt[k] == 0 && c[k] < 0) continue; /* empty child */
                for (int a = 0; a < 3; ++a)
                    if (lo[k][a] < it.lo[a] || hi[k][a] > it.hi[a]) return fail(err, "bvh: child box sticks out of its parent");
                if (cnt[k] == 0) {
                    Item ch{c[k], it.depth + 1, {lo[k][0], lo[k][1], lo[k][2]}, {hi[k][0], hi[k][1], hi[k][2]}};
                    todo.push_back(ch);
                    continue;
                }
                if (cnt[k] < 1 || cnt[k] > 4) return fail(err, "bvh: leaf size outside 1..4");
                for (int32_t t = c[k]; t < c[k] + cnt[k]; ++t) {
                    if (t < (int32_t)m.first_tri || t >= (int32_t)(m.first_tri + m.n_tris)) return fail(err, "bvh: leaf triangle outside the mesh");
                    if (seen[(size_t)t]++) return fail(err, "bvh: triangle in two leaves");
                    ++leaves_tris;
                    const DTri& tr = b.tris[(size_t)t];
                    for (int a = 0; a < 3; ++a) {
                        float v[3] = {tr.v0[a], tr.v0[a] + tr.e1[a], tr.v0[a] + tr.e2[a]};
                        for (float x : v)
                            if (x < lo[k][a] || x > hi[k][a]) return fail(err, "bvh: triangle vertex outside its leaf box");
                    }
                }
            }
        }
        if (leaves_tris != m.n_tris) return fail(err, "bvh: a triangle is in no leaf");
        /* tri_orig is a permutation of the mesh-local indices */
        std::vector<uint8_t> hit(m.n_tris, 0);
        for (uint32_t t = 0; t < m.n_tris; ++t) {
            uint32_t o = b.tri_orig[m.first_tri + t];
            if (o >= m.n_tris || hit[o]++) return fail(err, "bvh: tri_orig is not a permutation");
        }
    }
    return true;
}

bool verify_bvh4(const Built& b, const std::vector<DBvh4Node>& bvh4, int& max_stack, std::string& err) {
    max_stack = 0;
    std::vector<uint8_t> seen(b.tris.size(), 0);
    struct Item { int32_t node; int depth, pending; float lo[3], hi[3]; };
    for (size_t mi = 0; mi < b.meshes.size(); ++mi) {
        const DMesh& m = b.meshes[mi];
        if (m.bvh_root < 0) continue;
        if (m.bvh4_root < 0) return fail(err, "bvh4: a mesh with a binary tree has no four-wide tree");
        std::vector<Item> todo;
        Item root{m.bvh4_root, 1, 0, {m.lo[0], m.lo[1], m.lo[2]}, {m.hi[0], m.hi[1], m.hi[2]}};
        todo.push_back(root);
        size_t leaves_tris = 0;
        while (!todo.empty()) {
            Item it = todo.back(); todo.pop_back();
            if (it.node < 0 || (size_t)it.node >= bvh4.size()) return fail(err, "bvh4: node index out of range");
            if (it.depth > kBvhMaxDepth / 2 + 1) return fail(err, "bvh4: path longer than half the binary depth limit");
            const DBvh4Node& n = bvh4[(size_t)it.node];
            const float* lo[4] = {n.half[0].lo0, n.half[0].lo1, n.half[1].lo0, n.half[1].lo1};
            const float* hi[4] = {n.half[0].hi0, n.half[0].hi1, n.half[1].hi0, n.half[1].hi1};
            const int32_t c[4] = {n.half[0].c0, n.half[0].c1, n.half[1].c0, n.half[1].c1};
            const int32_t cnt[4] = {n.half[0].n0, n.half[0].n1, n.half[1].n0, n.half[1].n1};
            int used = 0;
            for (int k = 0; k < 4; ++k) used += (cnt[k] == 0 && c[k] < 0) ? 0 : 1;
            if (used == 0) return fail(err, "bvh4: node without children");
            /* the traversal leaves at most used - 1 entries on its stack per node of the path */
            const int pending = it.pending + used - 1;
            max_stack = std::max(max_stack, pending);
            for (int k = 0; k < 4; ++k) {
                if (cnt[k] == 0 && c[k] < 0) continue;
                for (int a = 0; a < 3; ++a)
                    if (lo[k][a] < it.lo[a] || hi[k][a] > it.hi[a]) return fail(err, "bvh4: child box sticks out of its parent");
                if (cnt[k] == 0) {
                    Item ch{c[k], it.depth + 1, pending, {lo[k][0], lo[k][1], lo[k][2]}, {hi[k][0], hi[k][1], hi[k][2]}};
                    todo.push_back(ch);
                    continue;
                }
                if (cnt[k] < 1 || cnt[k] > 4) return fail(err, "bvh4: leaf size outside 1..4");
                for (int32_t t = c[k]; t < c[k] + cnt[k]; ++t) {
                    if (t < (int32_t)m.first_tri || t >= (int32_t)(m.first_tri + m.n_tris)) return fail(err, "bvh4: leaf triangle outside the mesh");
                    if (seen[(size_t)t]++) return fail(err, "bvh4: triangle in two leaves");
                    ++leaves_tris;
                    const DTri& tr = b.tris[(size_t)t];
                    for (int a = 0; a < 3; ++a) {
                        float v[3] = {tr.v0[a], tr.v0[a] + tr.e1[a], tr.v0[a] + tr.e2[a]};
                        for (float x : v)
                            if (x < lo[k][a] || x > hi[k][a]) return fail(err, "bvh4: triangle vertex outside its leaf box");
                    }
                }
            }
        }
        if (leaves_tris != m.n_tris) return fail(err, "bvh4: a triangle is in no leaf");
    }
    if (max_stack > kBvhMaxDepth + kBvhMaxDepth / 2) return fail(err, "bvh4: a path needs more stack entries than the traversal has");
    return true;
}

bool verify_bvh4q(const std::vector<DBvh4Node>& bvh4, const std::vector<DBvh4Q>& quant, std::string& err) {
    if (bvh4.size() != quant.size()) return fail(err, "bvh4q: table sizes differ");
    for (size_t i = 0; i < bvh4.size(); ++i) {
        const DBvh4Node& n = bvh4[i];
        const DBvh4Q& q = quant[i];
        const float* lo[4] = {n.half[0].lo0, n.half[0].lo1, n.half[1].lo0, n.half[1].lo1};
        const float* hi[4] = {n.half[0].hi0, n.half[0].hi1, n.half[1].hi0, n.half[1].hi1};
        const int32_t c[4] = {n.half[0].c0, n.half[0].c1, n.half[1].c0, n.half[1].c1};
        const int32_t cnt[4] = {n.half[0].n0, n.half[0].n1, n.half[1].n0, n.half[1].n1};
        for (int k = 0; k < 4; ++k) {
            const bool empty = cnt[k] == 0 && c[k] < 0;
            const int32_t want = empty ? RT_BVH4Q_EMPTY : (cnt[k] > 0 ? ~(c[k] * 8 + cnt[k]) : c[k]);
            if (q.child[k] != want) return fail(err, "bvh4q: child code differs from the full-precision node");
            if (empty) continue;
            for (int a = 0; a < 3; ++a) {
                const float ql = (float)((q.q[a] >> (8 * k)) & 255u), qh = (float)((q.q[3 + a] >> (8 * k)) & 255u);
                const float dlo = std::fmaf(ql, q.scale[a], q.lo[a]), dhi = std::fmaf(qh, q.scale[a], q.lo[a]);
                if (!(dlo <= lo[k][a]) || !(dhi >= hi[k][a])) return fail(err, "bvh4q: a quantised child box does not enclose the original");
            }
        }
    }
    return true;
}

bool build_scene(const rt_scene_desc& d, Built& out, std::string& err, bool defer_meshes) {
    out = Built();
    if ((d.n_spheres && !d.spheres) || (d.n_squares && !d.squares) || (d.n_meshes && !d.meshes) ||
        (d.n_lights && !d.lights) || (d.n_materials && !d.materials))
        return fail(err, "rt_scene_create: null array with non-zero count");
    if (d.n_vertices && (!d.positions || !d.normals)) return fail(err, "rt_scene_create: null vertex arrays");
    if (d.n_triangles && !d.triangles) return fail(err, "rt_scene_create: null triangle array");
    auto mat_ok = [&](int32_t m) { return m >= 0 && (uint32_t)m < d.n_materials; };

    for (uint32_t i = 0; i < d.n_materials; ++i) {
        const rt_material& m = d.materials[i];
        DMaterial o{};
        for (int a = 0; a < 3; ++a) { o.kd[a] = m.diffuse[a]; o.ks[a] = m.specular[a]; }
        o.shininess = m.shininess; o.ior = m.index_medium; o.type = m.type;
        out.materials.push_back(o);
    }
    for (uint32_t i = 0; i < d.n_spheres; ++i) {
        const rt_sphere& s = d.spheres[i];
        if (!mat_ok(s.material)) return fail(err, "rt_scene_create: sphere material index out of range");
        DSphere o{};
        for (int a = 0; a < 3; ++a) o.c[a] = s.center[a];
        o.r = s.radius;
        o.r2 = s.radius * s.radius;
        o.mat = s.material;
        out.spheres.push_back(o);
    }
    for (uint32_t i = 0; i < d.n_squares; ++i) {
        const rt_square& q = d.squares[i];
        if (!mat_ok(q.material)) return fail(err, "rt_scene_create: square material index out of range");
        DSquare o{};
        F3 v0 = ld(q.v[0]), w = ld(q.v[1]) - v0, h = ld(q.v[3]) - v0, n = cross(w, h);
        st(o.v0, v0); st(o.w, w); st(o.h, h); st(o.n, n);
        edge_interval(len(w), o.w_lo, o.w_hi);
        edge_interval(len(h), o.h_lo, o.h_hi);
        float L = len(n);
        o.nn[0] = n.x / L; o.nn[1] = n.y / L; o.nn[2] = n.z / L;
        o.mat = q.material;
        out.squares.push_back(o);
    }
    const float scaling = 1.000001f;  /* R/src/Mesh.h:220 */
    uint64_t deferred_tris = 0;
    for (uint32_t i = 0; i < d.n_meshes; ++i) {
        const rt_mesh& m = d.meshes[i];
        if (!mat_ok(m.material)) return fail(err, "rt_scene_create: mesh material index out of range");
        if ((uint64_t)m.first_vertex + m.n_vertices > d.n_vertices || (uint64_t)m.first_triangle + m.n_triangles > d.n_triangles)
            return fail(err, "rt_scene_create: mesh slice outside the vertex / triangle pools");
        DMesh o{};
        o.mat = m.material; o.bvh_root = -1; o.bvh4_root = -1; o.pad1 = 0.f; o.n_tris = m.n_triangles;
        if (defer_meshes) { /* records, bounds and BVH are made on the device from the caller's arrays */
            o.first_tri = (uint32_t)deferred_tris;
            deferred_tris += m.n_triangles;
            out.meshes.push_back(o);
            continue;
        }
        o.first_tri = (uint32_t)out.tris.size();
        for (uint32_t t = 0; t < m.n_triangles; ++t) {
            const uint32_t* idx = d.triangles + 3 * ((size_t)m.first_triangle + t);
            if (idx[0] >= m.n_vertices || idx[1] >= m.n_vertices || idx[2] >= m.n_vertices)
                return fail(err, "rt_scene_create: triangle vertex index out of range");
            const float* P = d.positions + 3 * (size_t)m.first_vertex;
            F3 v0 = scaling * ld(P + 3 * idx[0]), v1 = scaling * ld(P + 3 * idx[1]), v2 = scaling * ld(P + 3 * idx[2]);
            DTri tr{};
            st(tr.v0, v0); st(tr.e1, v1 - v0); st(tr.e2, v2 - v0);
            tr.i0 = m.first_vertex + idx[0]; tr.i1 = m.first_vertex + idx[1]; tr.i2 = m.first_vertex + idx[2];
            out.tris.push_back(tr);
            out.tri_orig.push_back(t);
        }
        set_mesh_bounds(out, o.first_tri, o.n_tris, o);
        if (m.n_triangles > kBvhMinTris) {
            o.bvh_root = build_mesh_bvh(out, o.first_tri, o.n_tris, o);
            if (o.bvh_root < -1) return fail(err, "rt_scene_create: triangle BVH deeper than the traversal stack");
        }
        out.meshes.push_back(o);
    }
    if (deferred_tris > 0xffffffffull) return fail(err, "rt_scene_create: more than 2^32 triangles");
    if (!defer_meshes) out.normals.assign(d.normals, d.normals + 3 * (size_t)d.n_vertices);
    for (uint32_t i = 0; i < d.n_lights; ++i) {
        const rt_light& l = d.lights[i];
        DLight o{};
        o.type = l.type;
        for (int a = 0; a < 3; ++a) { o.pos[a] = l.pos[a]; o.color[a] = l.color[a]; o.p0[a] = l.p0[a]; o.p1[a] = l.p1[a]; o.p2[a] = l.p2[a]; }
        out.lights.push_back(o);
    }

    /* root box: Scene::computeGlobalAABB, R/src/Scene.h:133-166 */
    Box root;
    for (uint32_t i = 0; i < d.n_meshes; ++i)
        for (uint32_t v = 0; v < d.meshes[i].n_vertices; ++v) grow(root, d.positions + 3 * ((size_t)d.meshes[i].first_vertex + v));
    for (uint32_t i = 0; i < d.n_squares; ++i)
        for (int v = 0; v < 4; ++v) grow(root, d.squares[i].v[v]);
    for (uint32_t i = 0; i < d.n_spheres; ++i) {
        const rt_sphere& s = d.spheres[i];
        float lo[3] = {s.center[0] - s.radius, s.center[1] - s.radius, s.center[2] - s.radius};
        float hi[3] = {s.center[0] + s.radius, s.center[1] + s.radius, s.center[2] + s.radius};
        grow(root, lo); grow(root, hi);
    }
    for (int a = 0; a < 3; ++a) { out.box_lo[a] = root.lo[a]; out.box_hi[a] = root.hi[a]; }

    /* DLightSq per (quad light, square): see rt_types.h.  R bounds |X| for every point the rules are applied to (shadow origins
     * inside the root box, light samples, their 0.001 offsets); a float dot(v0 - X, n) or dot(X - v0, w) then carries a rounding
     * error below 1e-6 (|v0| + R) |n| resp. |w|, a light sample computed in floats lies within 1e-6 R of its exact place, and the
     * end of a shadow ray (two normalisations, one length) within 2e-6 R of sample + 0.001 N. */
    {
        double R = 0.0;
        for (int a = 0; a < 3; ++a) R = std::max(R, (double)std::max(std::fabs(root.lo[a]), std::fabs(root.hi[a])));
        for (const DLight& l : out.lights)
            for (const float* c : {l.p0, l.p1, l.p2, l.pos})
                for (int a = 0; a < 3; ++a) R = std::max(R, (double)std::fabs(c[a]));
        R = std::sqrt(3.0) * R + 0.002;
        out.cert_radius = std::isfinite(R) ? (float)R : 0.f;
        const float nan = std::numeric_limits<float>::quiet_NaN();
        for (const DLight& l : out.lights)
            for (const DSquare& q : out.squares) {
                DLightSq r;
                float* f = &r.h0;
                for (int i = 0; i < 20; ++i) f[i] = nan;
                const double nl = std::sqrt((double)q.n[0] * q.n[0] + (double)q.n[1] * q.n[1] + (double)q.n[2] * q.n[2]);
                const double v0l = std::sqrt((double)q.v0[0] * q.v0[0] + (double)q.v0[1] * q.v0[1] + (double)q.v0[2] * q.v0[2]);
                if (l.type == RT_LIGHT_QUAD && std::isfinite(R) && nl > 0.0 && std::isfinite(nl)) {
                    const float* corner[3] = {l.p0, l.p1, l.p2};
                    auto dotd = [](const float* a, const float* b, const float* c) { /* dot(a - b, c) in double */
                        return ((double)a[0] - b[0]) * c[0] + ((double)a[1] - b[1]) * c[1] + ((double)a[2] - b[2]) * c[2];
                    };
                    double g[3], pw[3], pk[3];
                    for (int i = 0; i < 3; ++i) { g[i] = dotd(q.v0, corner[i], q.n); pw[i] = dotd(corner[i], q.v0, q.w); pk[i] = dotd(corner[i], q.v0, q.h); }
                    const double wl = std::sqrt((double)q.w[0] * q.w[0] + (double)q.w[1] * q.w[1] + (double)q.w[2] * q.w[2]);
                    const double kl = std::sqrt((double)q.h[0] * q.h[0] + (double)q.h[1] * q.h[1] + (double)q.h[2] * q.h[2]);
                    const double a_max = (v0l + R) * nl;                 /* bound of any |numerator| */
                    const double delta = 8e-6 * a_max;                   /* rounding of numerators and of g evaluated in floats, sample and ray-end placement */
                    r.h0 = (float)g[0]; r.e1 = (float)(g[1] - g[0]); r.e2 = (float)(g[2] - g[0]);
                    r.g_min = (float)std::min(g[0], std::min(g[1], g[2])); r.g_max = (float)std::max(g[0], std::max(g[1], g[2]));
                    r.a_sign = (float)(2.0 * delta);
                    r.t_miss = (float)(0.001001 * nl + 2e-5 * a_max + 2.0 * delta);
                    r.t_hit = r.t_miss; /* |g| beyond the same margin on the other side: 0 < t < |S - P| (t > 0 needs no margin of its own) */
                    /* the accepted intervals of the rectangle test, shrunk by what the projections can be off: rounding for the
                     * origin's own (ow / ok), rounding plus the 0.001 N between a sample and the end of its ray for the sample's */
                    const double mo_w = 1e-5 * (v0l + R) * wl, mo_k = 1e-5 * (v0l + R) * kl;
                    const double ms_w = 0.001001 * wl + mo_w, ms_k = 0.001001 * kl + mo_k;
                    const double cw = 0.5 * ((double)q.w_lo + q.w_hi), ck = 0.5 * ((double)q.h_lo + q.h_hi);
                    const double half_w = 0.5 * ((double)q.w_hi - q.w_lo), half_k = 0.5 * ((double)q.h_hi - q.h_lo);
                    r.cw = (float)cw; r.ck = (float)ck;
                    r.w0 = (float)(pw[0] - cw); r.w1 = (float)(pw[1] - pw[0]); r.w2 = (float)(pw[2] - pw[0]);
                    r.k0 = (float)(pk[0] - ck); r.k1 = (float)(pk[1] - pk[0]); r.k2 = (float)(pk[2] - pk[0]);
                    r.rw = (float)(half_w > ms_w ? half_w - ms_w : -1.0); r.rk = (float)(half_k > ms_k ? half_k - ms_k : -1.0);
                    r.ow = (float)(half_w > mo_w ? half_w - mo_w : -1.0); r.ok = (float)(half_k > mo_k ? half_k - mo_k : -1.0);
                    for (int i = 0; i < 20; ++i) /* a record that is not finite decides nothing */
                        if (!std::isfinite(f[i])) { for (int j = 0; j < 20; ++j) f[j] = nan; break; }
                }
                out.light_sq.push_back(r);
            }
    }

    /* KDTree::buildKDTree, R/src/KDTree.cpp:477-519: spheres, squares, meshes */
    for (uint32_t i = 0; i < d.n_spheres; ++i) out.refs.push_back(DRef{RT_SPHERE, (int32_t)i});
    for (uint32_t i = 0; i < d.n_squares; ++i) out.refs.push_back(DRef{RT_SQUARE, (int32_t)i});
    for (uint32_t i = 0; i < d.n_meshes; ++i) out.refs.push_back(DRef{RT_MESH, (int32_t)i});
    build_node(d, out, root, 0, (int32_t)out.refs.size(), 0);
    if (!out.kd.empty() && !out.kd[0].leaf) { /* leaves of a real tree read their objects' records in reference order */
        DLeafRec none;
        std::memset(&none, 0, sizeof(none));
        out.leaf_recs.assign(out.refs.size(), none);
        for (size_t r = 0; r < out.refs.size(); ++r) {
            DLeafRec& rec = out.leaf_recs[r];
            rec.type = out.refs[r].type; rec.object = out.refs[r].object;
            if (rec.type == RT_SPHERE) rec.u.s = out.spheres[(size_t)rec.object];
            if (rec.type == RT_SQUARE) rec.u.q = out.squares[(size_t)rec.object];
        }
    }
    out.kd_epsilon = d.kd_epsilon;
    return true;
}

/* 4x4 inverse by cofactors in the reference's term order (R/src/matrixUtilities.h:12-122).
 * Each output element is six signed triple products; every index triple below is written as a
 * three-digit hex number ijk, negative terms carry bit 12. */
static const uint16_t kCofactor[16][6] = {
    {0x05af, 0x15be, 0x196f, 0x097e, 0x0d6b, 0x1d7a}, {0x11af, 0x01be, 0x092f, 0x193e, 0x1d2b, 0x0d3a},
    {0x016f, 0x117e, 0x152f, 0x053e, 0x0d27, 0x1d36}, {0x116b, 0x017a, 0x052b, 0x153a, 0x1927, 0x0936},
    {0x14af, 0x04be, 0x086f, 0x187e, 0x1c6b, 0x0c7a}, {0x00af, 0x10be, 0x182f, 0x083e, 0x0c2b, 0x1c3a},
    {0x106f, 0x007e, 0x042f, 0x143e, 0x1c27, 0x0c36}, {0x006b, 0x107a, 0x142b, 0x043a, 0x0827, 0x1836},
    {0x049f, 0x14bd, 0x185f, 0x087d, 0x0c5b, 0x1c79}, {0x109f, 0x00bd, 0x081f, 0x183d, 0x1c1b, 0x0c39},
    {0x005f, 0x107d, 0x141f, 0x043d, 0x0c17, 0x1c35}, {0x105b, 0x0079, 0x041b, 0x1439, 0x1817, 0x0835},
    {0x149e, 0x04ad, 0x085e, 0x186d, 0x1c5a, 0x0c69}, {0x009e, 0x10ad, 0x181e, 0x082d, 0x0c1a, 0x1c29},
    {0x105e, 0x006d, 0x041e, 0x142d, 0x1c16, 0x0c25}, {0x005a, 0x1069, 0x141a, 0x0429, 0x0816, 0x1825},
};

static bool invert4(const double m[16], double out[16]) {
    double inv[16];
    for (int o = 0; o < 16; ++o) {
        double acc = 0.0;
        for (int t = 0; t < 6; ++t) {
            uint16_t c = kCofactor[o][t];
            double p = m[(c >> 8) & 15] * m[(c >> 4) & 15] * m[c & 15];
            bool neg = (c & 0x1000) != 0;
            if (t == 0) acc = neg ? -p : p;
            else acc = neg ? acc - p : acc + p;
        }
        inv[o] = acc;
    }
    double det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    if (det == 0) return false;
    det = 1.0 / det;
    for (int i = 0; i < 16; ++i) out[i] = inv[i] * det;
    return true;
}

bool invert_camera(const rt_camera& cam, DCamera& out, std::string& err) {
    double p[16], mv[16];
    for (int i = 0; i < 16; ++i) { p[i] = cam.projection[i]; mv[i] = cam.modelview[i]; }
    if (!invert4(p, out.proj_inv) || !invert4(mv, out.mv_inv)) return fail(err, "camera matrix is singular");
    out.depth_near = cam.depth_near;
    /* cameraSpaceToWorldSpace(0,0,0), R/src/matrixUtilities.h:188-197 */
    const double* m = out.mv_inv;
    double r[4] = {m[0] * 0.0 + m[4] * 0.0 + m[8] * 0.0 + m[12] * 1.0, m[1] * 0.0 + m[5] * 0.0 + m[9] * 0.0 + m[13] * 1.0,
                   m[2] * 0.0 + m[6] * 0.0 + m[10] * 0.0 + m[14] * 1.0, m[3] * 0.0 + m[7] * 0.0 + m[11] * 0.0 + m[15] * 1.0};
    for (int a = 0; a < 3; ++a) out.eye[a] = (float)(r[a] / r[3]);
    out.pad = 0.f;
    return true;
}

}  // namespace rtb
