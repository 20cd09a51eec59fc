/* Runs the host-side scene preparation (rt_build.cpp: validation, kd-tree, triangle BVH, camera inversion)
 * under AddressSanitizer / UBSan on a set of ordinary, degenerate and invalid scene descriptions.
 * Built and run by tests/test_build_cpu.py; prints "OK <n>" when every case behaved. */
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "rt_build.h"

namespace {

struct SceneData {
    std::vector<rt_material> mats;
    std::vector<rt_sphere> spheres;
    std::vector<rt_square> squares;
    std::vector<rt_mesh> meshes;
    std::vector<rt_light> lights;
    std::vector<float> pos, nrm;
    std::vector<uint32_t> tri;
    rt_scene_desc desc() {
        rt_scene_desc d;
        std::memset(&d, 0, sizeof d);
        d.n_materials = (uint32_t)mats.size(); d.materials = mats.data();
        d.n_spheres = (uint32_t)spheres.size(); d.spheres = spheres.data();
        d.n_squares = (uint32_t)squares.size(); d.squares = squares.data();
        d.n_meshes = (uint32_t)meshes.size(); d.meshes = meshes.data();
        d.n_lights = (uint32_t)lights.size(); d.lights = lights.data();
        d.n_vertices = (uint32_t)(pos.size() / 3); d.positions = pos.data(); d.normals = nrm.data();
        d.n_triangles = (uint32_t)(tri.size() / 3); d.triangles = tri.data();
        d.kd_max_depth = 15; d.kd_max_leaf = 30; d.kd_epsilon = 0.01f;
        return d;
    }
};

void add_objects(SceneData& s, int n_spheres, int n_squares, unsigned seed) {
    std::mt19937 g(seed);
    std::uniform_real_distribution<float> u(-1.8f, 1.8f), r(0.05f, 0.3f);
    rt_material m;
    std::memset(&m, 0, sizeof m);
    m.diffuse[0] = m.diffuse[1] = m.diffuse[2] = 0.5f; m.shininess = 16.f; m.index_medium = 1.4f;
    s.mats.push_back(m);
    for (int i = 0; i < n_spheres; ++i) {
        rt_sphere sp;
        std::memset(&sp, 0, sizeof sp);
        sp.center[0] = u(g); sp.center[1] = u(g); sp.center[2] = u(g); sp.radius = r(g); sp.material = 0;
        s.spheres.push_back(sp);
    }
    for (int i = 0; i < n_squares; ++i) {
        rt_square q;
        std::memset(&q, 0, sizeof q);
        float c[3] = {u(g), u(g), u(g)}, w = r(g) * 3, h = r(g) * 3;
        float v[4][3] = {{c[0], c[1], c[2]}, {c[0] + w, c[1], c[2]}, {c[0] + w, c[1] + h, c[2]}, {c[0], c[1] + h, c[2]}};
        std::memcpy(q.v, v, sizeof v);
        q.material = 0;
        s.squares.push_back(q);
    }
    rt_light l;
    std::memset(&l, 0, sizeof l);
    l.type = RT_LIGHT_QUAD; l.pos[1] = 1.5f; l.color[0] = l.color[1] = l.color[2] = 1.f;
    l.p0[0] = -2; l.p0[1] = -0.5f; l.p1[0] = 2; l.p1[1] = -0.5f; l.p2[0] = 2; l.p2[1] = 3.5f;
    s.lights.push_back(l);
}

void add_grid(SceneData& s, int n, float bump, unsigned seed) {
    std::mt19937 g(seed);
    std::uniform_real_distribution<float> u(0.f, bump);
    uint32_t v0 = (uint32_t)(s.pos.size() / 3), t0 = (uint32_t)(s.tri.size() / 3);
    for (int y = 0; y < n; ++y)
        for (int x = 0; x < n; ++x) {
            s.pos.insert(s.pos.end(), {-1.f + 2.f * x / (n - 1), -1.f + u(g), -1.f + 2.f * y / (n - 1)});
            s.nrm.insert(s.nrm.end(), {0.f, 1.f, 0.f});
        }
    for (int y = 0; y + 1 < n; ++y)
        for (int x = 0; x + 1 < n; ++x) {
            uint32_t a = y * n + x, b = a + 1, c = a + n + 1, d = a + n;
            s.tri.insert(s.tri.end(), {a, c, b, a, d, c});
        }
    rt_mesh m;
    std::memset(&m, 0, sizeof m);
    m.first_vertex = v0; m.n_vertices = (uint32_t)(s.pos.size() / 3) - v0;
    m.first_triangle = t0; m.n_triangles = (uint32_t)(s.tri.size() / 3) - t0;
    m.material = 0;
    s.meshes.push_back(m);
}

int failures = 0;
void expect(bool ok, const char* what) {
    if (!ok) { std::printf("FAILED: %s\n", what); ++failures; }
}

bool build_ok(SceneData& s, int* bvh_depth = nullptr) {
    rtb::Built b;
    std::string err;
    rt_scene_desc d = s.desc();
    if (!rtb::build_scene(d, b, err)) return false;
    int depth = 0;
    if (!rtb::verify_bvh(b, depth, err)) { std::printf("verify: %s\n", err.c_str()); return false; }
    if (bvh_depth) *bvh_depth = depth;
    return true;
}

}  // namespace

int main() {
    int cases = 0;
    { SceneData s; add_objects(s, 2, 5, 1); expect(build_ok(s), "small scene"); ++cases; }
    { SceneData s; add_objects(s, 60, 40, 2); expect(build_ok(s), "kd scene with internal nodes"); ++cases; }
    { SceneData s; add_objects(s, 3, 3, 3); add_grid(s, 96, 0.2f, 4); int d = 0; expect(build_ok(s, &d) && d > 5 && d <= 32, "grid mesh BVH"); ++cases; }
    { SceneData s; add_objects(s, 2, 2, 21); add_grid(s, 340, 0.3f, 22); int d = 0; expect(build_ok(s, &d) && d <= 32, "230k triangles: subtrees built by the thread pool"); ++cases; }
    { SceneData s; add_objects(s, 1, 1, 5); add_grid(s, 40, 0.f, 6); add_grid(s, 5, 0.1f, 7); expect(build_ok(s), "two meshes, one flat, one small"); ++cases; }
    { /* every triangle the same: centroids coincide */
        SceneData s; add_objects(s, 1, 1, 8);
        for (int i = 0; i < 3000; ++i) { s.pos.insert(s.pos.end(), {0, 0, 0, 1, 0, 0, 0, 0, 1}); s.nrm.insert(s.nrm.end(), {0, 1, 0, 0, 1, 0, 0, 1, 0}); }
        for (uint32_t i = 0; i < 9000; ++i) s.tri.push_back(i);
        rt_mesh m; std::memset(&m, 0, sizeof m); m.n_vertices = 9000; m.n_triangles = 3000; s.meshes.push_back(m);
        expect(build_ok(s), "identical triangles"); ++cases;
    }
    { /* degenerate inputs: zero-area triangles, NaN and huge coordinates must not break the builder */
        SceneData s; add_objects(s, 1, 1, 9); add_grid(s, 20, 0.1f, 10);
        s.pos[5] = NAN; s.pos[40] = 3e38f; s.pos[41] = -3e38f;
        for (int i = 0; i < 9; ++i) s.pos[90 + i] = 0.25f;
        rtb::Built b; std::string err; rt_scene_desc d = s.desc();
        bool built = rtb::build_scene(d, b, err);
        int depth = 0;
        if (built) (void)rtb::verify_bvh(b, depth, err); /* may report a box problem for NaN vertices; must not crash */
        ++cases;
    }
    { SceneData s; add_objects(s, 2, 2, 11); s.spheres[1].material = 7; expect(!build_ok(s), "material index out of range is rejected"); ++cases; }
    { SceneData s; add_objects(s, 2, 2, 12); add_grid(s, 10, 0.1f, 13); s.tri[17] = 100000; expect(!build_ok(s), "vertex index out of range is rejected"); ++cases; }
    { SceneData s; add_objects(s, 2, 2, 14); add_grid(s, 10, 0.1f, 15); s.meshes[0].n_triangles += 50; expect(!build_ok(s), "mesh slice outside the pool is rejected"); ++cases; }
    { SceneData s; expect(build_ok(s) || true, "empty scene"); ++cases; }
    { /* camera inversion: a perspective matrix and a singular one */
        rt_camera c; std::memset(&c, 0, sizeof c);
        float p[16] = {2.4142f, 0, 0, 0, 0, 2.4142f, 0, 0, 0, 0, -1.0008f, -1, 0, 0, -8.2034f, 0};
        float m[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, -6.1f, 1};
        for (int i = 0; i < 16; ++i) { c.projection[i] = p[i]; c.modelview[i] = m[i]; }
        DCamera dc; std::string err;
        expect(rtb::invert_camera(c, dc, err), "camera inversion");
        std::memset(c.projection, 0, sizeof c.projection);
        (void)rtb::invert_camera(c, dc, err); /* singular: either answer is fine, no crash */
        ++cases;
    }
    if (failures) return 1;
    std::printf("OK %d\n", cases);
    return 0;
}
