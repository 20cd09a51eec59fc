/*
 * Scene.hpp -- the reference's Scene interface (R/src/Scene.h:26-1050) in front of the HIP path.
 *
 * The host keeps what the reference does on the host: the object lists, the lights, the four
 * setup_* builders and the kd-tree parameters.  buildKDTree() flattens the scene
 * (rt_scene_desc) and hands it to rt_scene_create, which builds the object kd-tree and uploads
 * everything; computeIntersection() and rayTrace() are answered by the GPU.  The batch forms
 * (computeIntersections / rayTraceBatch / render) are the efficient entry points; the single-ray
 * forms exist so existing call sites keep compiling.
 */
#pragma once
#include <cfloat>
#include <iostream>
#include <limits>
#include <string>
#include <vector>
#include "Device.hpp"
#include "Mesh.hpp"
#include "Sphere.hpp"
#include "Square.hpp"

const int MESH = 0;
const int SPHERE = 1;
const int SQUARE = 2;

enum LightType { LightType_Spherical, LightType_Quad };

struct Light {
    Vec3 material;
    bool isInCamSpace = false;
    LightType type = LightType_Spherical;
    Vec3 pos;
    float radius = 0.f;
    Mesh quad;
    float powerCorrection = 1.0f;
};

/* R/src/Scene.h:49-69: a width x height quad in the XY plane around light.pos */
inline void initialize_quad_light(Light& light, float width, float height) {
    if (light.type != LightType_Quad) {
        std::cerr << "La lumière doit être de type Quad pour utiliser cette fonction." << std::endl;
        return;
    }
    const float hw = width * 0.5f, hh = height * 0.5f;
    const float sx[4] = {-hw, hw, hw, -hw}, sy[4] = {-hh, -hh, hh, hh};
    light.quad.vertices.clear();
    for (int i = 0; i < 4; ++i) light.quad.vertices.push_back(MeshVertex(light.pos + Vec3(sx[i], sy[i], 0), Vec3(0, 0, 1)));
    light.quad.triangles = {MeshTriangle(0, 1, 2), MeshTriangle(0, 2, 3)};
    light.quad.build_arrays();
}

struct RaySceneIntersection { /* R/src/Scene.h:72-81 */
    bool intersectionExists;
    unsigned int typeOfIntersectedObject;
    unsigned int objectIndex;
    float t;
    RayTriangleIntersection rayMeshIntersection;
    RaySphereIntersection raySphereIntersection;
    RaySquareIntersection raySquareIntersection;
    RaySceneIntersection() : intersectionExists(false), typeOfIntersectedObject(0), objectIndex(0), t(FLT_MAX) {}
};

/* kd-tree parameters (R/src/KDTree.h:31-58); the tree itself is built inside rt_scene_create */
class KDTree {
    int maxDepth = 15, maxPrimitivesPerLeaf = 30;
    float epsilon = 0.01f;

public:
    void setMaxDepth(int d) { maxDepth = d; }
    int getMaxDepth() const { return maxDepth; }
    void setMaxPrimitivesPerLeaf(int n) { maxPrimitivesPerLeaf = n; }
    int getMaxPrimitivesPerLeaf() const { return maxPrimitivesPerLeaf; }
    void setEpsilon(float e) { epsilon = e; }
    float getEpsilon() const { return epsilon; }
};

/* Flat copy of a Scene: owns the arrays an rt_scene_desc points into. */
struct FlatScene {
    std::vector<rt_sphere> spheres;
    std::vector<rt_square> squares;
    std::vector<rt_mesh> meshes;
    std::vector<rt_light> lights;
    std::vector<rt_material> materials;
    std::vector<float> positions, normals;
    std::vector<uint32_t> triangles;
    rt_scene_desc desc{};
};

class Scene {
    std::vector<Mesh> meshes;
    std::vector<Sphere> spheres;
    std::vector<Square> squares;
    std::vector<Light> lights;
    KDTree kdtree;
    AABB aabbGlobal;
    rt_scene_t device = 0; /* 0 = not uploaded */
    uint32_t ctx_pixel = 0, ctx_sample = 0, ctx_seed = 1, ctx_k0 = 0;

    static int push_material(FlatScene& f, const Material& m) {
        rt_material o{};
        for (unsigned a = 0; a < 3; ++a) { o.diffuse[a] = m.diffuse_material[a]; o.specular[a] = m.specular_material[a]; }
        o.shininess = (float)m.shininess; /* the narrowing of Scene::getMaterialProperties */
        o.index_medium = m.index_medium;
        o.type = (int32_t)m.type;
        f.materials.push_back(o);
        return (int)f.materials.size() - 1;
    }
    void drop_device() {
        if (device) { rt_scene_destroy(device); device = 0; }
    }
    void reset_lists() { meshes.clear(); spheres.clear(); squares.clear(); lights.clear(); }
    void finish_setup() { aabbGlobal = computeGlobalAABB(); buildKDTree(); }

    Light& add_light(LightType type, Vec3 pos) {
        lights.resize(lights.size() + 1);
        Light& l = lights.back();
        l.pos = pos; l.type = type; l.material = Vec3(1, 1, 1); l.isInCamSpace = false;
        return l;
    }
    Square& add_unit_square(Vec3 bottom_left, Vec3 right) {
        squares.resize(squares.size() + 1);
        Square& s = squares.back();
        s.setQuad(bottom_left, right, Vec3(0., 1, 0.), 2., 2.);
        return s;
    }
    static void paint(Mesh& m, Vec3 kd, Vec3 ks, double shininess) {
        m.material.diffuse_material = kd; m.material.specular_material = ks; m.material.shininess = shininess;
    }
    /* the five walls shared by both Cornell scenes (R/src/Scene.h:828-891, 951-1014); each entry is
     * {scale first?, rotation axis (0 none, 1 x, 2 y), angle, diffuse, specular} */
    void add_cornell_walls() {
        struct Wall { bool scale_first; int axis; float angle; Vec3 kd, ks; };
        const Wall walls[5] = {
            {true, 0, 0.f, Vec3(1., 0.5, 0.), Vec3(1., 1., 1.)},      /* back    */
            {true, 2, 90.f, Vec3(1., 0., 0.), Vec3(1., 0., 0.)},      /* left    */
            {false, 2, -90.f, Vec3(0.0, 1.0, 0.0), Vec3(0.0, 1.0, 0.0)}, /* right   */
            {false, 1, -90.f, Vec3(0.5, 0.5, 1.0), Vec3(1.0, 1.0, 1.0)}, /* floor   */
            {false, 1, 90.f, Vec3(0.0, 0., 1.0), Vec3(1.0, 1.0, 1.0)},   /* ceiling */
        };
        for (const Wall& w : walls) {
            Square& s = add_unit_square(Vec3(-1., -1., 0.), Vec3(1., 0, 0.));
            if (w.scale_first) { s.scale(Vec3(2., 2., 1.)); s.translate(Vec3(0., 0., -2.)); }
            else { s.translate(Vec3(0., 0., -2.)); s.scale(Vec3(2., 2., 1.)); }
            if (w.axis == 1) s.rotate_x(w.angle);
            if (w.axis == 2) s.rotate_y(w.angle);
            s.build_arrays();
            paint(s, w.kd, w.ks, 16);
        }
    }
    Sphere& add_sphere(Vec3 c, float r, MaterialType type, Vec3 kd, Vec3 ks, double shininess) {
        spheres.resize(spheres.size() + 1);
        Sphere& s = spheres.back();
        s.m_center = c; s.m_radius = r;
        s.build_arrays();
        s.material.type = type;
        paint(s, kd, ks, shininess);
        return s;
    }

public:
    Scene() {
        kdtree.setMaxDepth(15);
        kdtree.setMaxPrimitivesPerLeaf(30);
        kdtree.setEpsilon(0.01f);
        aabbGlobal = computeGlobalAABB();
    }
    /* a copy owns no device state; it uploads again on first use */
    Scene(const Scene& o)
        : meshes(o.meshes), spheres(o.spheres), squares(o.squares), lights(o.lights), kdtree(o.kdtree), aabbGlobal(o.aabbGlobal),
          ctx_pixel(o.ctx_pixel), ctx_sample(o.ctx_sample), ctx_seed(o.ctx_seed), ctx_k0(o.ctx_k0) {}
    Scene& operator=(const Scene& o) {
        if (this != &o) {
            drop_device();
            meshes = o.meshes; spheres = o.spheres; squares = o.squares; lights = o.lights; kdtree = o.kdtree; aabbGlobal = o.aabbGlobal;
            ctx_pixel = o.ctx_pixel; ctx_sample = o.ctx_sample; ctx_seed = o.ctx_seed; ctx_k0 = o.ctx_k0;
        }
        return *this;
    }
    ~Scene() { drop_device(); }

    const std::vector<Mesh>& getMeshes() const { return meshes; }
    const std::vector<Sphere>& getSpheres() const { return spheres; }
    const std::vector<Square>& getSquares() const { return squares; }
    const std::vector<Light>& getLights() const { return lights; }
    std::vector<Mesh>& getMeshes() { return meshes; }
    std::vector<Sphere>& getSpheres() { return spheres; }
    std::vector<Square>& getSquares() { return squares; }
    std::vector<Light>& getLights() { return lights; }
    KDTree& getKDTree() { return kdtree; }

    /* R/src/Scene.h:133-166 */
    AABB computeGlobalAABB() const {
        AABB box;
        for (const Mesh& m : meshes) for (const MeshVertex& v : m.vertices) box.expandByPoint(v.position);
        for (const Square& s : squares) for (const MeshVertex& v : s.vertices) box.expandByPoint(v.position);
        for (const Sphere& sp : spheres) {
            box.expandByPoint(Vec3(sp.m_center[0] - sp.m_radius, sp.m_center[1] - sp.m_radius, sp.m_center[2] - sp.m_radius));
            box.expandByPoint(Vec3(sp.m_center[0] + sp.m_radius, sp.m_center[1] + sp.m_radius, sp.m_center[2] + sp.m_radius));
        }
        return box;
    }

    /* The state of this Scene as the C-ABI wants it (materials: spheres, squares, meshes in order). */
    void flatten(FlatScene& f) const {
        f = FlatScene();
        for (const Sphere& s : spheres) {
            rt_sphere o{};
            for (unsigned a = 0; a < 3; ++a) o.center[a] = s.m_center[a];
            o.radius = s.m_radius;
            o.material = push_material(f, s.material);
            f.spheres.push_back(o);
        }
        for (const Square& s : squares) {
            rt_square o{};
            for (unsigned v = 0; v < 4; ++v) for (unsigned a = 0; a < 3; ++a) o.v[v][a] = s.vertices[v].position[a];
            o.material = push_material(f, s.material);
            f.squares.push_back(o);
        }
        for (const Mesh& m : meshes) {
            rt_mesh o{};
            o.first_vertex = (uint32_t)(f.positions.size() / 3); o.n_vertices = (uint32_t)m.vertices.size();
            o.first_triangle = (uint32_t)(f.triangles.size() / 3); o.n_triangles = (uint32_t)m.triangles.size();
            o.material = push_material(f, m.material);
            for (unsigned a = 0; a < 3; ++a) { o.aabb_min[a] = m.aabb.minCoor[a]; o.aabb_max[a] = m.aabb.maxCoor[a]; }
            for (const MeshVertex& v : m.vertices) for (unsigned a = 0; a < 3; ++a) { f.positions.push_back(v.position[a]); f.normals.push_back(v.normal[a]); }
            for (const MeshTriangle& t : m.triangles) for (unsigned a = 0; a < 3; ++a) f.triangles.push_back(t.v[a]);
            f.meshes.push_back(o);
        }
        for (const Light& l : lights) {
            rt_light o{};
            o.type = l.type == LightType_Quad ? RT_LIGHT_QUAD : RT_LIGHT_SPHERICAL;
            for (unsigned a = 0; a < 3; ++a) { o.pos[a] = l.pos[a]; o.color[a] = l.material[a]; }
            if (l.type == LightType_Quad && l.quad.triangles.size() >= 1 && l.quad.vertices.size() >= 3) {
                const MeshTriangle& t = l.quad.triangles[0]; /* only triangle (0,1,2) is ever sampled */
                for (unsigned a = 0; a < 3; ++a) {
                    o.p0[a] = l.quad.vertices[t[0]].position[a]; o.p1[a] = l.quad.vertices[t[1]].position[a]; o.p2[a] = l.quad.vertices[t[2]].position[a];
                }
            }
            f.lights.push_back(o);
        }
        rt_scene_desc& d = f.desc;
        d.n_spheres = (uint32_t)f.spheres.size(); d.spheres = f.spheres.data();
        d.n_squares = (uint32_t)f.squares.size(); d.squares = f.squares.data();
        d.n_meshes = (uint32_t)f.meshes.size(); d.meshes = f.meshes.data();
        d.n_lights = (uint32_t)f.lights.size(); d.lights = f.lights.data();
        d.n_materials = (uint32_t)f.materials.size(); d.materials = f.materials.data();
        d.n_vertices = (uint32_t)(f.positions.size() / 3); d.n_triangles = (uint32_t)(f.triangles.size() / 3);
        d.positions = f.positions.data(); d.normals = f.normals.data(); d.triangles = f.triangles.data();
        d.kd_max_depth = kdtree.getMaxDepth(); d.kd_max_leaf = kdtree.getMaxPrimitivesPerLeaf(); d.kd_epsilon = kdtree.getEpsilon();
    }

    /* Scene::buildKDTree, R/src/Scene.h:112-115.  The tree is built inside rt_scene_create when the
     * scene is uploaded; this call only invalidates the uploaded copy, and the next query uploads the
     * current lists (so scene setup itself needs no GPU). */
    void buildKDTree() { drop_device(); }
    rt_scene_t deviceHandle() {
        if (!device) {
            rthost::ensure_device();
            FlatScene f;
            flatten(f);
            rthost::require(rt_scene_create(&f.desc, &device), "rt_scene_create");
        }
        return device;
    }

    /* The meshes moved (Mesh::translate / scale / rotate_* / a new vertex array with the same triangles, then
     * recomputeNormals as needed) after the scene was uploaded: sends their vertices again and refits the uploaded copy
     * (rt_scene_refit) instead of making it anew.  Returns false -- and drops the uploaded copy, so that the next query
     * makes a new one as buildKDTree() would -- when the scene cannot be refitted (more objects than a kd leaf holds,
     * another vertex count).  No counterpart in the reference, whose scenes do not move. */
    bool refitMeshes() {
        if (!device) return false;
        std::vector<float> pos, nrm;
        for (const Mesh& m : meshes)
            for (const MeshVertex& v : m.vertices) for (unsigned a = 0; a < 3; ++a) { pos.push_back(v.position[a]); nrm.push_back(v.normal[a]); }
        if (rt_scene_refit(device, pos.data(), nrm.data(), (uint32_t)(pos.size() / 3), nullptr) == RT_OK) return true;
        drop_device();
        return false;
    }

    /* ---- batch entry points ---------------------------------------------------------------- */
    void computeIntersections(const Ray* rays, size_t n, rt_hit* out) {
        std::vector<float> org(3 * n), dir(3 * n);
        for (size_t i = 0; i < n; ++i) for (unsigned a = 0; a < 3; ++a) { org[3 * i + a] = rays[i].origin()[a]; dir[3 * i + a] = rays[i].direction()[a]; }
        rthost::require(rt_intersect(deviceHandle(), org.data(), dir.data(), n, out), "rt_intersect");
    }
    /* ray i is sample (pixel[i], sample[i]) of the counter RNG stream with `seed`; its random draws start at k0 */
    void rayTraceBatch(const Ray* rays, size_t n, const uint32_t* pixel, const uint32_t* sample, uint32_t seed, uint32_t k0, Vec3* out,
                       rt_stats* stats = nullptr) {
        std::vector<float> org(3 * n), dir(3 * n), rgb(3 * n);
        for (size_t i = 0; i < n; ++i) for (unsigned a = 0; a < 3; ++a) { org[3 * i + a] = rays[i].origin()[a]; dir[3 * i + a] = rays[i].direction()[a]; }
        rthost::require(rt_trace(deviceHandle(), org.data(), dir.data(), n, pixel, sample, seed, k0, rgb.data(), stats), "rt_trace");
        for (size_t i = 0; i < n; ++i) out[i] = Vec3(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]);
    }
    /* whole image, row-major w*h Vec3 (R/main.cpp:182-194 without the loop) */
    /* With a communicator (rt_comm_init, see rt_main --gpus) this is a collective call: every process renders its tiles
     * and rank 0 receives the image; the other ranks get an empty one. */
    /* edge of the square image tiles that are dealt out to the ranks of a multi-GPU render: 8 evens out the expensive pixels of a
     * glass sphere over 8 ranks better than 16 (shares within 2 % instead of 4 % on the Cornell frame); one GPU does not care */
    uint32_t tile_size = 8;
    void render(const rt_camera& cam, unsigned w, unsigned h, unsigned spp, uint32_t seed, std::vector<Vec3>& image, rt_stats* stats = nullptr) {
        const uint32_t tw = tile_size, th = tile_size;
        const bool root = rt_comm_rank() == 0;
        std::vector<float> img(root ? (size_t)w * h * 3 : 0);
        rthost::require(rt_render_sharded(deviceHandle(), &cam, w, h, spp, seed, tw, th, root ? img.data() : nullptr, stats), "rt_render_sharded");
        if (!root) { image.clear(); return; }
        image.resize((size_t)w * h);
        for (size_t i = 0; i < image.size(); ++i) image[i] = Vec3(img[3 * i], img[3 * i + 1], img[3 * i + 2]);
    }

    /* ---- the reference's single-ray calls --------------------------------------------------- */
    /* Scene::computeIntersection, R/src/Scene.h:169-205: only t / position / normal of the matching
     * sub-struct are defined there; the mesh sub-struct additionally gets tIndex and the weights. */
    RaySceneIntersection computeIntersection(Ray const& ray) {
        rt_hit r;
        computeIntersections(&ray, 1, &r);
        RaySceneIntersection res;
        if (!r.hit) return res;
        res.intersectionExists = true;
        res.t = r.t;
        res.objectIndex = (unsigned)r.object;
        res.typeOfIntersectedObject = (unsigned)r.type;
        Vec3 P(r.p[0], r.p[1], r.p[2]), N(r.n[0], r.n[1], r.n[2]);
        if (r.type == SPHERE) { res.raySphereIntersection.intersectionExists = true; res.raySphereIntersection.intersection = P; res.raySphereIntersection.normal = N; res.raySphereIntersection.t = r.t; }
        else if (r.type == SQUARE) { res.raySquareIntersection.intersectionExists = true; res.raySquareIntersection.intersection = P; res.raySquareIntersection.normal = N; res.raySquareIntersection.t = r.t; }
        else {
            RayTriangleIntersection& m = res.rayMeshIntersection;
            m.intersectionExists = true; m.intersection = P; m.normal = N; m.t = r.t; m.tIndex = (unsigned)r.tri;
            m.w0 = 1.0f - r.u - r.v; m.w1 = r.u; m.w2 = r.v;
        }
        return res;
    }
    /* which counter-RNG sample the next rayTrace() calls belong to */
    void setSampleContext(uint32_t pixel, uint32_t sample, uint32_t seed = 1, uint32_t k0 = 0) { ctx_pixel = pixel; ctx_sample = sample; ctx_seed = seed; ctx_k0 = k0; }
    /* Scene::rayTrace, R/src/Scene.h:716-720 (10 bounces) */
    Vec3 rayTrace(Ray const& rayStart) {
        Vec3 c;
        rayTraceBatch(&rayStart, 1, &ctx_pixel, &ctx_sample, ctx_seed, ctx_k0, &c);
        return c;
    }

    /* ---- built-in scenes, R/src/Scene.h:722-1048 --------------------------------------------- */
    void setup_single_sphere() {
        reset_lists();
        Light& l = add_light(LightType_Spherical, Vec3(-5, 5, 5));
        l.radius = 2.5f; l.powerCorrection = 2.f;
        add_sphere(Vec3(1., 0., 0.), 1.f, Material_Mirror, Vec3(1., 0., 0.), Vec3(0.2, 0.2, 0.2), 20);
        add_sphere(Vec3(0., 0., 1.), 1.f, Material_Mirror, Vec3(0., 1., 0.), Vec3(0.2, 0.2, 0.2), 20);
        finish_setup();
    }
    void setup_single_square() {
        reset_lists();
        Light& l = add_light(LightType_Spherical, Vec3(-5, 5, 5));
        l.radius = 2.5f; l.powerCorrection = 2.f;
        { Square& s = add_unit_square(Vec3(-1., -1., 0.), Vec3(1., 0, 0.)); s.build_arrays(); paint(s, Vec3(1., 0., 0.), Vec3(0.8, 0.8, 0.8), 20); }
        { Square& s = add_unit_square(Vec3(-1.5, -1., 0.), Vec3(0.5, 0, 0.)); s.build_arrays(); paint(s, Vec3(0., 1., 0.), Vec3(0.8, 0.8, 0.8), 20); }
        finish_setup();
    }
    void setup_cornell_box() {
        reset_lists();
        initialize_quad_light(add_light(LightType_Quad, Vec3(0., 1.5, 0.)), 4.0f, 4.0f);
        add_cornell_walls();
        Sphere& mirror = add_sphere(Vec3(1.0, -1.25, 0.5), 0.75f, Material_Mirror, Vec3(1., 1., 1.), Vec3(1., 1., 1.), 16);
        mirror.material.transparency = 1.0; mirror.material.index_medium = 1.4;
        Sphere& glass = add_sphere(Vec3(-1.0, -1.25, -0.5), 0.75f, Material_Glass, Vec3(1., 1., 1.), Vec3(1., 1., 1.), 16);
        glass.material.transparency = 0.9; glass.material.index_medium = 1.5;
        finish_setup();
    }
    /* offFilePath defaults to the reference's relative path (R/src/Scene.h:899) */
    void setup_cornell_box_meshes(const std::string& offFilePath = "img/star.off") {
        reset_lists();
        initialize_quad_light(add_light(LightType_Quad, Vec3(0., 1.5, 0.)), 4.0f, 4.0f);
        add_cornell_walls();
        meshes.resize(meshes.size() + 1);
        Mesh& mesh = meshes.back();
        mesh.loadOFF(offFilePath);
        mesh.centerAndScaleToUnit();
        mesh.recomputeNormals();
        float minY = std::numeric_limits<float>::max();
        for (const MeshVertex& v : mesh.vertices) if (v.position[1] < minY) minY = v.position[1];
        Vec3 translation(0.0, -minY - 1.8, 0.0); /* rests the mesh 0.2 above the floor */
        for (MeshVertex& v : mesh.vertices) v.position += translation;
        std::cout << "Maillage chargé : " << offFilePath << std::endl;
        finish_setup();
    }
};
