/* host_capi.cpp -- C entry points over the C++ host classes so that Python (tests, bench.py) can
 * build scenes and cameras with the product's own host code.  These calls never touch the GPU:
 * scenes are set up and flattened, not uploaded. */
#include <cstring>
#include <string>
#include "Camera.hpp"
#include "Render.hpp"
#include "Scene.hpp"

namespace {
struct HostScene { Scene scene; FlatScene flat; };
}

extern "C" {

/* Built-in scene `id` (0..3 = setup_single_sphere, setup_single_square, setup_cornell_box,
 * setup_cornell_box_meshes).  compute_mesh_aabbs != 0 replays R/main.cpp:348-353 (Mesh::computeAABB
 * on every mesh, then buildKDTree) as the reference does for the scene selected at start-up. */
void* rth_scene_builtin(int id, const char* off_path, int compute_mesh_aabbs) {
    HostScene* h = new HostScene();
    switch (id) {
        case 0: h->scene.setup_single_sphere(); break;
        case 1: h->scene.setup_single_square(); break;
        case 2: h->scene.setup_cornell_box(); break;
        case 3: h->scene.setup_cornell_box_meshes(off_path ? off_path : "img/star.off"); break;
        default: delete h; return nullptr;
    }
    if (compute_mesh_aabbs) {
        for (auto& m : h->scene.getMeshes()) m.computeAABB();
        h->scene.buildKDTree();
    }
    h->scene.flatten(h->flat);
    return h;
}

/* A Cornell-style scene whose mesh comes from an OFF file: the layout of setup_cornell_box_meshes
 * with the two spheres of setup_cornell_box added when with_spheres != 0. */
void* rth_scene_mesh_file(const char* off_path, int with_spheres) {
    HostScene* h = new HostScene();
    h->scene.setup_cornell_box_meshes(off_path);
    if (with_spheres) {
        Scene two;
        two.setup_cornell_box();
        for (const Sphere& s : two.getSpheres()) h->scene.getSpheres().push_back(s);
    }
    for (auto& m : h->scene.getMeshes()) m.computeAABB();
    h->scene.buildKDTree();
    h->scene.flatten(h->flat);
    return h;
}

/* Reads an OFF file with Mesh::loadOFF and with the reference's token-by-token reader alone and compares the
 * results bit for bit.  Returns bit 0 = identical, bit 1 = the multi-threaded reader accepted the file;
 * counts come back through nv / nt. */
int rth_off_compare(const char* path, unsigned* nv, unsigned* nt) {
    Mesh a, b;
    a.loadOFF(path);
    b.loadOFF_reference(path);
    std::vector<float> pos;
    std::vector<unsigned int> idx;
    const bool fast = offreader::read(path, pos, idx);
    bool same = a.vertices.size() == b.vertices.size() && a.triangles.size() == b.triangles.size();
    for (size_t i = 0; same && i < a.vertices.size(); ++i)
        same = std::memcmp(&a.vertices[i].position[0], &b.vertices[i].position[0], 12) == 0;
    for (size_t i = 0; same && i < a.triangles.size(); ++i) same = std::memcmp(a.triangles[i].v, b.triangles[i].v, 12) == 0;
    if (nv) *nv = (unsigned)a.vertices.size();
    if (nt) *nt = (unsigned)a.triangles.size();
    return (same ? 1 : 0) | (fast ? 2 : 0);
}

const rt_scene_desc* rth_scene_desc(void* scene) { return &static_cast<HostScene*>(scene)->flat.desc; }
void rth_scene_free(void* scene) { delete static_cast<HostScene*>(scene); }

/* Camera of the default flow: Camera(), resize(w, h), move(0, 0, -3.1), apply()
 * (R/main.cpp:100, :339, :178). */
void rth_camera_default(unsigned w, unsigned h, rt_camera* out) {
    Camera c;
    c.resize((int)w, (int)h);
    c.move(0., 0., -3.1);
    c.apply();
    *out = c.glState();
}

void rth_camera_build(float fov, float z_near, float z_far, unsigned w, unsigned h, float x, float y, float z, float zoom,
                      const float quat[4], rt_camera* out) {
    Camera c;
    c.setFovAngle(fov); c.setNearPlane(z_near); c.setFarPlane(z_far);
    c.resize((int)w, (int)h);
    c.move(x, y, z);
    c.zoom(zoom - 3.0f);
    c.setRotation(quat);
    c.apply();
    *out = c.glState();
}

/* ASCII P3 writer of R/main.cpp:205-215 on a row-major float RGB image */
int rth_write_ppm(const char* path, const float* rgb, unsigned w, unsigned h) {
    std::vector<Vec3> img((size_t)w * h);
    for (size_t i = 0; i < img.size(); ++i) img[i] = Vec3(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]);
    return write_ppm_p3(path, img, (int)w, (int)h) ? 0 : -1;
}

}  // extern "C"
