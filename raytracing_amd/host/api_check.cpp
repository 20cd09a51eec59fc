/* api_check.cpp -- exercises the reference-style single-call interface of the host classes on the GPU
 * and prints the results as text, for tests/test_gpu_host_api.py to compare with the CPU oracle:
 *   Scene::computeIntersection(Ray), Scene::rayTrace(Ray) (with its counter-RNG sample context),
 *   Mesh::intersect(Ray), and the batch forms behind them.
 * usage: api_check <scene 0..3> <star.off> <rays.txt>     rays.txt: "ox oy oz dx dy dz pixel sample" per line
 * output per ray: hit type object t px py pz nx ny nz | r g b | mesh: hit tIndex t [| S hit t p n] [| Q hit t p n]   (hex floats) */
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>
#include "Scene.hpp"

int main(int argc, char** argv) {
    if (argc < 4) { std::fprintf(stderr, "usage: api_check scene off rays.txt\n"); return 2; }
    int sid = std::atoi(argv[1]);
    Scene scene;
    switch (sid) {
        case 0: scene.setup_single_sphere(); break;
        case 1: scene.setup_single_square(); break;
        case 2: scene.setup_cornell_box(); break;
        default: scene.setup_cornell_box_meshes(argv[2]); break;
    }
    std::ifstream in(argv[3]);
    std::string line;
    while (std::getline(in, line)) {
        std::istringstream ls(line);
        float o[3], d[3];
        unsigned pixel, sample;
        if (!(ls >> o[0] >> o[1] >> o[2] >> d[0] >> d[1] >> d[2] >> pixel >> sample)) continue;
        Ray ray(Vec3(o[0], o[1], o[2]), Vec3(d[0], d[1], d[2]));
        RaySceneIntersection h = scene.computeIntersection(ray);
        Vec3 P, N;
        if (h.intersectionExists) {
            if (h.typeOfIntersectedObject == (unsigned)SPHERE) { P = h.raySphereIntersection.intersection; N = h.raySphereIntersection.normal; }
            else if (h.typeOfIntersectedObject == (unsigned)SQUARE) { P = h.raySquareIntersection.intersection; N = h.raySquareIntersection.normal; }
            else { P = h.rayMeshIntersection.intersection; N = h.rayMeshIntersection.normal; }
        }
        scene.setSampleContext(pixel, sample, /*seed*/ 1, /*k0*/ 2);
        Vec3 c = scene.rayTrace(ray);
        std::printf("%d %d %d %a %a %a %a %a %a %a | %a %a %a", h.intersectionExists ? 1 : 0, h.intersectionExists ? (int)h.typeOfIntersectedObject : -1,
                    h.intersectionExists ? (int)h.objectIndex : -1, h.t, P[0], P[1], P[2], N[0], N[1], N[2], c[0], c[1], c[2]);
        if (!scene.getMeshes().empty()) {
            RayTriangleIntersection m = scene.getMeshes()[0].intersect(ray);
            std::printf(" | %d %d %a %a %a %a", m.intersectionExists ? 1 : 0, (int)m.tIndex, m.t, m.w0, m.w1, m.w2);
        }
        /* the objects' own routines: Sphere::intersect / Square::intersect of object 0 (prefix S / Q) */
        if (!scene.getSpheres().empty()) {
            RaySphereIntersection q = scene.getSpheres()[0].intersect(ray);
            std::printf(" | S %d %a %a %a %a %a %a %a", q.intersectionExists ? 1 : 0, q.t, q.intersection[0], q.intersection[1], q.intersection[2],
                        q.normal[0], q.normal[1], q.normal[2]);
        }
        if (!scene.getSquares().empty()) {
            RaySquareIntersection q = scene.getSquares()[0].intersect(ray);
            std::printf(" | Q %d %a %a %a %a %a %a %a", q.intersectionExists ? 1 : 0, q.t, q.intersection[0], q.intersection[1], q.intersection[2],
                        q.normal[0], q.normal[1], q.normal[2]);
        }
        std::printf("\n");
    }
    /* "refit" as a fifth argument: the first mesh is moved after the scene was uploaded, Scene::refitMeshes() sends the
     * vertices again, and every ray of the file must get the answer of a scene that was set up with the moved mesh */
    if (argc >= 5 && std::string(argv[4]) == "refit" && !scene.getMeshes().empty()) {
        const Vec3 by(0.25f, 0.125f, -0.0625f);
        scene.getMeshes()[0].translate(by);
        const bool refitted = scene.refitMeshes();
        Scene fresh;
        fresh.setup_cornell_box_meshes(argv[2]);
        fresh.getMeshes()[0].translate(by);
        fresh.buildKDTree();
        std::ifstream again(argv[3]);
        unsigned n = 0, differ = 0, on_mesh = 0;
        while (std::getline(again, line)) {
            std::istringstream ls(line);
            float o[3], d[3];
            if (!(ls >> o[0] >> o[1] >> o[2] >> d[0] >> d[1] >> d[2])) continue;
            Ray ray(Vec3(o[0], o[1], o[2]), Vec3(d[0], d[1], d[2]));
            RaySceneIntersection a = scene.computeIntersection(ray), b = fresh.computeIntersection(ray);
            ++n;
            if (a.intersectionExists && a.typeOfIntersectedObject == (unsigned)MESH) ++on_mesh;
            if (a.intersectionExists != b.intersectionExists ||
                (a.intersectionExists && (a.typeOfIntersectedObject != b.typeOfIntersectedObject || a.objectIndex != b.objectIndex ||
                                          std::memcmp(&a.t, &b.t, sizeof(float)) != 0)))
                ++differ;
        }
        std::printf("REFIT %s rays %u on_mesh %u differ %u\n", refitted ? "done" : "refused", n, on_mesh, differ);
    }
    rt_shutdown();
    return 0;
}
