/*
 * Mesh.hpp -- triangle mesh with the reference's interface (R/src/Mesh.h:22-282,
 * R/src/Mesh.cpp:5-69) minus the OpenGL preview (draw / GL arrays are kept as plain arrays).
 * The host keeps loading and preparing the mesh -- these passes fix the exact float bits of the
 * vertices and normals the device later reads; intersect() runs on the GPU.
 */
#pragma once
#include <cfloat>
#include <fstream>
#include <limits>
#include <string>
#include <vector>
#include "AABB.hpp"
#include "Device.hpp"
#include "Material.hpp"
#include "OffReader.hpp"
#include "Ray.hpp"

struct MeshVertex {
    MeshVertex() : u(0), v(0) {}
    MeshVertex(const Vec3& p, const Vec3& n) : position(p), normal(n), u(0), v(0) {}
    Vec3 position, normal;
    float u, v;
};

struct MeshTriangle {
    MeshTriangle() : v{0, 0, 0} {}
    MeshTriangle(unsigned int v0, unsigned int v1, unsigned int v2) : v{v0, v1, v2} {}
    unsigned int& operator[](unsigned int i) { return v[i]; }
    unsigned int operator[](unsigned int i) const { return v[i]; }
    unsigned int v[3];
};

struct RayTriangleIntersection { /* R/src/Triangle.h:7-14 */
    bool intersectionExists;
    float t;
    float w0, w1, w2;
    unsigned int tIndex;
    Vec3 intersection;
    Vec3 normal;
};

class Mesh {
public:
    std::vector<MeshVertex> vertices;
    std::vector<MeshTriangle> triangles;
    std::vector<float> positions_array, normalsArray, uvs_array;
    std::vector<unsigned int> triangles_array;
    Material material;
    AABB aabb; /* stays (+inf, -inf) until computeAABB() is called, as in the reference */

    virtual ~Mesh() {}

    void computeAABB() { /* R/src/Mesh.cpp:5-19 */
        aabb = AABB();
        for (const MeshVertex& v : vertices)
            for (unsigned a = 0; a < 3; ++a) {
                if (v.position[a] < aabb.minCoor[a]) aabb.minCoor[a] = v.position[a];
                if (v.position[a] > aabb.maxCoor[a]) aabb.maxCoor[a] = v.position[a];
            }
    }

    /* R/src/Mesh.cpp:21-39.  Plain decimal "OFF" files go through the multi-threaded reader (OffReader.hpp),
     * which yields the values of the reference's token-by-token reader or declines; everything else is read
     * by loadOFF_reference. */
    void loadOFF(const std::string& filename) {
        std::vector<float> pos;
        std::vector<unsigned int> idx;
        if (!offreader::read(filename, pos, idx)) { loadOFF_reference(filename); return; }
        vertices.resize(pos.size() / 3);
        triangles.resize(idx.size() / 3);
        for (size_t i = 0; i < vertices.size(); ++i) vertices[i].position = Vec3(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]);
        for (size_t i = 0; i < triangles.size(); ++i)
            for (unsigned int j = 0; j < 3; ++j) triangles[i].v[j] = idx[3 * i + j];
        check_indices(filename);
    }

    /* A face that names a vertex the file does not have would make recomputeNormals and every later pass read and write
     * outside the vertex array: refuse the file the way a missing one is refused (R/src/Mesh.cpp:23-24). */
    void check_indices(const std::string& filename) const {
        for (size_t i = 0; i < triangles.size(); ++i)
            for (unsigned int j = 0; j < 3; ++j)
                if (triangles[i].v[j] >= vertices.size()) {
                    std::cerr << filename << ": face " << i << " refers to vertex " << triangles[i].v[j] << " of " << vertices.size() << std::endl;
                    exit(EXIT_FAILURE);
                }
    }

    /* The reference's reader, token by token (the corner count of a face is read and ignored, three indices
     * follow).  "COFF" files carry colour values after each vertex position (and optionally after a face's
     * indices); the reference takes them for coordinates and cannot load such a file -- here the rest of the
     * line is skipped. */
    void loadOFF_reference(const std::string& filename) {
        std::ifstream in(filename.c_str());
        if (!in) exit(EXIT_FAILURE);
        std::string tag;
        unsigned int nv, nt, ne;
        in >> tag >> nv >> nt >> ne;
        const bool coloured = tag == "COFF";
        const std::streamsize rest = std::numeric_limits<std::streamsize>::max();
        vertices.resize(nv);
        triangles.resize(nt);
        for (unsigned int i = 0; i < nv; ++i) {
            in >> vertices[i].position;
            if (coloured) in.ignore(rest, '\n');
        }
        for (unsigned int i = 0; i < nt; ++i) {
            int corners;
            in >> corners;
            for (unsigned int j = 0; j < 3; ++j) in >> triangles[i].v[j];
            if (coloured) in.ignore(rest, '\n');
        }
        check_indices(filename);
    }

    void recomputeNormals() { /* R/src/Mesh.cpp:41-54: unit face normals summed per vertex */
        for (MeshVertex& v : vertices) v.normal = Vec3(0.f, 0.f, 0.f);
        for (const MeshTriangle& t : triangles) {
            Vec3 n = Vec3::cross(vertices[t.v[1]].position - vertices[t.v[0]].position,
                                 vertices[t.v[2]].position - vertices[t.v[0]].position);
            n.normalize();
            for (unsigned j = 0; j < 3; ++j) vertices[t.v[j]].normal += n;
        }
        for (MeshVertex& v : vertices) v.normal.normalize();
    }

    void centerAndScaleToUnit() { /* R/src/Mesh.cpp:56-69 */
        Vec3 c(0, 0, 0);
        for (const MeshVertex& v : vertices) c += v.position;
        c /= vertices.size();
        float maxD = (vertices[0].position - c).length();
        for (const MeshVertex& v : vertices) {
            float m = (v.position - c).length();
            if (m > maxD) maxD = m;
        }
        for (MeshVertex& v : vertices) v.position = (v.position - c) / maxD;
    }

    virtual void build_arrays() { /* R/src/Mesh.h:118-125 */
        recomputeNormals();
        positions_array.resize(3 * vertices.size());
        normalsArray.resize(3 * vertices.size());
        uvs_array.resize(2 * vertices.size());
        for (size_t v = 0; v < vertices.size(); ++v) {
            for (unsigned a = 0; a < 3; ++a) {
                positions_array[3 * v + a] = vertices[v].position[a];
                normalsArray[3 * v + a] = vertices[v].normal[a];
            }
            uvs_array[2 * v] = vertices[v].u;
            uvs_array[2 * v + 1] = vertices[v].v;
        }
        triangles_array.resize(3 * triangles.size());
        for (size_t t = 0; t < triangles.size(); ++t)
            for (unsigned a = 0; a < 3; ++a) triangles_array[3 * t + a] = triangles[t].v[a];
    }

    void translate(Vec3 const& translation) {
        for (MeshVertex& v : vertices) v.position += translation;
    }
    void apply_transformation_matrix(Mat3 transform) {
        for (MeshVertex& v : vertices) v.position = transform * v.position;
    }
    void scale(Vec3 const& s) { apply_transformation_matrix(Mat3(s[0], 0., 0., 0., s[1], 0., 0., 0., s[2])); }
    /* rotations: the angle is narrowed to float, cos / sin are taken on that float and narrowed
     * again by the Mat3 constructor (R/src/Mesh.h:151-173) */
    void rotate_x(float angle) {
        float a = angle * M_PI / 180.;
        apply_transformation_matrix(Mat3(1., 0., 0., 0., std::cos((double)a), -std::sin((double)a), 0., std::sin((double)a), std::cos((double)a)));
    }
    void rotate_y(float angle) {
        float a = angle * M_PI / 180.;
        apply_transformation_matrix(Mat3(std::cos((double)a), 0., std::sin((double)a), 0., 1., 0., -std::sin((double)a), 0., std::cos((double)a)));
    }
    void rotate_z(float angle) {
        float a = angle * M_PI / 180.;
        apply_transformation_matrix(Mat3(std::cos((double)a), -std::sin((double)a), 0., std::sin((double)a), std::cos((double)a), 0., 0., 0., 1.));
    }

    /* Mesh::intersect(Ray), R/src/Mesh.h:206-281, on the GPU.  The batch form is the one to use:
     * the single-ray form uploads the mesh, asks one question and frees it again. */
    void intersect(const Ray* rays, size_t n, RayTriangleIntersection* out) const {
        rthost::ensure_device();
        std::vector<float> pos(3 * vertices.size()), nrm(3 * vertices.size());
        std::vector<uint32_t> tri(3 * triangles.size());
        for (size_t v = 0; v < vertices.size(); ++v)
            for (unsigned a = 0; a < 3; ++a) { pos[3 * v + a] = vertices[v].position[a]; nrm[3 * v + a] = vertices[v].normal[a]; }
        for (size_t t = 0; t < triangles.size(); ++t)
            for (unsigned a = 0; a < 3; ++a) tri[3 * t + a] = triangles[t].v[a];
        rt_material mat{};
        rt_mesh m{};
        m.n_vertices = (uint32_t)vertices.size();
        m.n_triangles = (uint32_t)triangles.size();
        rt_scene_desc d{};
        d.n_meshes = 1; d.meshes = &m; d.n_materials = 1; d.materials = &mat;
        d.n_vertices = m.n_vertices; d.n_triangles = m.n_triangles;
        d.positions = pos.data(); d.normals = nrm.data(); d.triangles = tri.data();
        d.kd_max_depth = 15; d.kd_max_leaf = 30; d.kd_epsilon = 0.01f;
        rt_scene_t h;
        rthost::require(rt_scene_create(&d, &h), "rt_scene_create");
        std::vector<float> org(3 * n), dir(3 * n);
        for (size_t i = 0; i < n; ++i)
            for (unsigned a = 0; a < 3; ++a) { org[3 * i + a] = rays[i].origin()[a]; dir[3 * i + a] = rays[i].direction()[a]; }
        std::vector<rt_hit> hits(n);
        rthost::require(rt_mesh_intersect(h, 0, org.data(), dir.data(), n, hits.data()), "rt_mesh_intersect");
        rt_scene_destroy(h);
        for (size_t i = 0; i < n; ++i) {
            const rt_hit& r = hits[i];
            RayTriangleIntersection& o = out[i];
            o.intersectionExists = r.hit != 0;
            o.t = r.hit ? r.t : FLT_MAX;
            o.tIndex = (unsigned int)r.tri; /* -1 when nothing was hit, as in the reference */
            o.w0 = r.hit ? 1.0f - r.u - r.v : 0.f; o.w1 = r.u; o.w2 = r.v;
            o.intersection = Vec3(r.p[0], r.p[1], r.p[2]);
            o.normal = Vec3(r.n[0], r.n[1], r.n[2]);
        }
    }
    RayTriangleIntersection intersect(Ray const& ray) const {
        RayTriangleIntersection r;
        intersect(&ray, 1, &r);
        return r;
    }
};
